bash tools/gpu_job.sh r3w smoke tests bench prof
out=gpurun_out/r3w
for wl in histogram waveform vectorscope_rgb; do timeout 600 python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err; done
cat $out/bench*.json | cut -c1-700
find $out -name "*.csv" -size +2M -delete
du -sh $out
