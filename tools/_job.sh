PMC_WORKLOADS="vectorscope waveform histogram vectorscope_rgb" bash tools/gpu_job.sh r2x smoke tests bench prof traffic
for wl in waveform histogram vectorscope_rgb; do
  timeout 300 python bench.py --workload $wl > gpurun_out/r2x/bench_$wl.json 2> gpurun_out/r2x/bench_$wl.err
  python -c "import json;d=json.load(open('gpurun_out/r2x/bench_$wl.json'));print('$wl', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], d['parity'])"
done
