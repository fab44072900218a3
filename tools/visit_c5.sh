#!/bin/bash
# Round 5, third visit: the whole -m gpu suite and the randomised parity sweep at the sources with the compact table's new pitch and
# the ROWS form; the waveform's one-channel mode against its strip / vector options.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-c5}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
for seed in 51 52 53; do timeout 600 python tools/fuzz_parity.py --cases 200 --seed $seed >> $out/fuzz_parity.txt 2>&1; done; tail -4 $out/fuzz_parity.txt
for o in "wvs.strip=16" "wvs.wide=0" "wvs.per_cu=3" "his.unroll=1"; do
  timeout 300 python tools/ab.py $L $L --workload waveform --comp 0x04 --dists uniform --rounds 3 --frames 64 --opts ";$o" >> $out/ab_waveform_one_channel_options.txt 2>> $out/ab_wvs.err
done; cat $out/ab_waveform_one_channel_options.txt
du -sh $out
