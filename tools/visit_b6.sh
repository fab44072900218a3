#!/bin/bash
# Round 6, second visit: the run-length form laid out behind the loop (__builtin_expect: visit a6 found its mere presence cost noise
# 6 %), A/B against -DTRIO_RLE=0 and -DTRIO_RLE_FASTONLY=1; the placement probe, second form (translation latencies beside the rates).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-b6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
for fr in 64 3 1; do
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L tools/_lib_trio_fastonly.so --workload three_scope \
  --dists screen,uniform,natural,flat --rounds 5 --frames $fr > $out/ab_trio_rle_x$fr.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x$fr.txt
done
tail -3 $out/ab.err
timeout 900 ./tools/placement_vmm_probe 5 24 > $out/placement_vmm_probe.txt 2> $out/placement_vmm_probe.err; cat $out/placement_vmm_probe.txt | cut -c1-260; tail -3 $out/placement_vmm_probe.err
du -sh $out
