#!/bin/bash
# Round 6, fifth visit: the absorb mode on ONE state register (trio_absorb_pair) for whole-flat waves + the run-length form for the rest.
# A/B: round 5's kernel (-DTRIO_RLE=0 -DTRIO_ABSORB=4), the product, no absorb mode at all (-DTRIO_ABSORB=0: visit d6's product),
# round 5's structure with the one-register state (-DTRIO_RLE=0).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-e6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio or pipeline or surface" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 68 > $out/fuzz_parity.txt 2>&1; tail -1 $out/fuzz_parity.txt
timeout 900 python tools/ab.py tools/_lib_trio_r5.so $L tools/_lib_trio_r5packed.so --workload three_scope \
  --dists screen,uniform,natural,flat,ramp,concentrated --rounds 5 --frames 64 > $out/ab_trio_rle_x64.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x64.txt
for fr in 3 1; do
timeout 900 python tools/ab.py tools/_lib_trio_r5.so $L --workload three_scope --dists screen,uniform,natural,flat --rounds 5 --frames $fr \
  > $out/ab_trio_rle_x$fr.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x$fr.txt
done
tail -3 $out/ab.err
du -sh $out
