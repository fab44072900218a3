#!/bin/bash
# Round 6, third visit: the three-scope kernel WITHOUT the per-lane absorb mode (the run-length form takes all flat content, with a
# whole-wave shortcut) against round 5's kernel (-DTRIO_RLE=0), visit a6's form (both modes, -DTRIO_ABSORB=1) and lower vote
# thresholds (-DTRIO_RLE_LO=32 / 16); parity first; the placement probe (sections 6-8 and the chunked pools, which b6 lost).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-c6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio or pipeline or surface" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 67 > $out/fuzz_parity.txt 2>&1; tail -1 $out/fuzz_parity.txt
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L tools/_lib_trio_both.so tools/_lib_trio_lo32.so tools/_lib_trio_lo16.so --workload three_scope \
  --dists screen,uniform,natural,flat,ramp,concentrated --rounds 5 --frames 64 > $out/ab_trio_rle_x64.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x64.txt
for fr in 3 1; do
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L --workload three_scope --dists screen,uniform,natural,flat --rounds 5 --frames $fr \
  > $out/ab_trio_rle_x$fr.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x$fr.txt
done
tail -3 $out/ab.err
timeout 900 ./tools/placement_vmm_probe 5 24 > $out/placement_vmm_probe.txt 2> $out/placement_vmm_probe.err; cut -c1-250 $out/placement_vmm_probe.txt | grep -vE "^pool  " ; tail -3 $out/placement_vmm_probe.err
du -sh $out
