#!/bin/bash
# Round 6, fourth visit: the whole-wave shortcut of the run-length form asked only behind an all-flat vote (visit c6: asked in front of
# every slot it cost screen content 8 %).  A/B: round 5's kernel (-DTRIO_RLE=0), the product, the product without the shortcut
# (-DTRIO_RLE_WHOLE=0), visit a6's form with the structured slot exit (-DTRIO_ABSORB=1 -DTRIO_RLE_WHOLE=0).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-d6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio or pipeline or surface" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L tools/_lib_trio_whole0.so tools/_lib_trio_a6form.so --workload three_scope \
  --dists screen,uniform,natural,flat,ramp,concentrated --rounds 5 --frames 64 > $out/ab_trio_rle_x64.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x64.txt
for fr in 3 1; do
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L tools/_lib_trio_whole0.so --workload three_scope --dists screen,uniform,natural,flat --rounds 5 --frames $fr \
  > $out/ab_trio_rle_x$fr.txt 2>> $out/ab.err; cat $out/ab_trio_rle_x$fr.txt
done
tail -3 $out/ab.err
du -sh $out
