#!/bin/bash
# Round 6: write-through (sc1) table stores -- the waveform kernel's batch form (the product) against streaming stores (-DWVS_STORE_NT=1,
# rounds 1-5) on a second box; the same in the narrow one-frame form (-DWVS_NARROW_STORE=2) and in the three-scope kernel's read-out
# (-DTRIO_STORE_SC1=1); parity of everything that writes a waveform table.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-n6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "waveform or three_scopes or dock or pipeline or bench_shapes or config3 or surface" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
for comp in 0x04 0x02 0x01 0x07; do
timeout 900 python tools/ab.py tools/_lib_wvs_nt.so $L --workload waveform --comp $comp --dists uniform,natural --rounds 5 --frames 64 > $out/ab_wvs_stores_$comp.txt 2>> $out/ab.err
sed "s/^/comp $comp  /" $out/ab_wvs_stores_$comp.txt
done
timeout 900 python tools/ab.py tools/_lib_wvs_nt.so $L --workload waveform_yuv --comp 0x20 --dists uniform --rounds 5 --frames 64 > $out/ab_wvs_stores_0x20.txt 2>> $out/ab.err; sed "s/^/comp 0x20  /" $out/ab_wvs_stores_0x20.txt
for fr in 1 3; do
timeout 900 python tools/ab.py $L tools/_lib_wvs_narrow_sc1.so --workload waveform --comp 0x07 --dists uniform,natural --rounds 5 --frames $fr > $out/ab_wvs_narrow_x$fr.txt 2>> $out/ab.err; sed "s/^/frames $fr  /" $out/ab_wvs_narrow_x$fr.txt
done
timeout 900 python tools/ab.py $L tools/_lib_trio_sc1.so --workload three_scope --dists uniform,natural,screen --rounds 5 --frames 64 > $out/ab_trio_sc1.txt 2>> $out/ab.err; cat $out/ab_trio_sc1.txt
tail -3 $out/ab.err
