#!/bin/bash
# Round 6: the waveform's write side -- the batch form's table stores streaming (nt, the product) against plain cached stores
# (-DWVS_STORE_NT=0), every component mask of BASELINE config 3, same process, 64 x 4K.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-l6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
for comp in 0x04 0x07; do
timeout 900 python tools/ab.py $L tools/_lib_wvs_plain_stores.so tools/_lib_wvs_sc1_stores.so tools/_lib_wvs_sc1nt_stores.so --workload waveform --comp $comp --dists uniform,natural --rounds 5 --frames 64 > $out/ab_wvs_stores_$comp.txt 2>> $out/ab.err
sed "s/^/comp $comp  /" $out/ab_wvs_stores_$comp.txt
done
timeout 900 python tools/ab.py $L tools/_lib_wvs_plain_stores.so tools/_lib_wvs_sc1_stores.so tools/_lib_wvs_sc1nt_stores.so --workload waveform_yuv --comp 0x20 --dists uniform --rounds 5 --frames 64 > $out/ab_wvs_stores_0x20.txt 2>> $out/ab.err; sed "s/^/comp 0x20  /" $out/ab_wvs_stores_0x20.txt
tail -3 $out/ab.err
