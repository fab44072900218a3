#!/bin/bash
# Round 6: the fused-K0 vectorscope's absorb state in one register (VSS_PACKED_STATE, the product) against four (-DVSS_PACKED_STATE=0).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-p6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or vectorscope or meet_inside" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 900 python tools/ab.py tools/_lib_ps0.so $L --workload vectorscope_rgb --dists uniform,natural,screen,flat,concentrated,ramp --rounds 7 --frames 64 \
  > $out/ab_fused_packed_state.txt 2>> $out/ab.err; cat $out/ab_fused_packed_state.txt
timeout 900 python tools/ab.py tools/_lib_ps0.so $L --workload vectorscope_rgb --dists uniform,flat --rounds 5 --frames 1 > $out/ab_fused_packed_state_x1.txt 2>> $out/ab.err; cat $out/ab_fused_packed_state_x1.txt
tail -3 $out/ab.err
