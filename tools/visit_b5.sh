#!/bin/bash
# Round 5, second visit: the -m gpu suite at the new sources (ROWS form, fences, guards); padded planes with and without the ROWS
# form on the SAME frames; the three-scope kernel with another vectorscope table pitch and deeper load queues; the waveform's
# one-channel mode paced (s_sleep) -- is it the burstiness of its loads?; the one-frame launch's per-workgroup time stamps.
#   bash tools/visit_b5.sh <tag>
# Variant libraries this visit compared (not kept in the tree; from obs-color-monitor_amd/csrc):
#   make variant VARIANT_FLAGS=-DSCOPE_ROWS=0 VARIANT_OUT=../../tools/_lib_norows.so
#   make trio_variant VARIANT_FLAGS=-DVSS3_PITCH=116u VARIANT_TAG=head   (the default was 116 then; p122: -DVSS3_PITCH=122u; d9 / d12: -DTRIO_D=9 / 12)
#   _lib_pace1 / _lib_pace3: scope_kernels.hip with an s_sleep(1 / 3) behind every slot pair of a one-channel waveform unit (the macro is gone again)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-b5}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
fp=$(python -c "import bench; print(bench.kernel_fingerprint())")
echo "kernel_sources_sha256_16 $fp" | tee $out/sources.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
for wl in vectorscope_rgb vectorscope histogram; do
  timeout 300 python tools/ab.py $L tools/_lib_norows.so --workload $wl --dists uniform+pad,natural+pad,uniform --rounds 5 --frames 64 \
    >> $out/ab_padded_rows.txt 2>> $out/ab_padded_rows.err
done; cat $out/ab_padded_rows.txt
timeout 600 python tools/ab.py tools/_lib_trio_head.so tools/_lib_trio_p122.so tools/_lib_trio_d9.so tools/_lib_trio_d12.so \
  --workload three_scope --dists uniform,ramp,screen,natural,concentrated --rounds 5 --frames 64 > $out/ab_trio_pitch_depth.txt 2> $out/ab_trio.err; cat $out/ab_trio_pitch_depth.txt
for comp in 0x04 0x07 0x20; do
  timeout 300 python tools/ab.py $L tools/_lib_pace1.so tools/_lib_pace3.so --workload waveform --comp $comp --dists uniform,natural --rounds 5 --frames 64 \
    >> $out/ab_waveform_pace.txt 2>> $out/ab_pace.err
done; cat $out/ab_waveform_pace.txt
WG_TRACE_BACK_TO_BACK=1 timeout 300 python tools/wg_trace.py 1,3 > $out/wg_trace_frames1.txt 2>&1; cat $out/wg_trace_frames1.txt
du -sh $out
