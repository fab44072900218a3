#!/bin/bash
# Round 5, first visit of the GPU box: is the bench line short and whole; where does scope_trio_kernel's time go on ramp and
# screen content (TRIO_EXP builds with parts compiled out, SQ counters, per-wave phase times); the fused vectorscope with and
# without its run-length mode on the SAME frames (noise, natural, concentrated, padded pitch); the waveform kernel's one-channel
# forms against its RGB form under the SQ counters; the one-frame vectorscope launch's per-workgroup time stamps.
#   bash tools/visit_a5.sh <tag>      (from the repo root; results under gpurun_out/<tag>/)
# Variant libraries this visit compared (not kept in the tree; from obs-color-monitor_amd/csrc at round 4's kernel sources):
#   for e in 1 2 4 8 3; do make trio_variant VARIANT_FLAGS=-DTRIO_EXP=$e VARIANT_TAG=exp$e; done
#   make variant VARIANT_FLAGS=-DVSS_RLE=0 VARIANT_OUT=../../tools/_lib_vss_norle.so;  make trace;  make trio_variant VARIANT_FLAGS=-DTRIO_TRACE VARIANT_TAG=trace
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-a5}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
L=obs-color-monitor_amd/libscope_hip.so
fp=$(python -c "import bench; print(bench.kernel_fingerprint())")
echo "kernel_sources_sha256_16 $fp" | tee $out/sources.txt
timeout 700 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$? line bytes=$(wc -c < $out/bench.json)"
cp bench_detail.json $out/bench_detail.json 2>/dev/null
python - <<E
import json
d = json.load(open("$out/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "parity", "all_parity") if k in d}, d["roofline"], d.get("cpu_baseline"))
E
# the three-scope kernel with parts compiled out (wrong tables; timing only): 1 no waveform adds, 2 no vectorscope adds, 4 no K0,
# 8 no clamp passes, 3 neither table
timeout 600 python tools/ab.py $L tools/_lib_trio_exp1.so tools/_lib_trio_exp2.so tools/_lib_trio_exp4.so tools/_lib_trio_exp8.so tools/_lib_trio_exp3.so \
  --workload three_scope --dists uniform,ramp,screen,natural --rounds 3 > $out/trio_parts.txt 2> $out/trio_parts.err; cat $out/trio_parts.txt
# per-wave phase times on ramp and screen
timeout 300 python tools/trio_trace.py 64 uniform,ramp,screen > $out/trio_trace.txt 2>&1; cat $out/trio_trace.txt
# SQ counters of the three-scope kernel on ramp and screen content (two passes of eight counters, kernel trace only)
for d in ramp screen; do
  B="python $R/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload three_scope --dist $d"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/pmc_trio_${d}_$i -o p -- $B) > $out/pmc_trio_${d}_$i.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_trio_${d}_1 $out/pmc_trio_${d}_2 > $out/pmc_sq_three_scope_x64_${d}_$fp.txt 2>&1
  grep -A20 scope_trio $out/pmc_sq_three_scope_x64_${d}_$fp.txt | head -44
done
# the fused vectorscope: HEAD against the same sources without the run-length mode (-DVSS_RLE=0), same process, same frames
timeout 600 python tools/ab.py $L tools/_lib_vss_norle.so --workload vectorscope_rgb --dists uniform,natural,concentrated,uniform+pad,screen --rounds 5 --frames 64 \
  > $out/ab_fused_vectorscope_rle.txt 2> $out/ab_rle.err; cat $out/ab_fused_vectorscope_rle.txt
# the waveform kernel: one channel against three, SQ counters
for comp in 0x07 0x04; do
  B="python $R/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload waveform --components $comp"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/pmc_wvs_${comp}_$i -o p -- $B) > $out/pmc_wvs_${comp}_$i.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_wvs_${comp}_1 $out/pmc_wvs_${comp}_2 > $out/pmc_sq_waveform_${comp}_x64_$fp.txt 2>&1
  grep -A20 scope_wvs $out/pmc_sq_waveform_${comp}_x64_$fp.txt | head -22
done
# one frame per call: when do the 256 workgroups start, leave their pixel loop, finish their flush
WG_TRACE_BACK_TO_BACK=1 timeout 300 python tools/wg_trace.py 1,3 > $out/wg_trace_frames1.txt 2>&1; cat $out/wg_trace_frames1.txt
find $out -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
find $out -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
find $out -name "*.db" -size +1M -delete 2>/dev/null
du -sh $out
