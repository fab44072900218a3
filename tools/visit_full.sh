#!/bin/bash
# One evidence visit of the GPU box, all from ONE set of sources (the fingerprint bench.py prints as
# config.kernel_sources_sha256_16 goes into the file names): smoke, the full -m gpu suite, the default bench line,
# rocprofv3 --kernel-trace --stats of the headline launch alone and of the three-scope launch, PMC traffic of every
# workload (tools/gpu_job.sh traffic), SQ counters of the three-scope kernel on noise, ramp and screen content, and the default
# line once more with this visit's own PMC traffic in it.
#   bash tools/visit_full.sh <tag>      (from the repo root; results under gpurun_out/<tag>/)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-visit}
out=gpurun_out/$tag; mkdir -p $out
fp=$(python -c "import bench; print(bench.kernel_fingerprint())")
echo "kernel_sources_sha256_16 $fp" | tee $out/sources.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -2
BENCH_DETAIL=$out/bench_detail_first.json timeout 900 python bench.py > $out/bench_first.json 2> $out/bench_first.err; echo "bench rc=$? bytes=$(wc -c < $out/bench_first.json)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_headline -o vss -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 100 --no-cpu --no-extras --placement-candidates 1) > $out/prof_headline.log 2>&1
for f in $(find $out/prof_headline -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_headline_only_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -3; done
grep -E '^\{' $out/prof_headline.log | head -1 > $out/bench_headline_under_rocprof_$fp.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_trio -o t -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 50 --no-cpu --no-extras --placement-candidates 1 --workload three_scope) > $out/prof_trio.log 2>&1
for f in $(find $out/prof_trio -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_three_scope_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -2; done
for wl in vectorscope_rgb waveform histogram; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_$wl -o w -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 1000 --warmup 50 --no-cpu --no-extras --placement-candidates 1 --workload $wl) > $out/prof_$wl.log 2>&1
  for f in $(find $out/prof_$wl -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_${wl}_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -1; done
  grep -E '^\{' $out/prof_$wl.log | head -1 > $out/bench_${wl}_under_rocprof_$fp.json; done
for fr in 1 3; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_x$fr -o v -- \
   python $GRAFT_REPO_ROOT/bench.py --frames $fr --steps 2000 --warmup 100 --no-cpu --no-extras --placement-candidates 1) > $out/prof_x$fr.log 2>&1
  for f in $(find $out/prof_x$fr -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_frames${fr}_$fp.csv; done; done
bash tools/gpu_job.sh $tag traffic > $out/traffic_stage.log 2>&1; grep -E " x[0-9]+ .*algorithmic" $out/traffic_stage.log
for d in uniform ramp screen; do
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload three_scope --dist $d"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq_trio_${d}_$i -o p -- $B) > $out/pmc_sq_trio_${d}_$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc_sq_trio_${d}_1 $out/pmc_sq_trio_${d}_2 > $out/pmc_sq_three_scope_x64_${d}_$fp.txt 2>&1
done
for d in uniform screen; do
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload vectorscope_rgb --dist $d"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq_vrgb_${d}_$i -o p -- $B) > $out/pmc_sq_vrgb_${d}_$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc_sq_vrgb_${d}_1 $out/pmc_sq_vrgb_${d}_2 > $out/pmc_sq_vectorscope_rgb_x64_${d}_$fp.txt 2>&1
done
for d in ramp screen natural; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_trio_$d -o t -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 600 --warmup 50 --no-cpu --no-extras --placement-candidates 1 --workload three_scope --dist $d) > $out/prof_trio_$d.log 2>&1
  for f in $(find $out/prof_trio_$d -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_three_scope_${d}_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -1; done; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_vrgb_screen -o t -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 600 --warmup 50 --no-cpu --no-extras --placement-candidates 1 --workload vectorscope_rgb --dist screen) > $out/prof_vrgb_screen.log 2>&1
for f in $(find $out/prof_vrgb_screen -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_vectorscope_rgb_screen_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -1; done
for comp in 0x04; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_wvs_r -o t -- \
   python $GRAFT_REPO_ROOT/bench.py --steps 600 --warmup 50 --no-cpu --no-extras --placement-candidates 1 --workload waveform --components $comp) > $out/prof_wvs_r.log 2>&1
  for f in $(find $out/prof_wvs_r -name "*kernel_stats.csv"); do cp $f $out/rocprof_kernel_stats_${tag}_waveform_r_$fp.csv; cut -c1-60,150-330 $f | grep -E "scope_" | head -1; done; done
for seed in 81 82 83 84 85; do timeout 300 python tools/fuzz_parity.py --cases 200 --seed $seed 2>&1 | tail -1; done > $out/fuzz_parity.txt; cat $out/fuzz_parity.txt
# the default line once more, now that this visit's own PMC traffic exists: the KEPT line carries roofline.traffic (round 4's did not)
BENCH_PMC_DIR=$out BENCH_DETAIL=$out/bench_detail.json timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench (with this visit's traffic) rc=$? bytes=$(wc -c < $out/bench.json)"
find $out -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
find $out -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
find $out -name "*.db" -size +1M -delete 2>/dev/null
du -sh $out
