#!/bin/bash
# Round 5, fourth visit: the three-scope kernel's lazy absorb form against round 4's eager one -- parity first (the GPU suite's
# three-scope tests, a fuzz seed), then the same-process A/B on every content, then the SQ counters on screen content.
# The eager library: round 4's absorb form = the tree's; the library under test was the tree with profiles/round5/dropped_lazy_absorb.diff applied
# (this script ran with the lazy form as libscope_hip.so and `make trio_variant VARIANT_FLAGS=-DTRIO_LAZY=0 VARIANT_TAG=eager` beside it).
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-d5}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio or surface or pipeline" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 61 > $out/fuzz_parity.txt 2>&1; tail -2 $out/fuzz_parity.txt
timeout 600 python tools/ab.py $L tools/_lib_trio_eager.so --workload three_scope --dists screen,uniform,natural,flat,ramp,concentrated --rounds 5 --frames 64 \
  > $out/ab_trio_lazy_absorb.txt 2> $out/ab.err; cat $out/ab_trio_lazy_absorb.txt
B="python $R/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload three_scope --dist screen"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM \
   --kernel-trace --output-format csv -d $R/$out/pmc_screen -o p -- $B) > $out/pmc_screen.log 2>&1
python tools/pmc_summary.py $out/pmc_screen > $out/pmc_sq_three_scope_screen_lazy.txt 2>&1; grep -A8 scope_trio $out/pmc_sq_three_scope_screen_lazy.txt
find $out -name "*counter_collection.csv" -size +1M -delete 2>/dev/null; find $out -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
du -sh $out
