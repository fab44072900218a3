#!/bin/bash
# One GPU-box visit: smoke, -m gpu tests, bench, variant sweep, rocprofv3 kernel trace.
# Usage (from the repo root on the box): bash tools/gpu_job.sh <tag> [stages...]
tag=${1:-job}; shift
stages=${@:-smoke tests bench sweep prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) stages: $stages" | tee $out/job.log
rocm-smi --showproductname 2>/dev/null | head -8 >> $out/job.log
for s in $stages; do
  t0=$(date +%s)
  case $s in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; rc=$? ;;
    tests) timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$? ;;
    bench) timeout 600 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; rc=$? ;;
    sweep) timeout 900 python tools/sweep.py --out $out/sweep.jsonl > $out/sweep.out 2> $out/sweep.log; rc=$? ;;
    prof)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o vss -- \
             python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu) > $out/prof.log 2>&1; rc=$? ;;
    *) echo "unknown stage $s"; rc=99 ;;
  esac
  echo "== stage $s rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $out/job.log
done
tail -5 $out/smoke.log $out/tests.log 2>/dev/null
cat $out/bench.json 2>/dev/null
tail -3 $out/bench.err 2>/dev/null
grep -E "vss_batch|ref" $out/sweep.log 2>/dev/null | head -50
find $out/prof -name "*stats*" 2>/dev/null | head
