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
    bench) timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; rc=$? ;;
    bench1) timeout 600 python bench.py --no-extras --no-cpu > $out/bench_noextras.json 2> $out/bench1.err; rc=$?; cat $out/bench_noextras.json ;;
    benchfr) (timeout 600 python bench.py --frames 1 --no-cpu --no-extras > $out/bench_frames1.json 2> $out/benchfr.err; timeout 600 python bench.py --frames 3 --no-cpu --no-extras > $out/bench_frames3.json 2>> $out/benchfr.err); rc=$?; cat $out/bench_frames1.json $out/bench_frames3.json ;;
    benchred) timeout 600 python bench.py --force-reduce --no-cpu > $out/bench_force_reduce.json 2> $out/benchred.err; rc=$?; cat $out/bench_force_reduce.json; tail -3 $out/benchred.err ;;
    benchstream) timeout 600 python bench.py --workload stream > $out/bench_stream.json 2> $out/benchstream.err; rc=$?; cat $out/bench_stream.json; tail -3 $out/benchstream.err ;;
    prof1) for fr in 1 3; do (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_x$fr -o vss -- \
             python $GRAFT_REPO_ROOT/bench.py --frames $fr --steps 2000 --warmup 100 --no-cpu --no-extras) > $out/prof_x$fr.log 2>&1; done; rc=$?
           for f in $(find $out/prof_x1 $out/prof_x3 -name "*kernel_stats.csv"); do echo $f; cut -c1-70,150-260 $f | grep scope_ ; done ;;
    sweep) timeout 900 python tools/sweep.py $SWEEP_EXTRA --out $out/sweep.jsonl > $out/sweep.out 2> $out/sweep.log; rc=$? ;;
    prof)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o vss -- \
             python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu) > $out/prof.log 2>&1; rc=$? ;;
    hbm)   timeout 300 ./tools/hbm_read_bench > $out/hbm_read.txt 2>&1; rc=$?; sort -k7 -n -r $out/hbm_read.txt | head -12 ;;
    lds)   timeout 300 ./tools/lds_atomic_bench > $out/lds_atomic.txt 2>&1; rc=$? ;;
    sweepq) timeout 900 python tools/sweep.py --quick --out $out/sweep.jsonl > $out/sweep.out 2> $out/sweep.log; rc=$? ;;
    pmc)   # PMC passes on their own (no --stats / sys-trace), per the MI355X guide
           for d in ${PMC_DISTS:-uniform natural}; do
             (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY \
                --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq_$d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --no-cpu --placement-candidates 1 --dist $d $BENCH_EXTRA) > $out/pmc_sq_$d.log 2>&1
             (cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
                --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq2_$d -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --no-cpu --placement-candidates 1 --dist $d $BENCH_EXTRA) > $out/pmc_sq2_$d.log 2>&1
           done; rc=$?
           python tools/pmc_summary.py $out/pmc_sq_* $out/pmc_sq2_* > $out/pmc_summary.txt 2>&1; cat $out/pmc_summary.txt ;;
    traffic) # HBM bytes per step: FETCH_SIZE and WRITE_SIZE in separate PMC passes (TCC slots), kernel trace only; the frame pool is
           # taken as it comes (--placement-candidates 1): the placement probe's launches would be averaged into the step
           for spec in ${PMC_WORKLOADS:-vectorscope:64 vectorscope_rgb:64 waveform:64 histogram:64 three_scope:64 vectorscope:1 vectorscope:3 vectorscope_rgb:1 histogram:1 waveform:1 three_scope:1}; do
             wl=${spec%%:*}; fr=${spec##*:}
             for c in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_${wl}_x${fr}_$c -o p -- \
                  python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --settle-seconds 0.1 --min-seconds 0.05 --no-cpu --no-extras --placement-candidates 1 --workload $wl --frames $fr $BENCH_EXTRA) > $out/pmc_${wl}_x${fr}_$c.log 2>&1
             done
           done; rc=$?
           python tools/pmc_traffic_json.py $out $tag > $out/pmc_traffic.json 2> $out/pmc_traffic.err; cat $out/pmc_traffic.err
           python -c "
import json; d = json.load(open('$out/pmc_traffic.json'))
for e in d['entries']: print(e['workload'].ljust(16), 'x%-3d' % e['frames_per_step'], '%8.1f MB per step' % (e['bytes_per_launch'] / 1e6), ' algorithmic x %.3f' % (e['bytes_per_launch'] / (e['frames_per_step'] * (4 * 3840 * 2160 + {'vectorscope': 65536, 'vectorscope_rgb': 65536, 'histogram': 4096, 'waveform': 3932160, 'three_scope': 65536 + 4096 + 3932160}[e['workload']]))))
" ;;
    stream) (timeout 600 python tests/tools/stream_bench.py > $out/stream_max.json 2> $out/stream.err; timeout 600 python tests/tools/stream_bench.py --target-fps 240 --frames 720 > $out/stream_240.json 2>> $out/stream.err); rc=$?; cat $out/stream_max.json $out/stream_240.json; tail -3 $out/stream.err ;;
    trace) (python tests/tools/lds_trace_gen.py /tmp/lds_traces.bin 2> $out/lds_trace_model.txt && timeout 300 ./tools/lds_trace_bench /tmp/lds_traces.bin > $out/lds_trace.txt 2>&1); rc=$?; paste -d'|' $out/lds_trace_model.txt $out/lds_trace.txt | cut -c1-200 > $out/lds_trace_both.txt ;;
    *) echo "unknown stage $s"; rc=99 ;;
  esac
  echo "== stage $s rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $out/job.log
done
# the per-dispatch traces of these jobs run to hundreds of MB; gpurun copies at most 64 MiB back: keep the summaries
find $out -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
find $out -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
find $out -name "*.db" -size +2M -delete 2>/dev/null
du -sh $out | tail -1
tail -5 $out/smoke.log $out/tests.log 2>/dev/null
cat $out/bench.json 2>/dev/null
tail -3 $out/bench.err 2>/dev/null
grep -E "${SWEEP_GREP:-vss_batch|ref}" $out/sweep.log 2>/dev/null | head -80
cat $out/lds_atomic.txt 2>/dev/null
cat $out/lds_trace.txt 2>/dev/null
find $out/prof -name "*stats*" 2>/dev/null | head
for f in $(find $out/prof -name "*kernel_stats.csv" 2>/dev/null); do head -8 $f; done
