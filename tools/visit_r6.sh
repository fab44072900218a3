#!/bin/bash
# scope_frame_pool_alloc back on plain allocations, ranked by three interleaved rounds: does the placed pool hold against the first?
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r6; mkdir -p $out
timeout 600 python tools/pool_stress.py torch 8 > $out/pool_stress.txt 2>&1; echo "stress rc=$?"; tail -4 $out/pool_stress.txt
for i in 1 2 3 4; do BENCH_DETAIL=$out/d$i.json timeout 900 python bench.py --no-cpu > $out/full_$i.json 2> $out/full_$i.err; echo "full bench $i rc=$?"; done
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k pool 2>&1 | tail -2
