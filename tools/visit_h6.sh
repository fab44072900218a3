#!/bin/bash
# Round 6: scope_frame_pool_alloc as ONE physical allocation tried at several address ranges (hipMemCreate / hipMemMap); the bench's
# placed-pool leg measured in alternating rounds with the first pool.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-h6}
out=gpurun_out/$tag; mkdir -p $out
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "frame_pool or abi or bench" > $out/tests.log 2>&1; tail -5 $out/tests.log
for i in 1 2 3; do
BENCH_DETAIL=$out/bench_detail_$i.json timeout 900 python bench.py --no-extras --no-cpu > $out/bench_$i.json 2> $out/bench_$i.err; python - <<PY
import json
d = json.load(open("$out/bench_$i.json")); r = d["roofline"]
print("run $i: value", d["value"], "frac", r["frac"], "kernel", r["frac_kernel"], "| same rounds: first pool", r.get("frac_first_pool_same_rounds"), r.get("frac_kernel_first_pool_same_rounds"), "placed", r.get("frac_placed"), r.get("frac_kernel_placed"))
dd = json.load(open("$out/bench_detail_$i.json")); print("   ", json.dumps(dd["config"].get("frame_pool_placement"))[:900])
PY
tail -2 $out/bench_$i.err
done
du -sh $out
