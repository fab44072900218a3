#!/bin/bash
# Round 6: the vectorscope's partial tables meeting INSIDE the launch (vss_write_out's fold; 2 .. 4 slices per frame) against the
# reduction kernel (vss.fold = 0): parity (the fold's tests, a stress run, every vectorscope / three-scope / bench-shape test), the
# same-process A/B of the WHOLE call and of the kernel alone, a default bench line.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-j6}
out=gpurun_out/$tag; mkdir -p $out
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "meet_inside or fold_stress or vectorscope or three_scopes or bench_shapes or config4 or multi or packed" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 70 > $out/fuzz_parity.txt 2>&1; tail -1 $out/fuzz_parity.txt
for wl in vectorscope vectorscope_rgb three_scope; do
timeout 900 python tools/ab.py $L $L --opts "vss.fold=0;vss.fold=1" --whole --workload $wl --dists uniform,natural,screen --rounds 7 --frames 64 > $out/ab_fold_whole_$wl.txt 2>> $out/ab.err; cat $out/ab_fold_whole_$wl.txt
timeout 900 python tools/ab.py $L $L --opts "vss.fold=0;vss.fold=1" --workload $wl --dists uniform,natural --rounds 5 --frames 64 > $out/ab_fold_kernel_$wl.txt 2>> $out/ab.err; cat $out/ab_fold_kernel_$wl.txt
done
tail -3 $out/ab.err
BENCH_DETAIL=$out/bench_detail.json timeout 900 python bench.py --no-cpu > $out/bench.json 2> $out/bench.err; python - <<PY
import json
d = json.load(open("$out/bench.json")); r = d["roofline"]
print("bench: value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "kernel", r["frac_kernel"], "first/placed same rounds", r.get("frac_first_pool_same_rounds"), r.get("frac_placed"), "parity", d["parity"], "three_scope", d.get("three_scope_gpix"), d.get("three_scope_frac_kernel"), "screen", d.get("screen_three_scope_frac"), "vrgb", d.get("vectorscope_rgb_frac_kernel"))
PY
tail -2 $out/bench.err
du -sh $out
