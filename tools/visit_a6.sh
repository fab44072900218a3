#!/bin/bash
# Round 6, first visit: the three-scope kernel's run-length form (trio_rle_pair) -- parity first (the GPU suite's three-scope tests,
# a fuzz seed), the same-process A/B on every content against the same sources with -DTRIO_RLE=0 (round 5's kernel), -DTRIO_RLE_HI=65
# (no per-lane absorb mode at all) and =60, SQ counters on screen content before / after; SQ counters of the fused-K0 vectorscope on
# screen content and noise at these sources; the placement probe (tools/placement_vmm_probe.hip).
# Built beside the product: make -C obs-color-monitor_amd/csrc trio_variant VARIANT_FLAGS=-DTRIO_RLE=0 VARIANT_TAG=rle0 (etc.)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-a6}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q -k "three_scopes or dock or trio or surface or pipeline" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 66 > $out/fuzz_parity.txt 2>&1; tail -2 $out/fuzz_parity.txt
timeout 900 python tools/ab.py tools/_lib_trio_rle0.so $L tools/_lib_trio_hi65.so tools/_lib_trio_hi60.so --workload three_scope \
  --dists screen,uniform,natural,flat,ramp,concentrated --rounds 5 --frames 64 > $out/ab_trio_rle.txt 2> $out/ab.err; cat $out/ab_trio_rle.txt
timeout 600 python tools/ab.py tools/_lib_trio_rle0.so $L --workload three_scope --dists screen,uniform,natural --rounds 5 --frames 3 \
  > $out/ab_trio_rle_frames3.txt 2>> $out/ab.err; cat $out/ab_trio_rle_frames3.txt
timeout 600 python tools/ab.py tools/_lib_trio_rle0.so $L --workload three_scope --dists screen,uniform,natural --rounds 5 --frames 1 \
  > $out/ab_trio_rle_frames1.txt 2>> $out/ab.err; cat $out/ab_trio_rle_frames1.txt
tail -3 $out/ab.err
pmc2() { # $1 = name, $2 = library, rest = bench arguments: two SQ passes of eight counters each
  name=$1; lib=$2; shift 2
  B="python $R/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 $* ${lib:+--lib $lib}"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/pmc_${name}_$i -o p -- $B) > $out/pmc_${name}_$i.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_${name}_1 $out/pmc_${name}_2 > $out/pmc_sq_$name.txt 2>&1
  grep -E "scope_(trio|vss)_kernel|SQ_INSTS_VALU|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_WAIT_INST_LDS|SQ_WAVE_CYCLES|SQ_INSTS_LDS" $out/pmc_sq_$name.txt | head -16
}
pmc2 three_scope_x64_screen_rle "" --workload three_scope --dist screen
pmc2 three_scope_x64_screen_rle0 $R/tools/_lib_trio_rle0.so --workload three_scope --dist screen
pmc2 vectorscope_rgb_x64_screen "" --workload vectorscope_rgb --dist screen
pmc2 vectorscope_rgb_x64_uniform "" --workload vectorscope_rgb --dist uniform
timeout 600 ./tools/placement_vmm_probe 5 24 > $out/placement_vmm_probe.txt 2> $out/placement_vmm_probe.err; tail -80 $out/placement_vmm_probe.txt; tail -3 $out/placement_vmm_probe.err
find $out -name "*counter_collection.csv" -size +1M -delete 2>/dev/null; find $out -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
find $out -name "*.db" -size +1M -delete 2>/dev/null
du -sh $out
