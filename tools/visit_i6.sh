#!/bin/bash
# Round 6: the lean instruction sequence of the vectorscope kernels' run-length modes (bin_rle, keys_rle) against round 4's
# (-DVSS_RLE_LEAN=0): parity, the same-process A/B on the fused and the plain kernel, SQ counters of the fused kernel on screen content.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-i6}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
L=obs-color-monitor_amd/libscope_hip.so
python -c "import bench; print('kernel_sources_sha256_16', bench.kernel_fingerprint())" | tee $out/sources.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "vectorscope or fused or plain or config2 or bench_shapes" > $out/tests.log 2>&1; grep -E "passed|failed|error" $out/tests.log | tail -3
timeout 600 python tools/fuzz_parity.py --cases 200 --seed 69 > $out/fuzz_parity.txt 2>&1; tail -1 $out/fuzz_parity.txt
timeout 900 python tools/ab.py tools/_lib_lean0.so $L --workload vectorscope_rgb --dists screen,uniform,natural,flat,concentrated --rounds 5 --frames 64 \
  > $out/ab_fused_lean.txt 2>> $out/ab.err; cat $out/ab_fused_lean.txt
timeout 900 python tools/ab.py tools/_lib_lean0.so $L --workload vectorscope --dists screen,uniform,natural,flat --rounds 5 --frames 64 \
  > $out/ab_plain_lean.txt 2>> $out/ab.err; cat $out/ab_plain_lean.txt
tail -3 $out/ab.err
B="python $R/bench.py --steps 8 --warmup 2 --settle-seconds 0.2 --min-seconds 0.2 --no-cpu --no-extras --placement-candidates 1 --workload vectorscope_rgb --dist screen"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/pmc_vrgb_screen_$i -o p -- $B) > $out/pmc_vrgb_screen_$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc_vrgb_screen_1 $out/pmc_vrgb_screen_2 > $out/pmc_sq_vectorscope_rgb_x64_screen.txt 2>&1
grep -E "scope_vss_kernel|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|GRBM_GUI" $out/pmc_sq_vectorscope_rgb_x64_screen.txt | head -12
find $out -name "*counter_collection.csv" -size +1M -delete 2>/dev/null; find $out -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null; find $out -name "*.db" -size +1M -delete 2>/dev/null
du -sh $out
