#!/bin/bash
# Copies what an evidence visit (tools/visit_full.sh <tag>) left under gpurun_out/<tag>/ and is worth keeping into profiles/<round>/:
# the fingerprint, the test summary, both bench lines + the detail file, the rocprofv3 statistics, the PMC traffic and SQ summaries.
#   bash tools/keep_visit.sh <tag> profiles/round5
tag=$1; dest=$2; o=gpurun_out/$tag
fp=$(awk '{print $2}' $o/sources.txt)
mkdir -p $dest
cp $o/sources.txt $dest/sources_$tag.txt
(tail -1 $o/smoke.log; grep -E "passed|failed" $o/tests.log) > $dest/gpu_tests_$tag.txt
cp $o/bench.json $dest/bench_$tag.json
cp $o/bench_detail.json $dest/bench_detail_$tag.json
cp $o/bench_first.json $dest/bench_${tag}_first_pass_without_traffic.json 2>/dev/null
cp $o/rocprof_kernel_stats_${tag}_*_$fp.csv $o/bench_*_under_rocprof_$fp.json $o/pmc_sq_three_scope_x64_*_$fp.txt $o/pmc_sq_vectorscope_rgb_x64_*_$fp.txt $dest/
cp $o/fuzz_parity.txt $dest/fuzz_parity_$tag.txt 2>/dev/null
cp $o/pmc_traffic.json $dest/pmc_traffic.json
ls $dest | wc -l
