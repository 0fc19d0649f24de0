#!/bin/bash
# Fold what tools/gpu_round.sh left under gpurun_out/<tag>/ into profiles/<prefix>_*:  tools/fold_profiles.sh <tag> <prefix>
tag=$1; pre=profiles/$2; d=gpurun_out/$tag
[ -f $d/bench.json ] && cp $d/bench.json ${pre}_bench_default_run.json
[ -f $d/stats_bench.json ] && cp $d/stats_bench.json ${pre}_bench_line_under_rocprof.json
ks=$(find $d/stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$ks" ] && cp $ks ${pre}_bench_kernel_stats.csv
[ -f $d/launches.txt ] && grep -v "amdgpu.ids" $d/launches.txt > ${pre}_launches.txt
[ -f $d/dcn.jsonl ] && grep "^{" $d/dcn.jsonl > ${pre}_dcn_bench.jsonl
[ -f $d/dcn_bwd.jsonl ] && grep "^{" $d/dcn_bwd.jsonl > ${pre}_dcn_backward_bench.jsonl
m=$(find $d/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1)
f=$(find $d/pmc_fetch -name "*counter_collection.csv" 2>/dev/null | head -1)
w=$(find $d/pmc_write -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$m" ] && [ -n "$f" ] && [ -n "$w" ] && python3 tools/mfma_summary.py $m $f $w $pre
q=$(find $d/pmc_sq -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$q" ] && python3 tools/sq_summary.py $q > ${pre}_sq_wave_states.txt
# (round 3: the counter tables are folded on the GPU box by tools/gpu_round.sh -- they exceed what gpurun merges back)
for f in $d/folded/r_*; do [ -f "$f" ] && cp $f ${pre}_${f##*/r_}; done
ls -la ${pre}_*
for w in cfg4 cfg5; do
  ks=$(find $d/${w}_stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$ks" ] && cp $ks ${pre}_${w}_bench_kernel_stats.csv
  [ -f $d/${w}_stats_bench.json ] && cp $d/${w}_stats_bench.json ${pre}_${w}_bench_line_under_rocprof.json
  [ -f $d/${w}_launches.txt ] && grep -v "amdgpu.ids" $d/${w}_launches.txt > ${pre}_${w}_launches.txt
done
ks=$(find $d/dcn_stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$ks" ] && cp $ks ${pre}_dcn_forward_kernel_stats.csv
