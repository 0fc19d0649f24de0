#!/bin/bash
# One GPU-box session: parity tests, bench, per-launch profile, DCN bench, rocprofv3 passes.  Usage: tools/gpu_round.sh <tag> [what...]
# The rocprofv3 passes run ONE forward at a time on the SERIAL launch plan (--in-flight 1 --lanes 1: the plan every lane of the headline run replays); with
# GPU_MAX_HW_QUEUES=8 the two-lane plan overlaps its lanes for real and per-kernel durations / counters would be those of overlapping kernels.
# what: tests bench launches dcn stats pmc   (default: all).  Everything lands under gpurun_out/<tag>/.
tag=${1:-run}; shift
what=${@:-tests bench launches dcn stats pmc}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { echo BUILD FAILED; tail -30 $out/build.log; exit 1; }
for w in $what; do
  case $w in
    tests) timeout 1500 python3 -m pytest tests -m gpu -q --maxfail=15 > $out/tests.log 2>&1; echo "tests rc=$?"; tail -15 $out/tests.log;;
    smoke) timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $out/smoke.log;;
    bench) timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cat $out/bench.json | cut -c1-1500;;
    launches) timeout 600 python3 tools/profile_launches.py 8 fp16 > $out/launches.txt 2>&1; echo "launches rc=$?"; tail -45 $out/launches.txt;;
    dcn) timeout 600 python3 tools/bench_dcn.py --check > $out/dcn.jsonl 2>&1; echo "dcn rc=$?"; cat $out/dcn.jsonl | cut -c1-300;;
    dcnbwd) timeout 600 python3 tools/bench_dcn.py --backward --reps 10 > $out/dcn_bwd.jsonl 2>&1; echo "dcnbwd rc=$?"; cat $out/dcn_bwd.jsonl | cut -c1-300;;
    cfg4|cfg5) if [ $w = cfg4 ]; then A="--batch 4 --load-size 512"; PA="4 fp16 512 4"; else A="--batch 16 --hidden-dim-ratio 2"; PA="16 fp16 256 2"; fi
         timeout 400 python3 tools/profile_launches.py $PA > $out/${w}_launches.txt 2>&1; echo "$w launches rc=$?"
         (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/${w}_stats -- python3 $GRAFT_REPO_ROOT/bench.py $A --steps 10 --warmup 2 --no-cpu-baseline --in-flight 1 --lanes 1 --min-seconds 0 > $GRAFT_REPO_ROOT/$out/${w}_stats_bench.json 2> $GRAFT_REPO_ROOT/$out/${w}_stats.err); echo "$w stats rc=$?"
         for c in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
           n=${c%%:*}; ctr=${c#*:}
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$out/${w}_pmc_$n -- python3 $GRAFT_REPO_ROOT/bench.py $A --steps 2 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 --lanes 1 --min-seconds 0 > $GRAFT_REPO_ROOT/$out/${w}_pmc_$n.json 2> $GRAFT_REPO_ROOT/$out/${w}_pmc_$n.err); echo "$w pmc $n rc=$?"
         done
         mkdir -p $out/folded
         m=$(find $out/${w}_pmc_mfma -name "*counter_collection.csv" | head -1); f=$(find $out/${w}_pmc_fetch -name "*counter_collection.csv" | head -1); ww=$(find $out/${w}_pmc_write -name "*counter_collection.csv" | head -1)
         [ -n "$m" ] && [ -n "$f" ] && [ -n "$ww" ] && python3 tools/mfma_summary.py $m $f $ww $out/folded/r_${w} > $out/folded/${w}_summary.log 2>&1;;
    dcn_stats) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/dcn_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py --reps 5 > $GRAFT_REPO_ROOT/$out/dcn_stats.jsonl 2> $GRAFT_REPO_ROOT/$out/dcn_stats.err); echo "dcn_stats rc=$?";;
    dcn_pmc) for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
           n=${c%%:*}; ctr=${c#*:}
           (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$out/dcn_pmc_$n -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py --reps 3 > $GRAFT_REPO_ROOT/$out/dcn_pmc_$n.jsonl 2> $GRAFT_REPO_ROOT/$out/dcn_pmc_$n.err); echo "dcn pmc $n rc=$?"
         done;;
    dcnbwd_stats) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/dcnbwd_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py --backward --reps 5 --channels 24 > $GRAFT_REPO_ROOT/$out/dcnbwd_stats.jsonl 2> $GRAFT_REPO_ROOT/$out/dcnbwd_stats.err); echo "dcnbwd_stats rc=$?"; f=$(find $out/dcnbwd_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200;;
    stats) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --in-flight 1 --lanes 1 --min-seconds 0 > $GRAFT_REPO_ROOT/$out/stats_bench.json 2> $GRAFT_REPO_ROOT/$out/stats.err); echo "stats rc=$?";;
    sq) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-graph --in-flight 1 --lanes 1 --min-seconds 0 > $GRAFT_REPO_ROOT/$out/pmc_sq.json 2> $GRAFT_REPO_ROOT/$out/pmc_sq.err); echo "pmc sq rc=$?";;
    probe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe tools/repro/lds_graph_probe.hip > /dev/null 2>&1; timeout 120 /tmp/lds_probe > $out/lds_probe.txt 2>&1; cat $out/lds_probe.txt;;
    pmc) for c in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
           n=${c%%:*}; ctr=${c#*:}
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-graph --in-flight 1 --lanes 1 --min-seconds 0 > $GRAFT_REPO_ROOT/$out/pmc_$n.json 2> $GRAFT_REPO_ROOT/$out/pmc_$n.err); echo "pmc $n rc=$?"
         done;;
  esac
done
# keep the merged-back payload small: drop per-process agent info / huge traces beyond the CSVs we fold
find $out -name "*.db" -delete 2>/dev/null
# gpurun merges at most 64 MiB back: keep the CSV summaries the folding tools read (kernel stats, counter collection), drop traces / agent info / json
find $out -type f \( -name "*.json" -size +1M -o -name "*.rocpd" -o -name "*.pftrace" -o -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*_domain_stats.csv" \) -delete 2>/dev/null
# fold the per-dispatch counter tables into the per-kernel summaries ON THE BOX (the tables are 7 - 60 MB each), then drop them
mkdir -p $out/folded
m=$(find $out/pmc_mfma -name "*counter_collection.csv" 2>/dev/null | head -1)
f=$(find $out/pmc_fetch -name "*counter_collection.csv" 2>/dev/null | head -1)
w=$(find $out/pmc_write -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$m" ] && [ -n "$f" ] && [ -n "$w" ] && python3 tools/mfma_summary.py $m $f $w $out/folded/r > $out/folded/mfma_summary.log 2>&1
q=$(find $out/pmc_sq -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$q" ] && python3 tools/sq_summary.py $q > $out/folded/r_sq_wave_states.txt 2>&1
df=$(find $out/dcn_pmc_fetch -name "*counter_collection.csv" 2>/dev/null | head -1); dw=$(find $out/dcn_pmc_write -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$df" ] && [ -n "$dw" ] && python3 tools/pmc_summary.py $df $dw $out/folded/r_dcn > $out/folded/dcn_pmc_summary.log 2>&1
find $out -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $out; find $out -type f -size +4M -exec ls -la {} \;
