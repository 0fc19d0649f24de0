#!/bin/bash
# One GPU-box session: parity tests, bench, per-launch profile, DCN bench, rocprofv3 passes.  Usage: tools/gpu_round.sh <tag> [what...]
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
    dcnbwd_stats) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/dcnbwd_stats -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py --backward --reps 5 --channels 24 > $GRAFT_REPO_ROOT/$out/dcnbwd_stats.jsonl 2> $GRAFT_REPO_ROOT/$out/dcnbwd_stats.err); echo "dcnbwd_stats rc=$?"; f=$(find $out/dcnbwd_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200;;
    stats) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 0 > $GRAFT_REPO_ROOT/$out/stats_bench.json 2> $GRAFT_REPO_ROOT/$out/stats.err); echo "stats rc=$?";;
    sq) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --min-seconds 0 > $GRAFT_REPO_ROOT/$out/pmc_sq.json 2> $GRAFT_REPO_ROOT/$out/pmc_sq.err); echo "pmc sq rc=$?";;
    probe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe tools/repro/lds_graph_probe.hip > /dev/null 2>&1; timeout 120 /tmp/lds_probe > $out/lds_probe.txt 2>&1; cat $out/lds_probe.txt;;
    pmc) for c in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
           n=${c%%:*}; ctr=${c#*:}
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --min-seconds 0 > $GRAFT_REPO_ROOT/$out/pmc_$n.json 2> $GRAFT_REPO_ROOT/$out/pmc_$n.err); echo "pmc $n rc=$?"
         done;;
  esac
done
# keep the merged-back payload small: drop per-process agent info / huge traces beyond the CSVs we fold
find $out -name "*.db" -delete 2>/dev/null
# gpurun merges at most 64 MiB back: keep the CSV summaries the folding tools read (kernel stats, counter collection), drop traces / agent info / json
find $out -type f \( -name "*.json" -size +1M -o -name "*.rocpd" -o -name "*.pftrace" -o -name "*kernel_trace.csv" -o -name "*agent_info.csv" -o -name "*_domain_stats.csv" \) -delete 2>/dev/null
du -sh $out; find $out -type f -size +4M -exec ls -la {} \;
