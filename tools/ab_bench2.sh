#!/bin/bash
# same-box A/B of CFEN_TUNE settings with several forwards in flight (INFLIGHT, default 4): tools/ab_bench2.sh <outfile> "<tune1>" "<tune2>" ...   ("-" = shipped defaults)
out=$1; shift
: > $out
for t in "$@"; do
  tt=$t; [ "$t" = "-" ] && tt=""
  r=$(CFEN_TUNE="$tt" timeout 600 python3 bench.py --no-cpu-baseline --no-extra-configs --min-seconds ${MINSEC:-0.6} --steps 60 --in-flight ${INFLIGHT:-4} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['self_check']['ok'], j['self_check'].get('image0_vs_reference_vectors'))")
  echo "[$t] $r" | tee -a $out
done
