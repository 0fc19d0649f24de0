#!/bin/bash
# same-box A/B of CFEN_TUNE settings on SEVERAL configurations (defaults were tuned on the headline shape only).
# tools/ab_cfg.sh <outfile> "<cfgs: 2 4 5>" "<tune1>" "<tune2>" ...   ("-" = shipped defaults); INFLIGHT (default 4), MINSEC (default 0.6)
out=$1; shift
cfgs=$1; shift
: > $out
for c in $cfgs; do
  case $c in
    2) A="";;
    4) A="--batch 4 --load-size 512";;
    5) A="--batch 16 --hidden-dim-ratio 2";;
  esac
  for t in "$@"; do
    tt=$t; [ "$t" = "-" ] && tt=""
    r=$(CFEN_TUNE="$tt" timeout 600 python3 bench.py $A --no-cpu-baseline --no-extra-configs --min-seconds ${MINSEC:-0.6} --steps 60 --in-flight ${INFLIGHT:-4} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['self_check']['ok'], j['self_check'].get('image0_vs_reference_vectors'))")
    echo "cfg$c [$t] $r" | tee -a $out
  done
done
