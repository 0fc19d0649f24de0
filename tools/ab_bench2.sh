#!/bin/bash
# same-box A/B like ab_bench.sh, with extra bench.py flags per case: tools/ab_bench2.sh <outfile> "<tune>|<flags>" ...   ("-" = defaults)
out=$1; shift
: > $out
for c in "$@"; do
  t=${c%%|*}; f=""; [[ "$c" == *"|"* ]] && f=${c#*|}
  tt=$t; [ "$t" = "-" ] && tt=""
  r=$(CFEN_TUNE="$tt" timeout -k 10 600 python3 bench.py --no-cpu-baseline --min-seconds 0.6 --steps 60 $f 2>>$out.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['self_check']['ok'], json.dumps({k: round(v['ms'], 3) for k, v in j['kernel_classes'].items()}))")
  echo "[$c] $r" | tee -a $out
done
