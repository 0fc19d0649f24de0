#!/bin/bash
# The other BASELINE configurations and the sibling generators on the same code (informational; the headline is `python bench.py`):
#   tools/bench_configs.sh <outfile>
out=$1; : > $out
run() { echo "== $*" >> $out; timeout -k 10 600 python3 bench.py --no-cpu-baseline --min-seconds 0.6 --steps 30 "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(json.dumps({'value': j['value'], 'ms_per_step': j['ms_per_step'], 'workload': j['config']['workload'], 'gflop_per_image': j['config']['gflop_per_image'], 'whole_forward_tflops': j['roofline']['whole_forward_tflops'], 'self_check': j['self_check']}))" >> $out; }
run --batch 4 --load-size 512                      # BASELINE cfg 4: 1024 x 1024
run --batch 16 --hidden-dim-ratio 2                # BASELINE cfg 5: hidden_dim_ratio 2
run --dtype fp32                                   # the exact-fp32 path
run --variant v5
run --variant cfs --load-size 256
run --variant crs --load-size 256
cat $out
