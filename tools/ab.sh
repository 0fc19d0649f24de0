#!/bin/bash
# same-box A/B of two CFEN_TUNE settings: tools/ab.sh "<tune A>" "<tune B>" [rounds]   (box-to-box variance is ~3 %)
A="$1"; B="$2"; N="${3:-3}"
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then T="$A"; else T="$B"; fi
    echo -n "$v [$T]: "; CFEN_TUNE="$T" python bench.py --no-cpu-baseline --steps 50 2>&1 | tail -1 | cut -c52-70
  done
done
