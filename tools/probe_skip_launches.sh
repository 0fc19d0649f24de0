#!/bin/bash
# what a launch costs with several forwards in flight: bench.py with launch number i of the forward left out (net.skip_from / net.skip_to; results invalid), i = $2 .. $3
# usage: tools/probe_skip_launches.sh <outfile> <first> <last>      (launch numbers = line numbers of profiles/rNN_launches.txt, from 0)
out=$1; a=$2; b=$3
args=("-")
for ((i = a; i <= b; ++i)); do args+=("net.skip_from=$i,net.skip_to=$i"); done
args+=("-")
MINSEC=${MINSEC:-0.4} bash tools/ab_bench2.sh $out "${args[@]}"
