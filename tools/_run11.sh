run() { env $1 python3 bench.py --brief --no-cpu-baseline --no-extra-configs --in-flight $3 --min-seconds 0.6 $2 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['self_check']['ok'], j.get('lane_search'))"; }
echo "[4 lanes] $(run A=1 '' 4)"
echo "[4 lanes, search] $(run CFEN_BENCH_LANE_SEARCH=1 '' 4)"
echo "[5 lanes, search] $(run CFEN_BENCH_LANE_SEARCH=1 '' 5)"
echo "[4 lanes] $(run A=1 '' 4)"
echo "[4 lanes, search] $(run CFEN_BENCH_LANE_SEARCH=1 '' 4)"
