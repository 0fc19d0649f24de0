mkdir -p gpurun_out/r6c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_probe tools/repro/mfma_rate_probe.hip && /tmp/mfma_rate_probe > gpurun_out/r6c/mfma_rate_probe.jsonl; cat gpurun_out/r6c/mfma_rate_probe.jsonl
run() { # dir args...
  d=$1; shift
  (cd $d && timeout 600 python3 bench.py "$@" --no-cpu-baseline --no-extra-configs --min-seconds 0.6 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['self_check']['ok'])")
}
for rep in 1 2; do
  for d in . _ab/r04; do
    echo "cfg5 4-in-flight [$d] $(run $d --batch 16 --hidden-dim-ratio 2 --steps 60 --in-flight 4)" | tee -a gpurun_out/r6c/ab_r04_vs_head.txt
    echo "cfg5 1-in-flight [$d] $(run $d --batch 16 --hidden-dim-ratio 2 --steps 30 --in-flight 1)" | tee -a gpurun_out/r6c/ab_r04_vs_head.txt
    echo "cfg5 20-step regions [$d] $(run $d --batch 16 --hidden-dim-ratio 2 --steps 20 --in-flight 4)" | tee -a gpurun_out/r6c/ab_r04_vs_head.txt
    echo "cfg2 1-in-flight [$d] $(run $d --steps 30 --in-flight 1)" | tee -a gpurun_out/r6c/ab_r04_vs_head.txt
  done
done
python3 -m pytest tests/test_hip_net.py -q -x -m gpu -k "reference_init_weights_fp16" -s 2>&1 | grep -i "refinit\|passed\|failed" | tee gpurun_out/r6c/refinit_fp16.txt
