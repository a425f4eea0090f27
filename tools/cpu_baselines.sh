#!/bin/bash
# GPU box: the CPU oracle (reference box rule, 8 and 16 threads) on every BASELINE workload, beside a short GPU run.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
for w in weekend cornell smoke; do
  timeout -k 10 280 python bench.py --workload $w --steps 1 --warmup 1 --cpu-spp 16 > gpurun_out/cpu_$w.json 2> gpurun_out/cpu_$w.err || exit 1
  echo "$w done"
done
timeout -k 10 400 python bench.py --workload mesh --bvh lbvh --steps 1 --warmup 1 --cpu-spp 1 > gpurun_out/cpu_mesh_lbvh.json 2> gpurun_out/cpu_mesh_lbvh.err || exit 1
echo "mesh done"
