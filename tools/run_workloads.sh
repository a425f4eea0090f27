#!/bin/bash
# GPU box: bench.py on the workloads other than the headline one, and the timing of rt_bvh_build; JSON lines under
# gpurun_out/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for w in weekend cornell smoke; do
  timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 --cpu-spp 0 > gpurun_out/w_$w.json 2> gpurun_out/w_$w.err || exit 1
  echo "$w done"
done
timeout -k 10 200 python bench.py --workload weekend --shutter-bvh --steps 3 --warmup 1 --cpu-spp 0 > gpurun_out/w_weekend_shutter.json 2> gpurun_out/w_weekend_shutter.err || exit 1
echo "weekend shutter done"
for b in 40 60 100; do
  timeout -k 10 200 python bench.py --boxes-per-side $b --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/w_final_$b.json 2> gpurun_out/w_final_$b.err || exit 1
done
echo "final with larger grids done"
timeout -k 10 250 python bench.py --workload mesh --bvh lbvh --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/w_mesh_lbvh.json 2> gpurun_out/w_mesh_lbvh.err || exit 1
echo "mesh lbvh done"
timeout -k 10 280 python bench.py --workload mesh --bvh widest --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/w_mesh_widest.json 2> gpurun_out/w_mesh_widest.err || exit 1
echo "mesh widest done"
timeout -k 10 280 python bench.py --workload mesh --bvh reference --spp-per-step 16 --steps 1 --warmup 1 --cpu-spp 0 > gpurun_out/w_mesh_reference.json 2> gpurun_out/w_mesh_reference.err || exit 1
echo "mesh reference done"
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bvh -o bvh -- python3 tools/bvh_build_time.py 10000000 > gpurun_out/bvh_build_time.txt 2>&1 || exit 1
echo "bvh build done"
