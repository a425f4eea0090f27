for w in final cornell smoke weekend; do
  timeout -k 10 250 python bench.py --workload $w --steps 3 --cpu-spp 4 > gpurun_out/f32_$w.json 2>gpurun_out/f32_$w.err || exit 1
done
timeout -k 10 280 python bench.py --workload mesh --bvh lbvh --steps 3 --cpu-spp 0 > gpurun_out/f32_mesh.json 2>gpurun_out/f32_mesh.err || exit 1
