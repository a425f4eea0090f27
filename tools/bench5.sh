#!/bin/bash
# GPU box: the five BASELINE workloads with the library as built, one short bench.py run each (4-spp CPU comparison
# for the image check where it finishes quickly). Lines land in gpurun_out/b5_<workload>.json.
#     bash tools/bench5.sh [extra bench.py arguments]
for w in final cornell smoke weekend; do
  timeout -k 10 250 python bench.py --workload $w --steps 3 --cpu-spp 4 "$@" > gpurun_out/b5_$w.json 2> gpurun_out/b5_$w.err || exit 1
done
timeout -k 10 280 python bench.py --workload mesh --bvh lbvh --steps 3 --cpu-spp 0 "$@" > gpurun_out/b5_mesh.json 2> gpurun_out/b5_mesh.err || exit 1
python - <<'PY'
import json
for t in ("final", "cornell", "smoke", "weekend", "mesh"):
    d = json.loads(open("gpurun_out/b5_%s.json" % t).read().strip().splitlines()[-1]); r = d["roofline"]
    print("%-8s %7.1f M samples/s  extend %7.1f  shade %6.1f ms/step  entries/ray %.2f  rmse vs CPU %s  pixels off %s" % (
        t, d["value"] / 1e6, r["ms_extend_per_step"], r["ms_shade_per_step"], r["entries_per_ray"],
        d.get("rmse_vs_cpu", {}).get("value"), d.get("rmse_vs_cpu", {}).get("pixels_off")))
PY
