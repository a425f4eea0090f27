#!/bin/bash
# GPU box: the four LDS-resident BASELINE workloads with the library as built and with an environment knob set,
# side by side.     bash tools/ab_workloads.sh KNOB=VALUE [workloads...]
knob=$1; shift
for w in ${@:-final cornell smoke weekend}; do
  timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/ab_a_$w.json 2>/dev/null || exit 1
  env $knob timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-spp 0 > gpurun_out/ab_b_$w.json 2>/dev/null || exit 1
  python - $w "$knob" <<'PY'
import json, sys
w, knob = sys.argv[1], sys.argv[2]
for t, name in (("a", "as built"), ("b", knob)):
    d = json.load(open("gpurun_out/ab_%s_%s.json" % (t, w))); r = d["roofline"]
    print("%-8s %-22s %7.1f M samples/s  extend %6.1f  shade %6.1f  entries/ray %.2f" % (w, name, d["value"] / 1e6, r["ms_extend_per_step"], r["ms_shade_per_step"], r["entries_per_ray"]))
PY
done
