#!/bin/bash
# GPU box: A/B timing of library builds (developer tool). Every argument names a library under raytracer_amd/exp/
# (librt_<name>.so; "base" = the library as built); one short bench.py run each, one summary line each.
#     bash tools/ab.sh base NAME [NAME ...]      (WORKLOAD=final by default; extra bench arguments in BENCH_ARGS)
w=${WORKLOAD:-final}
for v in "$@"; do
  if [ "$v" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$GRAFT_REPO_ROOT/raytracer_amd/exp/librt_$v.so; fi
  timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --cpu-spp 0 $BENCH_ARGS > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit 1
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open("gpurun_out/ab_%s.json" % v).read().strip().splitlines()[-1]); r = d["roofline"]
print("%-16s %7.1f M samples/s  %7.1f ms/step  extend %7.1f  shade %6.1f" % (v, d["value"] / 1e6, d["ms_per_step"], r["ms_extend_per_step"], r["ms_shade_per_step"]))
PY
done
