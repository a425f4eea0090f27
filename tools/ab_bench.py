"""Developer tool (GPU box): run bench.py once per library variant under gpurun_variants/ (and the default build)
and print samples/s, extend and shade time per step. Usage: python tools/ab_bench.py [bench args...]"""
import glob
import json
import os
import subprocess
import sys

variants = [None] + sorted(glob.glob("gpurun_variants/*.so"))
for v in variants:
    env = dict(os.environ)
    if v:
        env["RT_HIP_LIB"] = os.path.abspath(v)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--cpu-spp", "0"] + sys.argv[1:],
                         env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(v, "FAILED", out.stderr[-300:])
        continue
    d = json.loads(line[-1])
    print("%-40s %7.1f Msamples/s  extend %7.2f  shade %7.2f ms/step" % (os.path.basename(v) if v else "default", d["value"] / 1e6,
          d["roofline"]["ms_extend_per_step"], d["roofline"]["ms_shade_per_step"]), flush=True)
