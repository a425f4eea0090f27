"""Developer tool (GPU box): lane utilisation per phase of the closest-hit loop. Runs bench.py with the default
library (work counters count lanes) and with every gpurun_variants/diag*.so (built with -DRT_DIAG_WAVE: the same
counters count wave-level executions) and prints lanes / (64 * wave executions) per entry kind."""
import glob
import json
import os
import subprocess
import sys


def run(libpath):
    env = dict(os.environ)
    if libpath:
        env["RT_HIP_LIB"] = os.path.abspath(libpath)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--cpu-spp", "0"] + sys.argv[1:], env=env,
                         capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1])["roofline"]


lanes = run(None)["visits_per_ray"]
print("lane visits per ray:", lanes)
for v in sorted(glob.glob("gpurun_variants/diag*.so")):
    r = run(v)
    w = r["visits_per_ray"]
    print(os.path.basename(v), "extend ms", round(r["ms_extend_per_step"], 2))
    for k in w:
        if w[k]:
            print("   %-9s wave executions/ray %.4f  utilisation %.3f" % (k, w[k], lanes[k] / w[k] / 64))
