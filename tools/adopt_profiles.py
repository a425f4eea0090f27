"""Copies the summaries of one tools/collect_profiles.sh run from gpurun_out/ into profiles/ and derives the counter
records bench.py reads (tools/pmc_to_json.py) from them.

    python tools/adopt_profiles.py TAG KEY
TAG: the tag given to collect_profiles.sh; KEY: the workload key of bench.py (profile_key: "final", "weekend", "cornell",
"smoke", "mesh_lbvh", ...). The timed closest-hit kernel is the k_extend<false, ...> instantiation of the kernel trace,
its launch time the average over the launches of the run's JSON line (roofline.avg_launch_ms)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_to_json  # noqa: E402


def main(tag, key):
    out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    for part in ("kernel_stats", "pmc_issue", "pmc_mix", "pmc_fetch", "pmc_write"):
        shutil.copy(os.path.join(out, "%s_%s.txt" % (tag, part)), os.path.join(prof, "%s_%s.txt" % (tag, part)))
    line = open(os.path.join(out, "prof_%s_bench.json" % tag)).read().strip().splitlines()[-1]
    d = json.loads(line)
    open(os.path.join(prof, "%s_bench.json" % tag), "w").write(line + "\n")
    stats = open(os.path.join(prof, "%s_kernel_stats.txt" % tag)).read()
    extend = re.search(r"(rtd::k_extend<false, [^>]*>)", stats).group(1)
    shade = re.search(r"(rtd::k_shade<[^>]*>)", stats).group(1)
    r = d["roofline"]
    os.environ["SHADE_KERNEL"] = shade
    rays_per_launch = None if r["tables_in_lds"] else d["mrays_per_s"] * 1e6 * d["ms_per_step"] * 1e-3 / 50.0
    spp = d["config"]["spp_per_step_per_gpu"]
    args = [key, tag, extend, spp, r["avg_launch_ms"] * 1e-3] + ([rays_per_launch] if rays_per_launch else [])
    pmc_to_json.main(*args)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
