"""Turns the counter summaries of one tools/collect_profiles.sh run into the two small JSON files bench.py reads for
its roofline object: profiles/hbm_traffic.json (HBM bytes per closest-hit launch) and profiles/valu_issue.json (vector
issue and lane utilisation of the same kernel).

    python tools/pmc_to_json.py KEY TAG KERNEL SPP SECONDS_PER_LAUNCH [RAYS_PER_LAUNCH]
KEY: workload name as bench.py knows it, with the tree policy appended when not the reference's ("mesh_lbvh");
TAG: the tag the summaries were written under (profiles/TAG_pmc_fetch.txt ...); KERNEL: e.g. "rtd::k_extend<false, true, 4>";
SPP: samples per step of the profiled run; SECONDS_PER_LAUNCH: the kernel's average duration from the kernel trace;
RAYS_PER_LAUNCH: give it for a scene whose tables live in HBM. FETCH_SIZE tallies the wide in-order reads of the ray
records at half their bytes and scattered 32-byte entry reads at the 64 bytes they fetch (tools/fetch_calibrate.hip), so
for a scene in LDS (reads = records only) FETCH_SIZE is doubled, and for a scene in HBM only the records' missing half,
48 bytes per ray, is added.
Environment: SHADE_KERNEL (e.g. "rtd::k_shade<true, 20u>") and SHADE_SECONDS_PER_STEP also record the shade launches of
one step under KEY + "/shade": the sorting variant gathers whole records through a permutation, which FETCH_SIZE tallies
at 0.89 of the bytes (tools/fetch_calibrate.hip), the plain variant streams them in order (tallied at half).
Every record carries a hash of the kernel sources it was measured with; bench.py ignores records of other sources."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from raytracer_amd.lib import kernel_source_hash  # noqa: E402


def counters(path, kernel):
    out, on = {}, False
    for line in open(path):
        if line.startswith(kernel + "  ("):
            on = True
            out["_dispatches"] = int(line.split("(")[-1].split()[0])
            continue
        if on:
            if not line.strip():
                break
            parts = line.split()
            if len(parts) == 2 and parts[0].isupper():
                out[parts[0]] = float(parts[1])
    return out


def main(key, tag, kernel, spp, secs, rays_per_launch=None):
    prof = os.path.join(ROOT, "profiles")
    f = counters(os.path.join(prof, tag + "_pmc_fetch.txt"), kernel)
    w = counters(os.path.join(prof, tag + "_pmc_write.txt"), kernel)
    s = counters(os.path.join(prof, tag + "_pmc_issue.txt"), kernel)
    n = f["_dispatches"]
    raw = (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0 / n
    doubled = (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0 / n
    path = os.path.join(prof, "hbm_traffic.json")
    t = json.load(open(path)) if os.path.exists(path) else {}
    if rays_per_launch is None:
        corrected = doubled
        how = ("FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 bytes; this kernel's HBM reads are the in-order ray "
               "records, the calibrated case); WRITE_SIZE as reported")
    else:
        corrected = raw + 48.0 * float(rays_per_launch)
        how = ("FETCH_SIZE as reported (scattered 32-byte entry reads are tallied at the 64 bytes they fetch, tools/fetch_calibrate.hip) "
               "plus 48 bytes per ray for the in-order ray records, which are tallied at half; WRITE_SIZE as reported")
    t[key] = {"kernel": kernel, "launches": n, "fetch_size_kib": f["FETCH_SIZE"], "write_size_kib": w["WRITE_SIZE"],
              "bytes_per_launch": corrected, "bytes_per_launch_uncorrected": raw,
              "correction": how,
              "source": "profiles/%s_pmc_fetch.txt and profiles/%s_pmc_write.txt (rocprofv3 --pmc, separate passes)" % (tag, tag),
              "spp_per_step": int(spp), "kernel_sources": kernel_source_hash()}
    shade = os.environ.get("SHADE_KERNEL")
    if shade:
        sf = counters(os.path.join(prof, tag + "_pmc_fetch.txt"), shade)
        sw = counters(os.path.join(prof, tag + "_pmc_write.txt"), shade)
        steps = sf["_dispatches"] / 50.0  # 50 launches per step (the timed step and the instrumented pass of bench.py)
        sorting = shade.rstrip(">").split(",")[-1].strip() not in ("0u", "0", "false")  # k_shade<LDS, records per thread of the sorting window>
        factor = 1.0 / 0.89 if sorting else 2.0
        t[key + "/shade"] = {"kernel": shade, "launches": sf["_dispatches"], "fetch_size_kib": sf["FETCH_SIZE"], "write_size_kib": sw["WRITE_SIZE"],
                             "bytes_per_step": (factor * sf["FETCH_SIZE"] + sw["WRITE_SIZE"]) * 1024.0 / steps,
                             "bytes_per_step_uncorrected": (sf["FETCH_SIZE"] + sw["WRITE_SIZE"]) * 1024.0 / steps,
                             "correction": ("FETCH_SIZE / 0.89: the gather of 96-byte records within sorting windows is tallied at 0.89 of its bytes "
                                            "(tools/fetch_calibrate.hip, profiles/r02_fetch_calibrate_*)" if sorting else
                                            "FETCH_SIZE doubled: in-order reads of the records are tallied at half") + "; WRITE_SIZE as reported",
                             "source": "profiles/%s_pmc_fetch.txt and profiles/%s_pmc_write.txt" % (tag, tag),
                             "spp_per_step": int(spp), "kernel_sources": kernel_source_hash()}
    json.dump(t, open(path, "w"), indent=1)
    lanes = s["SQ_THREAD_CYCLES_VALU"] / (s["SQ_ACTIVE_INST_VALU"] * 64.0)
    lane_ops = s["SQ_INSTS_VALU"] * 64.0 * lanes / (s["_dispatches"] * float(secs))
    path = os.path.join(prof, "valu_issue.json")
    v = json.load(open(path)) if os.path.exists(path) else {}
    v[key] = {"kernel": kernel, "active_lane_fraction": lanes,
              "valu_active_per_wave_cycle": s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"],
              "wave_cycles_waiting_on_memory_or_barrier": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"],
              "lane_ops_per_s": lane_ops, "lane_op_peak_per_s": 39.3e12, "lane_op_frac": lane_ops / 39.3e12,
              "note": "lane_ops = SQ_INSTS_VALU x 64 x active-lane fraction / kernel time; peak = 256 CUs x 64 f64 lanes x 2.4 GHz (all vector instructions priced as f64)",
              "source": "profiles/%s_pmc_issue.txt" % tag, "kernel_sources": kernel_source_hash()}
    if shade:
        ss = counters(os.path.join(prof, tag + "_pmc_issue.txt"), shade)
        sl = ss["SQ_THREAD_CYCLES_VALU"] / (ss["SQ_ACTIVE_INST_VALU"] * 64.0)
        v[key + "/shade"] = {"kernel": shade, "active_lane_fraction": sl,
                             "valu_active_per_wave_cycle": ss["SQ_ACTIVE_INST_VALU"] / ss["SQ_WAVE_CYCLES"],
                             "wave_cycles_waiting_on_memory_or_barrier": ss["SQ_WAIT_ANY"] / ss["SQ_WAVE_CYCLES"],
                             "wave_instructions_per_step": ss["SQ_INSTS_VALU"] / (ss["_dispatches"] / 50.0),
                             "source": "profiles/%s_pmc_issue.txt" % tag, "kernel_sources": kernel_source_hash()}
    json.dump(v, open(path, "w"), indent=1)
    print(json.dumps(t[key], indent=1))
    print(json.dumps(v[key], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:7])
