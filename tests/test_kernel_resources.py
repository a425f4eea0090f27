"""The register budget of the hot kernels, read from the compiler's own resource report.

The closest-hit kernel is built to run four waves per SIMD: 128 vector registers (gfx950 shares one file of 512 per
SIMD lane between the waves resident on it) with a few spilled values outside the box loop. A change of a few lines in a
leaf test can push it over: the compiler then either drops to three waves or spills inside the hot loop, and the kernel
loses up to half of its speed without any test of its results failing. The build keeps the compiler's report
(-Rpass-analysis=kernel-resource-usage, raytracer_amd/csrc/build/*.remarks); this test holds every instantiation to the
budget it was tuned for."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def parse_remarks(path):
    """{demangled kernel name without arguments: {field: int}} from a -Rpass-analysis=kernel-resource-usage log."""
    rows, cur = {}, None
    for line in open(path):
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in body:
            k, v = body.rsplit(":", 1)
            rows[cur][k.strip()] = v.strip()
    names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True, check=True).stdout.splitlines()
    out = {}
    for name, r in zip(names, rows.values()):
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        out[name] = {"vgprs": int(r["VGPRs"]), "agprs": int(r["AGPRs"]), "scratch": int(r["ScratchSize [bytes/lane]"]),
                     "occupancy": int(r["Occupancy [waves/SIMD]"]), "vgpr_spills": int(r["VGPRs Spill"]),
                     "dynamic_stack": r["Dynamic Stack"] != "False", "lds": int(r["LDS Size [bytes/block]"])}
    return out


@pytest.fixture(scope="module")
def kernels():
    subprocess.run(["make", "-C", CSRC], check=True, stdout=subprocess.DEVNULL)
    path = os.path.join(CSRC, "build", "rt_hip.remarks")
    assert os.path.exists(path), "the build did not leave the compiler's resource report"
    assert os.path.getmtime(path) >= os.path.getmtime(os.path.join(CSRC, "build", "rt_hip.o")), "stale report"
    return parse_remarks(path)


LEAN = (0, 1, 4, 5, 8, 12)   # extend profiles without the general boundary walk: 16 waves per CU
GENERAL = (3, 7)             # with it: 12 waves per CU


def test_every_extend_profile_is_reported(kernels):
    for profile in LEAN + GENERAL:
        for count in ("false", "true"):
            for lds in ("false", "true"):
                assert "rtd::k_extend<%s, %s, %d>" % (count, lds, profile) in kernels


@pytest.mark.parametrize("lds", ["true", "false"])
@pytest.mark.parametrize("profile", LEAN)
def test_lean_extend_fits_four_waves(kernels, profile, lds):
    k = kernels["rtd::k_extend<false, %s, %d>" % (lds, profile)]
    assert k["occupancy"] == 4 and k["vgprs"] + k["agprs"] <= 128, k
    # what is spilled today: a handful of values of the refill / leaf code, 48 bytes per lane; more than that is a
    # spill that reached the box loop
    assert k["scratch"] <= 48 and k["vgpr_spills"] <= 8 and not k["dynamic_stack"], k
    # the instrumented twin (work counters; it also puts the media that are not in the stream to the ray itself) keeps
    # the occupancy, its counters may spill; it is never timed
    c = kernels["rtd::k_extend<true, %s, %d>" % (lds, profile)]
    assert c["occupancy"] == 4 and c["scratch"] <= 400, c


@pytest.mark.parametrize("lds", ["true", "false"])
@pytest.mark.parametrize("profile", GENERAL)
def test_general_extend_fits_three_waves(kernels, profile, lds):
    k = kernels["rtd::k_extend<false, %s, %d>" % (lds, profile)]
    assert k["occupancy"] == 3 and k["vgprs"] + k["agprs"] <= 168, k
    assert k["scratch"] <= 112 and not k["dynamic_stack"], k


def test_shade_fits_two_waves_without_scratch(kernels):
    for rounds in ("0u", "12u", "20u"):  # no sorting, the two sorting windows
        k = kernels["rtd::k_shade<true, %s>" % rounds]   # tables in LDS: C1 - C4
        assert k["occupancy"] == 2 and k["vgprs"] + k["agprs"] <= 256 and k["scratch"] == 0, k
        assert k["lds"] <= 160 * 1024 - 144 * 1024 + 64 * 1024 + 20 * 1024, k  # static part; the staged tables come on top at launch
        g = kernels["rtd::k_shade<false, %s>" % rounds]  # tables in global memory: C5
        assert g["occupancy"] == 2 and g["scratch"] <= 40, g


def test_streaming_kernels_stay_small(kernels):
    for name in ("rtd::k_accumulate", "rtd::k_resolve"):
        k = kernels[name]
        assert k["scratch"] == 0 and k["vgprs"] <= 64 and k["occupancy"] >= 6, (name, k)
    # k_generate also asks the media that every ray reaches (Philox, log10, two square roots): still bound by its writes
    k = kernels["rtd::k_generate"]
    assert k["scratch"] == 0 and k["vgprs"] <= 128 and k["occupancy"] >= 4, k
