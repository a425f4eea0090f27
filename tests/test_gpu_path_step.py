"""GPU parity of one bounce of the render path itself (rt_path_step): the closest-hit kernel variant and the
shade kernel that rt_render launches, run on fixed ray sets and compared ray by ray with the CPU oracle's
statement of one ray_color iteration (src/color.rs:41-62) — closest hit, emitted light, attenuation and
scattered ray. Every closest-hit kernel profile a scene can legally run is exercised, not only the one the
library would pick.

Bars. Closest hit: primitive id, box side and wrapper id exact, t within 4 ulp (0 ulp where no ConstantMedium
is hit: its t carries a log10), traversal work (box tests, primitive tests by kind) equal to the oracle's.
Shading: scattered / absorbed decisions equal; emitted light exact; attenuation and scattered ray exact where
only + - * / sqrt and the counter-based RNG are involved (Metal, Dielectric, Isotropic, solid and checker
textures), within the stated absolute tolerances after sin / cos (Lambertian direction: 1e-15; noise texture:
1e-12) and after atan2 / asin (image texture look-up: the same texel)."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, SCENE_OPTS, SEED, load_golden, scene_kwargs, ulp_diff, oracle_mode
from raytracer_amd import abi, lib

pytestmark = pytest.mark.gpu

# seeded random scenes per test; RT_FUZZ_CASES widens the campaign for a one-off run (e.g. 100)
FUZZ_CASES = int(os.environ.get("RT_FUZZ_CASES", "12"))

T_ULP = 4
PROFILES = (0, 1, 3, 4, 5, 7)
M_LAMBERTIAN, M_METAL, M_DIELECTRIC, M_LIGHT, M_ISOTROPIC = range(5)
T_SOLID, T_CHECKER, T_NOISE, T_IMAGE = range(4)


def legal_profiles(hs):
    """Profiles able to walk this scene: a medium bounded by a box inside wrappers needs bit 1 or 2, a medium
    bounded by anything but one sphere or one such box needs bit 2 (mirrors the rule stated in include/rt/abi.h)."""
    if hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND:
        return [0, 4]  # the per-axis box rule is built for these two only (and implied)
    nodes = hs.table("nodes")
    kinds = nodes["meta"] & 15
    need_general = need_box = False
    for i in np.nonzero(kinds == 8)[0]:
        link = int(nodes["link"][i])
        if link == i + 2 and kinds[i + 1] == 1:
            continue
        j = i + 1
        while j + 1 < link and kinds[j] == 6:
            j += 1
        if kinds[j] == 4 and link == i + 2 + 2 * (j - (i + 1)):
            need_box = True
        else:
            need_general = True
    out = []
    for pr in PROFILES:
        if need_general and not pr & 2:
            continue
        if need_box and not pr & 3:
            continue
        out.append(pr)
    return out


def material_kinds(hs, hits):
    """Material kind and texture kind of every ray's hit (-1 on a miss), looked up by primitive id in the flattened
    scene's own tables."""
    mats = hs.table("materials")
    texs = hs.table("textures")
    n_prims = 0
    tables = [hs.table(name) for name in ("spheres", "msphere", "rects", "boxes", "triangles", "media")]
    for t in tables:
        if len(t):
            n_prims = max(n_prims, int(t["prim_id"].max()) + 1)
    mat_of = np.zeros(n_prims, dtype=np.int64)
    for t in tables:
        if len(t):
            mat_of[t["prim_id"]] = t["mat"]
    hit = hits["prim_id"] != abi.RT_MISS
    m = mat_of[np.where(hit, hits["prim_id"], 0)]
    mk = np.where(hit, mats["kind"][m], -1)
    textured = np.isin(mk, (M_LAMBERTIAN, M_LIGHT, M_ISOTROPIC))
    if len(texs) == 0:  # a scene whose objects all took the materials without a texture
        return mk, np.full(len(mk), -1)
    tk = np.where(textured, texs["kind"][np.where(textured, mats["tex"][m], 0)], -1)
    return mk, tk


def compare_steps(want, got, mk, tk, what, media):
    assert np.array_equal(want["prim_id"], got["prim_id"]), what + ": primitive ids differ"
    assert np.array_equal(want["sub"], got["sub"]), what + ": box side / wrapper ids differ"
    hit = want["prim_id"] != abi.RT_MISS
    d = np.zeros(len(want), dtype=np.int64)
    d[hit] = ulp_diff(want["t"][hit], got["t"][hit])
    assert d.max(initial=0) <= (T_ULP if media else 0), what + ": t off by %d ulp" % d.max(initial=0)
    # only a medium's t goes through log10 (device math library against the C library): everything else to the bit
    assert d[mk != M_ISOTROPIC].max(initial=0) == 0, what + ": t of a hit that is no medium's off by %d ulp" % d[mk != M_ISOTROPIC].max(initial=0)
    assert np.array_equal(want["scattered"], got["scattered"]), what + ": scatter decisions differ"
    assert np.array_equal(want["emitted"], got["emitted"]), what + ": emitted light differs"
    sc = want["scattered"] == 1
    exact_t = d == 0
    # attenuation: exact for solid / checker / untextured materials, 1e-12 through the noise texture's sin, same texel for images
    plain = sc & (tk != T_NOISE)
    assert np.array_equal(want["attenuation"][plain], got["attenuation"][plain]), what + ": attenuation differs"
    noisy = sc & (tk == T_NOISE)
    assert np.abs(want["attenuation"][noisy] - got["attenuation"][noisy]).max(initial=0) <= 1e-12, what + ": noise texture"
    # origin of the scattered ray = the hit point: exact where t is
    assert np.array_equal(want["o"][sc & exact_t], got["o"][sc & exact_t]), what + ": scattered origin differs"
    rest = sc & ~exact_t
    assert np.allclose(want["o"][rest], got["o"][rest], rtol=1e-13, atol=1e-11)
    # direction: exact for Metal / Dielectric / Isotropic, 1e-15 after the sin / cos of random_unit_vector
    exact_d = sc & exact_t & np.isin(mk, (M_METAL, M_DIELECTRIC, M_ISOTROPIC))
    assert np.array_equal(want["d"][exact_d], got["d"][exact_d]), what + ": scattered direction differs"
    assert np.array_equal(want["d"][sc & (mk == M_ISOTROPIC)], got["d"][sc & (mk == M_ISOTROPIC)])
    assert np.abs(want["d"][sc] - got["d"][sc]).max(initial=0) <= 1e-15 * max(1.0, np.abs(want["d"][sc]).max(initial=0)), what
    return {"rays": len(want), "scattered": int(sc.sum()), "by_material": {int(k): int((sc & (mk == k)).sum()) for k in range(5)},
            "by_texture": {int(k): int((hit & (tk == k)).sum()) for k in range(4)}}


@pytest.mark.parametrize("sid", sorted(SCENE_OPTS))
def test_golden_rays_through_the_render_kernels(gpu, orc, sid):
    g = load_golden(sid)
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        meta = json.load(f)["scenes"][str(sid)]
    hs = lib.HostScene.build(sid, seed=SEED, **scene_kwargs(sid))
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(sid, seed=SEED, **scene_kwargs(sid))
    _, _, ost = orc.trace(osc, g["rays"], seed=SEED, mode=oracle_mode(hs))
    su = gpu.host.scene_setup(sid)
    p = lib.make_params(meta["width"], meta["height"], 0, 4, list(su.background), seed=SEED)
    mk, tk = material_kinds(hs, g["hits"])
    legal = legal_profiles(hs)
    assert (ds.profile() & ~8) in legal  # bit 8, the per-axis box rule, is implied by the scene, not chosen
    for pr in PROFILES:
        if pr not in legal:
            with pytest.raises(lib.RtError):
                ds.path_step(p, g["rays"], profile=pr)
            continue
        got, work = ds.path_step(p, g["rays"], profile=pr, want_work=True)
        compare_steps(g["steps"], got, mk, tk, "scene %d profile %d" % (sid, pr), media=sid in (8, 9))
        assert work["inner"] == ost["box_tests"], (pr, work, ost)
        assert work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"] and work["triangle"] == ost["triangle"]
        assert work["box6"] == ost["box6"] and work["medium"] == ost["medium"]
        assert work["rect"] == ost["rect"] - 6 * ost["box6"]
        again = ds.path_step(p, g["rays"], profile=pr)
        assert again.tobytes() == got.tobytes(), "instrumented and plain kernels disagree"
    ds.close(); hs.close(); orc.api.scene_free(osc)


SEEDED = [(1, {}), (7, {}), (8, {}), (9, {}), (9, {"boxes_per_side": 40}), (11, {}), (12, {})]


@pytest.mark.parametrize("sid,extra", SEEDED)
def test_seeded_rays_through_the_render_kernels(gpu, orc, sid, extra):
    # 30 000+ harvested rays per BASELINE scene (and the final scene with a 40 x 40 ground grid, whose tables are
    # staged in LDS only in part): every legal profile against the oracle, hits, work and shading
    seed = 0xC0FFEE + sid
    kw = dict(scene_kwargs(sid), **extra)
    hs = lib.HostScene.build(sid, seed=seed, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(sid, seed=seed, **kw)
    su = orc.api.scene_setup(sid)
    cam = orc.api.camera(su)
    W = 256
    p = orc.make_params(W, int(W / su.aspect_ratio), 0, 4, list(su.background), seed=seed)
    rays = orc.harvest(osc, cam, p, n_paths=40000, max_rays=1 << 16, max_bounce=3)
    assert len(rays) > 30000
    oh, _, ost = orc.trace(osc, rays, seed=seed, mode=oracle_mode(hs))
    want = orc.path_step(osc, p, rays, mode=oracle_mode(hs))
    mk, tk = material_kinds(hs, oh)
    seen = None
    for pr in legal_profiles(hs):
        got, work = ds.path_step(p, rays, profile=pr, want_work=True)
        seen = compare_steps(want, got, mk, tk, "scene %d profile %d" % (sid, pr), media=sid in (8, 9))
        assert work["inner"] == ost["box_tests"]
        assert work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"] and work["triangle"] == ost["triangle"]
        assert work["box6"] == ost["box6"] and work["medium"] == ost["medium"]
        assert work["rect"] == ost["rect"] - 6 * ost["box6"]
    if sid == 9:  # the final scene must have exercised every material and every texture kind
        assert all(seen["by_material"][k] > 0 for k in (M_LAMBERTIAN, M_METAL, M_DIELECTRIC, M_ISOTROPIC)), seen
        assert all(seen["by_texture"][k] > 0 for k in (T_SOLID, T_NOISE, T_IMAGE)), seen
    if sid == 1:
        assert seen["by_texture"][T_CHECKER] > 0, seen
    ds.close(); hs.close(); orc.api.scene_free(osc)


@pytest.mark.parametrize("case", range(FUZZ_CASES))
def test_random_scenes_through_the_render_kernels(gpu, orc, case):
    from test_gpu_trace import random_rays, random_scene_pair
    rng = np.random.default_rng(4321 + case)
    builders, (hroot, oroot) = random_scene_pair(rng, gpu, orc, n_obj=int(rng.integers(4, 40)))
    hs = lib.HostScene(gpu, hroot)
    ds = lib.DeviceScene(hs)
    n = 20000
    W, H = 256, 128
    rays = random_rays(rng, n)
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    rays["pixel"] = rng.permutation(W * H)[:n]
    rays["sample"] = rng.integers(0, 4, n)
    rays["bounce"] = rng.integers(0, 4, n)
    p = lib.make_params(W, H, 0, 4, [0.3, 0.5, 0.7], seed=77)
    oh, _, ost = orc.trace(oroot, rays, seed=77, mode=oracle_mode(hs))
    want = orc.path_step(oroot, p, rays, mode=oracle_mode(hs))
    mk, tk = material_kinds(hs, oh)
    assert (want["prim_id"] != abi.RT_MISS).sum() > 40
    for pr in legal_profiles(hs):
        got, work = ds.path_step(p, rays, profile=pr, want_work=True)
        compare_steps(want, got, mk, tk, "random scene %d profile %d" % (case, pr), media=True)
        assert work["inner"] == ost["box_tests"] and work["medium"] == ost["medium"]
        assert work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"] and work["triangle"] == ost["triangle"]
    ds.close(); hs.close(); orc.api.scene_free(oroot)


def test_argument_errors(gpu):
    hs = lib.HostScene.build(7)
    ds = lib.DeviceScene(hs)
    p = lib.make_params(16, 16, 0, 2, [0, 0, 0])
    rays = np.zeros(2, dtype=abi.RAY_DTYPE)
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    rays["d"] = [0, 0, 1]
    with pytest.raises(lib.RtError, match="share"):
        ds.path_step(p, rays)
    rays["pixel"] = [0, 256]
    with pytest.raises(lib.RtError, match="outside"):
        ds.path_step(p, rays)
    rays["pixel"] = [0, 1]
    rays["tmax"] = 5.0
    with pytest.raises(lib.RtError, match="tmin"):
        ds.path_step(p, rays)
    rays["tmax"] = np.inf
    with pytest.raises(lib.RtError, match="profile"):
        ds.path_step(p, rays, profile=2)
    out = ds.path_step(p, rays[:0])
    assert len(out) == 0
    ds.close(); hs.close()
