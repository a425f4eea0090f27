"""Known-answer tests that pin the CPU oracle: the reference's own 21 Vec3 tests
(src/vec3.rs:302-440), the resolve known answer (src/main.rs:153,216-224), the published
Philox4x32-10 vectors, and loose statistics of the reference's example render."""
import json
import math
import os

import numpy as np

from conftest import GOLDEN

ADD, ADD_S, SUB, SUB_S, DOT, MUL_ASSIGN, MUL_S, DIV_S, ELEMUL, CROSS, NEG, SQLEN, LEN, UNIT = range(14)


def v(orc, op, a, b=None, s=0.0):
    rc, out = orc.vec3_op(op, a, b, s)
    assert rc == 0
    return out


def test_new(orc):  # test_new: construction round-trips (checked through add of zero)
    assert v(orc, ADD, [1.0, 2.0, 3.0], [0.0, 0.0, 0.0]) == [1.0, 2.0, 3.0]


def test_add_and_add_assign(orc):  # test_add, test_add_assign
    assert v(orc, ADD, [1.0, 0.0, -1.0], [2.0, 4.0, 6.0]) == [3.0, 4.0, 5.0]


def test_add_f64_and_assign(orc):  # test_add_f64, test_add_assign_f64
    assert v(orc, ADD_S, [1.0, 0.0, -1.0], s=233.0) == [234.0, 233.0, 232.0]


def test_sub_and_sub_assign(orc):  # test_sub, test_sub_assign
    assert v(orc, SUB, [1.0, 0.0, -1.0], [2.0, 4.0, 6.0]) == [-1.0, -4.0, -7.0]


def test_sub_f64_and_assign(orc):  # test_sub_f64, test_sub_assign_f64
    assert v(orc, SUB_S, [1.0, 0.0, -1.0], s=1.0) == [0.0, -1.0, -2.0]


def test_mul_is_dot(orc):  # test_mul: Vec3 * Vec3 is the dot product
    assert v(orc, DOT, [1.0, 0.0, -1.0], [1.0, 1.0, 1.0])[0] == 0.0


def test_mul_assign_is_elementwise(orc):  # test_mul_assign
    assert v(orc, MUL_ASSIGN, [1.0, 0.0, -1.0], [1.0, 1.0, 1.0]) == [1.0, 0.0, -1.0]


def test_mul_f64_and_assign(orc):  # test_mul_f64, test_mul_assign_f64
    assert v(orc, MUL_S, [1.0, 0.0, -1.0], s=1.0) == [1.0, 0.0, -1.0]


def test_div(orc):  # test_div
    assert v(orc, DIV_S, [1.0, -2.0, 0.0], s=2.0) == [0.5, -1.0, 0.0]


def test_elemul(orc):  # test_elemul
    assert v(orc, ELEMUL, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) == [1.0, 4.0, 9.0]


def test_cross(orc):  # test_cross
    assert v(orc, CROSS, [1.0, 2.0, 3.0], [2.0, 3.0, 4.0]) == [8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0]


def test_neg(orc):  # test_neg
    assert v(orc, NEG, [1.0, -2.0, 3.0]) == [-1.0, 2.0, -3.0]


def test_squared_length(orc):  # test_squared_length
    assert v(orc, SQLEN, [1.0, 2.0, 3.0])[0] == 14.0


def test_length(orc):  # test_length
    assert v(orc, LEN, [3.0, 4.0, 5.0])[0] == math.sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0)


def test_unit(orc):  # test_unit
    assert v(orc, UNIT, [233.0, 0.0, 0.0]) == [1.0, 0.0, 0.0]
    assert v(orc, UNIT, [-233.0, 0.0, 0.0]) == [-1.0, 0.0, 0.0]


def test_unit_panic(orc):  # test_unit_panic: unit() of the zero vector must fail
    rc, _ = orc.vec3_op(UNIT, [0.0, 0.0, 0.0])
    assert rc == -1


def test_resolve_known_answer(orc):
    # background (0.52, 0.80, 0.92) (src/main.rs:153) -> sqrt, clamp(0, 0.999) * 255 as u8 = [183, 228, 244],
    # the colour of all four corners and the centre pixel of the reference's img/kaleidoscope.png
    acc = np.zeros((2, 2, 3))
    acc[:] = np.array([0.52, 0.80, 0.92]) * 7
    out = orc.resolve_u8(acc, 7)
    assert (out == np.array([183, 228, 244], dtype=np.uint8)).all()
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        meta = json.load(f)
    assert meta["kaleidoscope_png"]["corner_rgb"] == [183, 228, 244]
    assert meta["kaleidoscope_png"]["centre_rgb"] == [183, 228, 244]


def test_resolve_saturation(orc):
    acc = np.array([[[4.0, -1.0, float("nan")], [0.0, 1.0, 0.999 ** 2]]])
    out = orc.resolve_u8(acc, 1)
    # > 1 clamps to 0.999 -> 254; negative -> sqrt NaN -> 0; NaN -> 0 (Rust float->int casts saturate, NaN -> 0)
    assert out[0, 0].tolist() == [254, 0, 0]
    assert out[0, 1].tolist() == [0, 254, 254]


def test_philox_published_vectors(orc):
    # Random123 kat_vectors, philox4x32 with 10 rounds
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_unit_interval_mapping(orc):
    # draw k of a stream = 53 high bits of the k-th 64-bit word of the Philox blocks, in [0, 1)
    seed, a, b, c = 0x5EED, 12, 34, (5 << 16) | 3
    for k in range(6):
        blk = orc.philox([a, b, c, k >> 1], [seed & 0xFFFFFFFF, seed >> 32])
        j = k & 1
        u64 = (blk[2 * j + 1] << 32) | blk[2 * j]
        assert orc.rand_unit(seed, a, b, c, k) == (u64 >> 11) * 2.0 ** -53
    xs = np.array([orc.rand_unit(1, 0, 0, 0, k) for k in range(4000)])
    assert xs.min() >= 0.0 and xs.max() < 1.0 and abs(xs.mean() - 0.5) < 0.02


def box_down(a, f):
    h, w, _ = a.shape
    return a.reshape(h // f, f, w // f, f, 3).mean((1, 3))


def test_whole_render_against_the_reference_picture(orc):
    # The one output of the Rust binary the reference holds: img/kaleidoscope.png, its scene 11 at 800x800
    # (README.md:11; decoded pixels in tests/golden/kaleidoscope_reference_800.npz). The oracle's render of the same
    # scene, camera and size must be that picture up to Monte-Carlo noise. This pins, end to end, Camera::new /
    # get_ray and the row convention of src/main.rs:206-208, Box6, RotateY, Translate, Triangle, Metal (fuzz 0.01),
    # the background term, the depth-50 recursion and the sqrt / clamp / u8 resolve. Measured: RMSE 3.1 levels per
    # pixel at 8 spp, 0.83 after 4x4 averaging, mean colour within 0.03 levels.
    ref = np.load(os.path.join(GOLDEN, "kaleidoscope_reference_800.npz"))["rgb"].astype(np.float64)
    sc = orc.api.scene_build(11)
    su = orc.api.scene_setup(11)
    cam = orc.api.camera(su)
    spp = 8
    acc, _, _ = orc.render(sc, cam, orc.make_params(800, 800, 0, spp, list(su.background)), mode=0, threads=8)
    img = orc.resolve_u8(acc, spp).astype(np.float64)
    orc.api.scene_free(sc)
    d4 = box_down(img, 4) - box_down(ref, 4)
    assert np.sqrt((d4 ** 2).mean()) < 1.5
    assert np.abs((img - ref).reshape(-1, 3).mean(0)).max() < 0.3
    # the picture is not the render shifted by a row or a column (src/main.rs:206-208)
    rmse = lambda a, b: float(np.sqrt(((a - b) ** 2).mean()))
    here = rmse(img[8:-8, 8:-8], ref[8:-8, 8:-8])
    assert here < 4.5
    for axis in (0, 1):
        for shift in (-1, 1):
            assert rmse(np.roll(img, shift, axis=axis)[8:-8, 8:-8], ref[8:-8, 8:-8]) > here + 0.5


def test_kaleidoscope_statistics(orc):
    # Loose end-to-end sanity against the reference's example picture of scene 11 (800x800, 256 spp, README.md:11):
    # the oracle at 160x160x16 must land within a few levels of its mean colour. Stochastic, hence loose.
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        ref = json.load(f)["kaleidoscope_png"]
    sc = orc.api.scene_build(11)
    su = orc.api.scene_setup(11)
    cam = orc.api.camera(su)
    p = orc.make_params(160, 160, 0, 16, list(su.background))
    acc, _, _ = orc.render(sc, cam, p, mode=0, threads=8)
    img = orc.resolve_u8(acc, 16).astype(np.float64)
    mean = img.reshape(-1, 3).mean(0)
    assert np.abs(mean - np.array(ref["mean_rgb"])).max() < 6.0, (mean, ref["mean_rgb"])
    assert img.max() <= ref["max"]
    orc.api.scene_free(sc)
