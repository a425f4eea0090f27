"""Image output, checkpoints and the command-line front end (raytracer_amd/render.py, raytracer_amd/imageio.py)."""
import os

import numpy as np
import pytest

from raytracer_amd import imageio, render


def test_png_round_trip_and_known_header(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    data = imageio.encode_png(img)
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[12:16] == b"IHDR" and data[-8:-4] == b"IEND"
    assert np.array_equal(imageio.decode_png(data), img)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    path = str(tmp_path / "a.png")
    imageio.write_image(path, img)
    if Image is not None:  # an independent decoder agrees
        assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), img)
    with pytest.raises(ValueError):
        imageio.encode_png(img.astype(np.float32))
    bad = bytearray(data)
    bad[40] ^= 1
    with pytest.raises(ValueError):
        imageio.decode_png(bytes(bad))


def test_ppm(tmp_path):
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    path = str(tmp_path / "a.ppm")
    imageio.write_image(path, img)
    assert open(path, "rb").read() == b"P6\n3 2\n255\n" + img.tobytes()


def test_checkpoint_round_trip_and_parameter_check(tmp_path):
    acc = np.random.default_rng(1).random((5, 7, 3)).astype(np.float32)
    meta = {"scene": 7, "width": 7, "height": 5, "seed": 1}
    path = str(tmp_path / "c.npz")
    imageio.save_checkpoint(path, acc, 12, meta)
    a2, done = imageio.load_checkpoint(path, dict(meta))
    assert done == 12 and np.array_equal(a2, acc)
    with pytest.raises(ValueError, match="seed"):
        imageio.load_checkpoint(path, dict(meta, seed=2))


def test_segments_cut_at_fixed_multiples():
    assert render.segments(0, 10, 0) == [(0, 10)]
    assert render.segments(0, 10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert render.segments(6, 10, 4) == [(6, 8), (8, 10)]  # a resumed run cuts where the first one did
    assert render.segments(10, 10, 4) == []


def test_plan_defaults_follow_the_reference(rtlib):
    args = render.parse_args([])
    assert args.scene == 11 and args.out == "output/test.png" and args.max_depth == 50  # src/main.rs:65, :240, :49
    su = rtlib.host.scene_setup(9)
    w, h, spp, meta = render.plan(render.parse_args(["--scene", "9"]), su)
    assert (w, h) == (800, 800) and spp == su.samples_per_pixel  # src/main.rs:164-165, :131-139
    su1 = rtlib.host.scene_setup(1)
    w, h, _, _ = render.plan(render.parse_args(["--scene", "1", "--width", "1200"]), su1)
    assert (w, h) == (1200, 800)
    with pytest.raises(SystemExit):
        render.plan(render.parse_args(["--width", "1"]), su)


@pytest.mark.gpu
def test_cli_render_checkpoint_resume_is_bit_identical(gpu, tmp_path):
    common = ["--scene", "7", "--width", "96", "--seed", "0x1234", "--spp-per-batch", "4", "--quiet"]
    whole = str(tmp_path / "whole.png")
    assert render.main(common + ["--spp", "12", "--out", whole]) == 0
    ck = str(tmp_path / "ck.npz")
    part = str(tmp_path / "part.png")
    assert render.main(common + ["--spp", "8", "--out", part, "--checkpoint", ck, "--checkpoint-every", "4"]) == 0
    resumed = str(tmp_path / "resumed.ppm")
    assert render.main(common + ["--spp", "12", "--out", resumed, "--checkpoint", ck, "--checkpoint-every", "4"]) == 0
    a = imageio.decode_png(open(whole, "rb").read())
    assert a.shape == (96, 96, 3) and a.std() > 5
    ppm = open(resumed, "rb").read()
    assert ppm.startswith(b"P6\n96 96\n255\n")
    b = np.frombuffer(ppm[len(b"P6\n96 96\n255\n"):], dtype=np.uint8).reshape(96, 96, 3)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, imageio.decode_png(open(part, "rb").read()))
    with pytest.raises(ValueError):  # other parameters: the checkpoint is refused
        render.main(common[:2] + ["--width", "64", "--spp", "12", "--out", part, "--checkpoint", ck, "--quiet"])


@pytest.mark.gpu
def test_cli_tree_options_give_the_same_picture(gpu, tmp_path):
    # the mesh scene with the reference's tree, the widest-axis tree and the tree built on the GPU: the same samples
    # hit the same primitives whatever the tree (up to exact ties in t), so the pictures agree pixel for pixel but for
    # a handful of pixels
    common = ["--scene", "12", "--triangles", "20000", "--width", "160", "--height", "90", "--spp", "8", "--quiet"]
    pics = {}
    for tree in ("reference", "widest", "lbvh"):
        out = str(tmp_path / (tree + ".png"))
        assert render.main(common + ["--bvh", tree, "--out", out]) == 0
        pics[tree] = imageio.decode_png(open(out, "rb").read()).astype(np.int32)
        assert pics[tree].shape == (90, 160, 3)
    for tree in ("widest", "lbvh"):
        differs = (pics[tree] != pics["reference"]).any(axis=2)
        assert differs.mean() < 0.01, "%s: %d pixels differ" % (tree, differs.sum())


def test_checkpoint_write_is_atomic(tmp_path, monkeypatch):
    # a save that dies half way must leave the previous checkpoint readable
    path = str(tmp_path / "ck.npz")
    meta = {"scene": 7, "seed": 1}
    first = np.full((4, 5, 3), 2.0, dtype=np.float32)
    imageio.save_checkpoint(path, first, 8, meta)

    def dying_savez(f, **kw):
        f.write(b"PK\x03\x04 truncated")
        raise KeyboardInterrupt

    monkeypatch.setattr(np, "savez", dying_savez)
    with pytest.raises(KeyboardInterrupt):
        imageio.save_checkpoint(path, first * 2, 16, meta)
    monkeypatch.undo()
    accum, done = imageio.load_checkpoint(path, meta)
    assert done == 8 and np.array_equal(accum, first)


def test_checkpointed_renders_pin_the_batch_size(rtlib):
    su = rtlib.host.scene_setup(9)
    args = render.parse_args(["--scene", "9", "--checkpoint", "x.npz"])
    _, _, _, meta = render.plan(args, su)
    assert meta["spp_per_batch"] == 838 == (512 << 20) // (800 * 800)  # fixed by the image size, not by free memory
    args = render.parse_args(["--scene", "9"])
    assert render.plan(args, su)[3]["spp_per_batch"] == 0  # without a checkpoint the library sizes batches itself
