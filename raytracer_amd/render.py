"""Command-line front end: what the reference's main() (src/main.rs:44-241) does, with its edit-and-recompile
choices turned into flags: the scene (`let scene = 11`, :65), the image width (:164, height follows from the scene's
aspect ratio, :165), samples per pixel (:55-179), and the output path (:240, "output/test.png").

    python -m raytracer_amd.render --scene 9 --width 800 --spp 1000 --out final.png
    python -m torch.distributed.run --nproc-per-node 8 -m raytracer_amd.render --scene 9 --spp 10000 --out final.png

One process per GPU: with WORLD_SIZE > 1 every rank renders its own sample range of every pixel and the float
accumulation buffers are summed by one all-reduce (RCCL). `--checkpoint FILE` saves the accumulation buffer every
`--checkpoint-every` samples and resumes from it when it exists (single process only). There is no CPU fallback."""
import argparse
import os
import sys
import time

import numpy as np

from . import abi, assets, imageio, lib
from .distributed import shard_samples

SCENE_NAMES = {1: "random_scene", 2: "random_scene (pinhole)", 3: "simple_light", 4: "light_demo", 5: "two_perlin_spheres",
               6: "earth", 7: "cornell_box", 8: "cornell_smoke", 9: "final_scene", 10: "try_triangle", 11: "kaleidoscope",
               12: "kaleidoscope + synthetic mesh"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd.render", description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", type=int, default=11, choices=sorted(SCENE_NAMES), help="scene index of src/main.rs:65-162 (default: the reference's 11)")
    ap.add_argument("--width", type=int, default=0, help="image width (default: the reference's 800, src/main.rs:164)")
    ap.add_argument("--height", type=int, default=0, help="image height (default: width / the scene's aspect ratio)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (default: the scene's value in src/main.rs)")
    ap.add_argument("--spp-per-batch", type=int, default=0, help="samples of every pixel per wavefront batch (default: sized to the GPU's memory)")
    ap.add_argument("--max-depth", type=int, default=50, help="src/main.rs:49")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--bvh", default="reference", choices=["reference", "widest", "lbvh"],
                    help="BVHNode::new as in the reference (random split axis, src/bvh.rs:24), with the widest-spread axis, or replaced by the linear BVH built on the GPU")
    ap.add_argument("--shutter-bvh", action="store_true", help="random_scene: build the BVH over the shutter interval instead of (0, 0.1) (src/main.rs:315)")
    ap.add_argument("--triangles", type=int, default=1_000_000, help="scene 12: number of synthetic triangles")
    ap.add_argument("--boxes-per-side", type=int, default=0, help="final_scene ground grid (src/main.rs:591)")
    ap.add_argument("--out", default="output/test.png", help="PNG, or PPM when the name ends in .ppm (reference: output/test.png)")
    ap.add_argument("--checkpoint", default="", help="file to save the accumulation buffer to, and to resume from when present")
    ap.add_argument("--checkpoint-every", type=int, default=0, help="samples per pixel between checkpoints (default: at the end only)")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def plan(args, setup):
    """Image size, sample count and the parameters a checkpoint must agree on."""
    width = args.width or int(setup.width)
    height = args.height or int(width / setup.aspect_ratio)  # src/main.rs:165
    spp = args.spp or int(setup.samples_per_pixel)
    if width < 2 or height < 2 or spp < 1:
        raise SystemExit("width and height must be at least 2 and spp at least 1")
    if args.checkpoint and not args.spp_per_batch:
        # the sums depend on where the batches are cut (f32 accumulation per batch); the library's own choice follows
        # the free device memory of the moment, so a checkpointed render pins the batch size instead
        args.spp_per_batch = max(1, min(1024, (512 << 20) // (width * height)))
    meta = {"scene": args.scene, "width": width, "height": height, "seed": args.seed, "max_depth": args.max_depth, "bvh": args.bvh,
            "shutter_bvh": bool(args.shutter_bvh), "triangles": args.triangles if args.scene == 12 else 0,
            "boxes_per_side": args.boxes_per_side, "spp_per_batch": args.spp_per_batch}
    return width, height, spp, meta


def segments(begin, end, every):
    """[begin, end) cut at multiples of `every` counted from 0 (so a resumed run cuts where the first one did)."""
    if every <= 0:
        return [(begin, end)] if end > begin else []
    out = []
    lo = begin
    while lo < end:
        hi = min(end, (lo // every + 1) * every)
        out.append((lo, hi))
        lo = hi
    return out


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    L = lib.get()
    if L.device_count() < 1:
        raise SystemExit("no HIP device: this renderer has no CPU path")
    say = (lambda *a: None) if (args.quiet or rank != 0) else (lambda *a: print(*a, file=sys.stderr, flush=True))
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.checkpoint:
            raise SystemExit("--checkpoint is for single-process renders")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    flags = {"reference": 0, "widest": abi.RT_SCENE_BVH_WIDEST_AXIS, "lbvh": abi.RT_SCENE_BVH_LBVH}[args.bvh]
    flags |= abi.RT_SCENE_BVH_SHUTTER_TIMES if args.shutter_bvh else 0
    kw = dict(seed=args.seed, flags=flags, boxes_per_side=args.boxes_per_side)
    if args.scene in (6, 9):
        kw["image"] = assets.earthmap()
    if args.scene == 12:
        kw["n_triangles"] = args.triangles
    t0 = time.time()
    hs = lib.HostScene.build(args.scene, **kw)
    ds = lib.DeviceScene(hs)
    setup = L.host.scene_setup(args.scene)
    cam_setup = setup
    width, height, spp, meta = plan(args, setup)
    if args.height:
        # an explicit height overrides the scene's aspect ratio; otherwise the camera keeps the nominal one and the
        # height is its truncated quotient, as in src/main.rs:165-179
        cam_setup.aspect_ratio = width / height
    cam = L.host.camera(cam_setup)
    say("scene %d (%s): %d entries, built in %.1f s; %dx%d, %d spp" % (args.scene, SCENE_NAMES[args.scene], hs.desc.n_nodes, time.time() - t0, width, height, spp))

    bg = list(setup.background)
    t0 = time.time()
    if world > 1:
        accum_t = torch.zeros(height, width, 3, dtype=torch.float32, device="cuda")
        lo, hi = shard_samples(0, spp, rank, world)
        if hi > lo:
            p = lib.make_params(width, height, lo, hi, bg, seed=args.seed, max_depth=args.max_depth, samples_per_batch=args.spp_per_batch)
            ds.render_device(cam, p, accum_t.data_ptr(), torch.cuda.current_stream().cuda_stream)
        dist.all_reduce(accum_t)
        torch.cuda.synchronize()
        accum = accum_t.cpu().numpy()
    else:
        accum = np.zeros((height, width, 3), dtype=np.float32)
        done = 0
        if args.checkpoint and os.path.exists(args.checkpoint):
            accum, done = imageio.load_checkpoint(args.checkpoint, meta)
            if accum.shape != (height, width, 3) or done > spp:
                raise SystemExit("checkpoint %s does not fit this render" % args.checkpoint)
            say("resuming from %s at %d spp" % (args.checkpoint, done))
        for lo, hi in segments(done, spp, args.checkpoint_every if args.checkpoint else 0):
            p = lib.make_params(width, height, lo, hi, bg, seed=args.seed, max_depth=args.max_depth, samples_per_batch=args.spp_per_batch)
            accum, st = ds.render(cam, p, accum)
            if args.checkpoint:
                imageio.save_checkpoint(args.checkpoint, accum, hi, meta)
            say("  %d / %d spp, %.1f s, %.1f M rays/s" % (hi, spp, time.time() - t0, st.rays / max(st.ms_total, 1e-9) / 1e3))
    secs = time.time() - t0
    if rank == 0:
        rgb = lib.resolve_u8(accum, spp)
        out_dir = os.path.dirname(args.out)
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        imageio.write_image(args.out, rgb)
        say("wrote %s: %.2f s, %.1f M samples/s" % (args.out, secs, width * height * spp / max(secs, 1e-9) / 1e6))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
