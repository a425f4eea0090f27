"""Headline benchmark: path samples/s of the wavefront path tracer on the Next Week final scene
(BASELINE.json configs[3]: 800x800, the configuration the >= 1 Gpath-samples/s target is quoted on; it
fits one GPU).

    python bench.py --gpus N --steps K --warmup W

A step is one wavefront batch: `--spp-per-step` samples of every pixel (default: 1024, or what fits 672 Mi path slots on larger images;
final scene: 671 M path slots, 156 GB of the 288 GB; the batch is large because every bounce's launch lasts at least as long as its
longest ray, however few rays are left) (generate, then up to 50 bounces of
extend + shade, then accumulate) rendered by every rank into its device-resident accumulation buffer; scene
tables, path state and image never leave HBM inside the timed region. With N > 1 (one process per GPU,
torch.distributed over RCCL) every rank renders its own sample range of every pixel (weak scaling: per-GPU
work per step is fixed) and one all-reduce of the float accumulation buffers closes the timed region.

`--scaling strong --spp T` renders a fixed job instead (T samples of every pixel in all, BASELINE configs[2] and [3]
are stated this way): a step is the whole job, rank r renders its share of the T samples in equal batches (no
runt batch), so per-GPU work shrinks as N grows.

Started without torch.distributed.run (no WORLD_SIZE in the environment) with --gpus N > 1, the script starts the N
ranks itself: it launches `python -m torch.distributed.run --nproc-per-node N` on this file as a child process, before
anything in this process has touched the GPU, relays the child's output and exits with its code.

Rank 0 prints one JSON line: whole-job samples/s, the roofline of the dominant kernel (extend: BVH closest
hit) from HIP-event timings taken inside the timed region and algorithmic bytes counted by an instrumented
pass outside it, the CPU oracle timed on the host cores on a bounded sample of the same workload, and the RMSE between
the GPU's and the oracle's pictures of that sample."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import abi, assets, lib  # noqa: E402
from raytracer_amd.distributed import allreduce_accum, equal_batches, shard_samples  # noqa: E402

F64_PEAK_TFLOPS = 78.6  # MI355X vector f64 peak (FMA counted as two)
# Algorithmic f64 operations of the closest-hit walk, per visit: a box test is 6 fma + 6 max / min + the compare; the
# primitive tests as the reference writes them (sphere src/hittable.rs:72-113, rect src/aarect.rs:30-48, a box of six
# sides = 6 rects, triangle src/triangle.rs:26-69); wrappers rotate / translate the ray; a medium adds sqrt, log10
# and the interval arithmetic of src/constant_medium.rs:52-73.
FLOPS = {"inner": 30, "sphere": 25, "msphere": 40, "rect": 15, "box6": 90, "triangle": 45, "xf_enter": 12, "xf_exit": 12, "medium": 40}

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
# Algorithmic bytes of one closest-hit query: 64 (ray in) + 32 per stream entry fetched + the f64 row of every
# primitive tested + 32 (hit out). Rows: sphere 40, moving sphere 72, rect 48, box 56, triangle 80 bytes
# (geometry + material id); wrapper and medium entries are charged their 32-byte entry only.
ROW_BYTES = {"inner": 0, "sphere": 40, "msphere": 72, "rect": 48, "box6": 56, "triangle": 80, "xf_enter": 0, "xf_exit": 0,
             "medium": 0}
RAY_IN_BYTES = 64
HIT_OUT_BYTES = 32

WORKLOADS = {
    "final": dict(scene=9, width=800, height=800, name="Next Week final scene 800x800 (BASELINE configs[3])"),
    "cornell": dict(scene=7, width=600, height=600, name="Cornell box 600x600 (BASELINE configs[1])"),
    "smoke": dict(scene=8, width=600, height=600, name="Cornell smoke 600x600 (BASELINE configs[2])"),
    "weekend": dict(scene=1, width=1200, height=800, name="Weekend random spheres 1200x800 (BASELINE configs[0])"),
    "mesh": dict(scene=12, width=3840, height=2160, n_triangles=10_000_000,
                 name="Kaleidoscope + 10M synthetic triangles 3840x2160 (BASELINE configs[4])"),
}


def scene_kwargs(wl, bvh="reference"):
    """Options of the scene builder; a workload dict may carry boxes_per_side (--boxes-per-side)."""
    kw = {"flags": {"widest": abi.RT_SCENE_BVH_WIDEST_AXIS, "lbvh": abi.RT_SCENE_BVH_LBVH}[bvh]} if bvh != "reference" else {}
    if wl["scene"] in (6, 9):
        kw["image"] = assets.earthmap()
    if "n_triangles" in wl:
        kw["n_triangles"] = wl["n_triangles"]
    if wl.get("boxes_per_side"):
        kw["boxes_per_side"] = wl["boxes_per_side"]
    if wl.get("shutter_bvh"):
        kw["flags"] = kw.get("flags", 0) | abi.RT_SCENE_BVH_SHUTTER_TIMES
    return kw


def profile_key(workload, bvh, wl=None):
    """Key of a committed counter measurement: workload, tree policy and whatever else changes the scene."""
    key = workload if bvh == "reference" else "%s_%s" % (workload, bvh)
    if wl and wl.get("boxes_per_side"):
        key += "_boxes%d" % wl["boxes_per_side"]
    if wl and wl.get("shutter_bvh"):
        key += "_shutter"
    return key


def measured_traffic(workload, spp, field="bytes_per_launch"):
    """HBM bytes per extend launch (or, for key "<workload>/shade", per step of shade launches) from the committed
    rocprofv3 counter passes (profiles/hbm_traffic.json), or None when there is no measurement for this workload, tree
    policy and batch size, or when the kernels have changed since it was taken (the record carries a hash of the kernel
    sources): counters cannot be read from inside a run."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path)).get(workload)
    if not t or t.get("spp_per_step") != spp or t.get("kernel_sources") != lib.kernel_source_hash():
        return None
    return t[field]


def measured_valu(workload):
    """What the committed SQ counter passes say about a kernel (profiles/valu_issue.json): share of issue cycles with a
    vector instruction, active-lane fraction, f64 lane-operations per second; None if absent or taken with other
    kernel sources than the current ones."""
    path = os.path.join(ROOT, "profiles", "valu_issue.json")
    if not os.path.exists(path):
        return None
    v = json.load(open(path)).get(workload)
    if not v or v.get("kernel_sources") != lib.kernel_source_hash():
        return None
    return v


def algorithmic_flops(work):
    return sum(FLOPS[k] * work[k] for k in FLOPS)


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child job and relay it. Nothing in this
    process has initialised the GPU yet (torch is imported, no device call made), and the child is a new process,
    not a replacement of this one."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def algorithmic_bytes(rays, work):
    return rays * (RAY_IN_BYTES + HIT_OUT_BYTES) + 32 * work["entries"] + sum(ROW_BYTES[k] * work[k] for k in ROW_BYTES)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, spp, threads, more_threads=0, bvh="reference"):
    """The CPU oracle (oracle/, a C++ restatement of the reference) on the same scene and image size, `spp`
    samples per pixel, `threads` std::threads splitting columns as src/main.rs:193-195 does; when `more_threads`
    is given, a second run with that many threads (the share of the host's cores this job may use) is reported too.
    Returns the baseline record, the picture (sum over the samples, f64 [H, W, 3]) and the noise floor at this sample
    count: the RMSE between two oracle renders that differ only in the seed, on resolved [0, 1] values, taken at a
    quarter of the image size."""
    import oracle
    o = oracle.get()
    kw = scene_kwargs(wl, bvh)  # the same tree policy as the GPU run (the oracle builds the linear tree on the CPU)
    sc = o.api.scene_build(wl["scene"], **kw)
    su = o.api.scene_setup(wl["scene"])
    su.aspect_ratio = wl["width"] / wl["height"]
    cam = o.api.camera(su)
    p = o.make_params(wl["width"], wl["height"], 0, spp, list(su.background))
    n = wl["width"] * wl["height"] * spp
    # the reference's own box rule (src/aabb.rs:15-30) wherever it finishes: on the 10 M-triangle tree it lets a ray
    # through more than a million boxes (50 ms per ray), so that workload is timed with the carried-interval rule
    mode, rule = (oracle.AABB_TIGHT, "carried-interval box test") if "n_triangles" in wl else (oracle.AABB_REFERENCE, "reference AABB test")
    acc, st, secs = o.render(sc, cam, p, mode=mode, threads=threads)
    out = {"value": n / secs, "unit": "path samples/s", "cores": threads, "kind": "port",
           "sample": "%s, %d spp (%d path samples, %.1f s), %s, %d threads over image columns"
                     % (wl["name"], spp, n, secs, rule, threads),
           "rays_per_s": st["rays"] / secs, "cpu_model": cpu_model()}
    if more_threads > threads:
        _, _, secs2 = o.render(sc, cam, p, mode=mode, threads=more_threads)
        out["more_threads"] = {"value": n / secs2, "cores": more_threads, "seconds": round(secs2, 1)}
    w4, h4 = max(8, wl["width"] // 4), max(8, wl["height"] // 4)
    pa = o.make_params(w4, h4, 0, spp, list(su.background))
    pb = o.make_params(w4, h4, 0, spp, list(su.background), seed=0x5EED + 1)
    t = max(threads, more_threads)
    a = np.sqrt(o.render(sc, cam, pa, mode=mode, threads=t)[0] / spp)
    b = np.sqrt(o.render(sc, cam, pb, mode=mode, threads=t)[0] / spp)
    noise = float(np.sqrt(np.mean((a - b) ** 2)))
    o.api.scene_free(sc)
    return out, acc, noise, rule


def rmse_vs_cpu(acc_gpu, acc_cpu, spp, noise, rule):
    """BASELINE.json's "image RMSE vs CPU": the GPU's render of the very samples the CPU leg rendered (same seed, same
    sample range) against the oracle's picture, on resolved un-quantised display values sqrt(sum / spp)
    (src/main.rs:216-219)."""
    g = np.sqrt(np.asarray(acc_gpu, dtype=np.float64) / spp)
    c = np.sqrt(acc_cpu / spp)
    off = np.abs(acc_gpu - acc_cpu) > 1e-6 * np.abs(acc_cpu) + 1e-9
    return {"value": float(np.sqrt(np.mean((g - c) ** 2))), "spp": spp, "noise_floor": noise,
            "noise_floor_is": "RMSE between two CPU renders that differ only in the seed, same spp, a quarter of the image size",
            "pixels_off": float(off.any(axis=2).mean()), "pixels_off_is": "share of pixels whose accumulated value differs by more than 1e-6 relative",
            "max_abs_diff": float(np.abs(g - c).max()), "cpu_box_rule": rule, "on": "resolved [0,1] display values before quantisation"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="final", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples of every pixel per step and GPU (default: 1024, fewer on images above 0.69 Mpixel)")
    ap.add_argument("--bvh", default="reference", choices=["reference", "widest", "lbvh"],
                    help="BVHNode::new as in the reference (random split axis, src/bvh.rs:24), with the widest-spread axis, "
                         "or replaced by the linear BVH built on the GPU (rt_bvh_build)")
    ap.add_argument("--boxes-per-side", type=int, default=0,
                    help="final scene only: side of the ground grid of boxes (src/main.rs:591 has 20); larger grids no longer fit LDS")
    ap.add_argument("--shutter-bvh", action="store_true",
                    help="weekend only: build the tree over the shutter interval (0, 1) instead of the reference's (0, 0.1) (src/main.rs:315); boxes then "
                         "bound the moving spheres and the kernels run the carried-interval box rule instead of the reference's per-axis one")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every GPU renders --spp-per-step samples per step; strong: a step is a fixed job of --spp samples of every pixel split over the GPUs")
    ap.add_argument("--spp", type=int, default=10000, help="--scaling strong: samples of every pixel in the whole job (BASELINE configs[3]: 10000)")
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the CPU baseline run (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=8, help="the reference's own thread count (src/main.rs:189)")
    ap.add_argument("--cpu-more-threads", type=int, default=16,
                    help="a second CPU run with this many threads (the CPU share of a one-GPU job on the box; 0 = skip)")
    args = ap.parse_args()

    under_launcher = "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: they must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no fallback)")
    # RT_BENCH_REHEARSAL=1: several ranks share the GPUs that exist and talk over gloo, to exercise the N > 1
    # code path on a box with fewer GPUs than ranks; never a measurement
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # under a launcher the process group is made even for one rank, so that the all-reduce below goes through RCCL
    # on a one-GPU box as well
    use_dist = world > 1 or under_launcher
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = dict(WORKLOADS[args.workload])
    if args.boxes_per_side and args.workload == "final":
        wl["boxes_per_side"] = args.boxes_per_side
        wl["name"] += " with %d x %d ground boxes" % (args.boxes_per_side, args.boxes_per_side)
    if args.shutter_bvh and args.workload == "weekend":
        wl["shutter_bvh"] = True
        wl["name"] += ", tree built over the shutter interval"
    L = lib.get()
    t_build = time.time()
    hs = lib.HostScene.build(wl["scene"], **scene_kwargs(wl, args.bvh))
    t_build = time.time() - t_build
    ds = lib.DeviceScene(hs)
    su = L.host.scene_setup(wl["scene"])
    su.aspect_ratio = wl["width"] / wl["height"]  # the camera follows the image (C5: 16:9 at 4K; the others: their nominal ratio)
    cam = L.host.camera(su)
    W, H = wl["width"], wl["height"]
    S = args.spp_per_step
    # 232 bytes of path state per slot; never more than 60 % of what is free on this GPU
    free_bytes, _ = torch.cuda.mem_get_info()
    s_max = max(1, min(1024, (672 << 20) // (W * H), int(0.6 * free_bytes) // (232 * W * H)))
    if not S:
        S = s_max
    bg = list(su.background)
    accum = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    strong = args.scaling == "strong"
    if strong:
        # the job: args.spp samples of every pixel; this rank's share, cut into equal batches of at most s_max samples
        lo, hi = shard_samples(0, args.spp, rank, world)
        batches = equal_batches(lo, hi, args.spp_per_step or s_max)
        S = max((b - a for a, b in batches), default=0)

    launches = [0]

    def step(i, stats):
        if strong:
            # step i renders the whole job again, with other samples: [i * spp + lo, i * spp + hi)
            tot = abi.RtRenderStats()
            for a, b in batches:
                ds.render_device(cam, lib.make_params(W, H, i * args.spp + a, i * args.spp + b, bg, samples_per_batch=b - a), accum.data_ptr(), stream, stats=stats)
                tot.ms_extend += stats.ms_extend; tot.ms_shade += stats.ms_shade; tot.rays += stats.rays
                launches[0] += 50
            stats.ms_extend, stats.ms_shade, stats.rays = tot.ms_extend, tot.ms_shade, tot.rays
            return
        # weak: global step i: rank r renders samples [ (i*world + r)*S, +S ) of every pixel
        s0 = (i * world + rank) * S
        ds.render_device(cam, lib.make_params(W, H, s0, s0 + S, bg, samples_per_batch=S), accum.data_ptr(), stream, stats=stats)
        launches[0] += 50

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    st = abi.RtRenderStats()
    for i in range(args.warmup):
        step(i, st)
    barrier()
    ms_extend = ms_shade = 0.0
    rays = 0
    launches[0] = 0
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, st)
        ms_extend += st.ms_extend
        ms_shade += st.ms_shade
        rays += st.rays
    torch.cuda.synchronize()
    own_render = time.perf_counter() - t0  # this rank's renders alone, before it meets the others in the all-reduce
    if rehearsal and world > 1:
        host = accum.cpu()
        allreduce_accum(host)
        accum.copy_(host)
    else:
        allreduce_accum(accum)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = {"min": own_render * 1e3 / args.steps, "max": own_render * 1e3 / args.steps}
    if use_dist:
        t = torch.tensor([elapsed, own_render, -own_render], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        rank_ms = {"min": -float(t[2].item()) * 1e3 / args.steps, "max": float(t[1].item()) * 1e3 / args.steps}

    samples_total = (W * H * args.spp if strong else world * W * H * S) * args.steps
    own_samples = (W * H * (hi - lo) if strong else W * H * S) * args.steps  # of this rank, what `rays` counts
    if rank == 0:
        # algorithmic bytes of the extend launches: instrumented pass over one step's samples, outside the timed region
        scratch = torch.zeros_like(accum)
        s0 = args.warmup * world * S
        wst, work = ds.render_work(cam, lib.make_params(W, H, s0, s0 + S, bg, samples_per_batch=S), scratch.data_ptr(), stream)
        # Media that the library asks before the walk (world-space media bounded by one sphere: the kernel that makes a ray
        # puts them to it, k_generate or the previous bounce's k_shade) are counted by the instrumented pass but are not
        # work of the closest-hit launches: they come off k_extend's account and are reported on their own.
        first = {"medium": work.pop("first_media"), "sphere": work.pop("first_spheres")}
        for k in first:
            work[k] -= first[k]
            work["entries"] -= first[k]
        bytes_per_ray = algorithmic_bytes(wst.rays, work) / max(1, wst.rays)
        flops_per_ray = algorithmic_flops(work) / max(1, wst.rays)
        launches = max(1, launches[0])
        d = hs.desc
        table_bytes = int(32 * d.n_nodes + 40 * d.n_spheres + 80 * d.n_msphere + 56 * d.n_rects + 56 * d.n_boxes + 80 * d.n_triangles +
                          72 * d.n_xforms + 24 * d.n_media)
        in_lds = bool(table_bytes <= 144 * 1024)
        secs_extend = ms_extend * 1e-3
        gbs = rays * bytes_per_ray / secs_extend / 1e9 if ms_extend > 0 else 0.0
        tflops = rays * flops_per_ray / secs_extend / 1e12 if ms_extend > 0 else 0.0
        traffic = measured_traffic(profile_key(args.workload, args.bvh, wl), S)
        byte_side = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": rays * bytes_per_ray / launches, "bytes_per_ray": bytes_per_ray,
                     "measured_hbm_bytes_per_launch": traffic,
                     "measured_hbm_frac": (traffic / (secs_extend / launches) / 1e9 / HBM_PEAK_GBS) if traffic and ms_extend > 0 else None}
        flop_side = {"achieved": tflops, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_PEAK_TFLOPS,
                     "algorithmic_flops_per_ray": flops_per_ray, "measured": measured_valu(profile_key(args.workload, args.bvh, wl))}
        # What binds: with the scene tables staged in LDS the entry and primitive bytes never leave the CU (HBM carries
        # only ray records and hit slots), and the kernel is bound by f64 vector issue at low lane utilisation; the byte
        # formula is then reported beside it, not as the bound. A scene whose tables stay in global memory is HBM /
        # L2-latency bound and the byte formula is the roofline.
        head = flop_side if in_lds else byte_side
        roofline = {"bound": "valu_f64" if in_lds else "hbm", "kernel": "k_extend", "served_from": "lds" if in_lds else "hbm",
                    "achieved": head["achieved"], "peak": head["peak"], "unit": head["unit"], "frac": head["frac"],
                    "traffic": traffic, "bytes": byte_side, "flops": flop_side,
                    "entries_per_ray": work["entries"] / max(1, wst.rays),
                    "visits_per_ray": {k: round(work[k] / max(1, wst.rays), 4) for k in ROW_BYTES},
                    "ms_extend_per_step": ms_extend / args.steps, "ms_shade_per_step": ms_shade / args.steps,
                    "avg_launch_ms": ms_extend / launches, "launches": launches,
                    "tables_in_lds": in_lds, "table_bytes": table_bytes,
                    "asked_before_the_walk": {"per_ray": {k: round(first[k] / max(1, wst.rays), 4) for k in first},
                                              "flops_per_ray": sum(FLOPS[k] * first[k] for k in first) / max(1, wst.rays),
                                              "done_by": "k_generate (camera rays) and k_shade (the rays it makes); not in k_extend's figures"}}
        # The second kernel of a bounce. Algorithmic bytes of the shade launches of one step: every ray's 96-byte record
        # and 16-byte hit slot in, a 96-byte record out per ray that goes on (every ray after the camera rays is such a
        # survivor), and the three 8-byte radiance columns read and written back for a path that gathers light.
        own_steps = max(1, args.steps)
        rays_step = rays / own_steps
        survivors_step = max(0.0, rays_step - own_samples / own_steps)
        lit_step = work["lit_slots"] * (own_samples / own_steps) / max(1, wst.paths)
        shade_bytes = 112.0 * rays_step + 96.0 * survivors_step + 48.0 * lit_step
        secs_shade = ms_shade / own_steps * 1e-3
        shade_key = profile_key(args.workload, args.bvh, wl) + "/shade"
        shade_traffic = measured_traffic(shade_key, S, "bytes_per_step")
        roofline["shade"] = {"kernel": "k_shade", "bound": "hbm", "algorithmic_bytes_per_step": shade_bytes,
                             "bytes_per_ray": shade_bytes / max(1.0, rays_step),
                             "counts_per_step": {"rays": rays_step, "survivors": survivors_step, "paths_that_gathered_light": lit_step},
                             "achieved": shade_bytes / secs_shade / 1e9 if secs_shade > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": shade_bytes / secs_shade / 1e9 / HBM_PEAK_GBS if secs_shade > 0 else 0.0,
                             "traffic": shade_traffic,
                             "measured_hbm_frac": (shade_traffic / secs_shade / 1e9 / HBM_PEAK_GBS) if shade_traffic and secs_shade > 0 else None,
                             "measured": measured_valu(shade_key)}
        out = {
            "metric": "path samples/s", "value": samples_total / elapsed, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: ranks share GPUs, not a measurement)",
            "config": {"workload": wl["name"], "width": W, "height": H, "max_depth": 50, "bvh": args.bvh, "scene_build_s": round(t_build, 3),
                       "sharding": "sample ranges per rank, one RCCL all-reduce of the f32 accumulation buffer",
                       **({"spp_total": args.spp, "spp_this_rank": hi - lo, "batches_per_step": len(batches), "spp_per_batch": S} if strong
                          else {"spp_per_step_per_gpu": S})},
            # per step: the fastest and the slowest rank's own render time (imbalance between the ranks), and what the
            # all-reduce and the waiting for the slowest rank add on top
            "rank_render_ms_per_step": rank_ms, "allreduce_and_wait_ms_per_step": elapsed * 1e3 / args.steps - rank_ms["max"],
            "mrays_per_s": rays / max(1, own_samples) * samples_total / elapsed / 1e6,
            "rays_per_sample": rays / max(1, own_samples),
            "roofline": roofline,
        }
        if args.cpu_spp > 0 and world == 1:
            out["cpu_baseline"], acc_cpu, noise, rule = cpu_baseline(wl, args.cpu_spp, args.cpu_threads, args.cpu_more_threads, args.bvh)
            # the same samples on the GPU, outside the timed region
            scratch.zero_()
            ds.render_device(cam, lib.make_params(W, H, 0, args.cpu_spp, bg), scratch.data_ptr(), stream)
            torch.cuda.synchronize()
            out["rmse_vs_cpu"] = rmse_vs_cpu(scratch.cpu().numpy(), acc_cpu, args.cpu_spp, noise, rule)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
