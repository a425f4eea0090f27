"""GPU box: seeded campaign of rt_bvh_build, rt_bvh_build_device and rt_bvh_build_device_noalloc against the oracle's CPU
statement of the same tree (tokens bit for bit) over box sets of random size and shape: clustered, duplicated, flat, identical, tiny and huge coordinates.
    python tools/fuzz_bvh_build.py [N_CASES]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from raytracer_amd import abi, lib

NOALLOC_MAX = 65536  # RT_BVH_NOALLOC_MAX_OBJECTS


def device_build(gpu, entry, boxes):
    """rt_bvh_build_device / rt_bvh_build_device_noalloc on device-resident boxes (torch holds the memory)."""
    import torch
    n = boxes.shape[0]
    d_boxes = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float64)).cuda()
    d_tokens = torch.zeros((2 * n - 1) * abi.NODE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    args = [d_boxes.data_ptr(), n, d_tokens.data_ptr(), None, None]
    if entry == "rt_bvh_build_device":
        args.append(None)
    gpu.check(getattr(gpu.lib, entry)(*args))
    torch.cuda.synchronize()
    return d_tokens.cpu().numpy().tobytes()


def box_set(rng):
    n = int(rng.choice([rng.integers(2, 70), rng.integers(60, 200), rng.integers(200, 5000), rng.integers(5000, 60000)]))
    scale = float(10.0 ** rng.uniform(-6, 6))
    c = rng.normal(0.0, scale, (n, 3))
    style = int(rng.integers(0, 6))
    if style == 1:   # clusters far apart
        c += rng.integers(0, 4, (n, 1)) * scale * 100.0
    elif style == 2:  # many centres several times
        c[: n // 2] = c[rng.integers(0, max(1, n // 8), n // 2)]
    elif style == 3:  # on a plane / on a line
        c[:, int(rng.integers(0, 3))] = float(rng.normal())
        if rng.uniform() < 0.5:
            c[:, int(rng.integers(0, 3))] = float(rng.normal())
    elif style == 4:  # all the same
        c[:] = c[0]
    elif style == 5:  # a regular grid (equal Morton codes along rows)
        g = int(np.ceil(n ** (1.0 / 3.0)))
        idx = np.arange(n)
        c = np.stack([idx % g, (idx // g) % g, idx // (g * g)], axis=1).astype(np.float64) * scale
    h = rng.uniform(0.0, 0.3 * scale, (n, 3))
    if rng.uniform() < 0.3:
        h[::3, int(rng.integers(0, 3))] = 0.0
    if rng.uniform() < 0.1:
        h[:] = 0.0
    return np.concatenate([c - h, c + h], axis=1), style


def main(n_cases):
    gpu, orc = lib.get(), oracle.get()
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(31000 + case)
        boxes, style = box_set(rng)
        got, _ = gpu.bvh_build(boxes)
        want = orc.lbvh_tokens(boxes)
        same = got.tobytes() == want.tobytes()
        # the device-resident entry points, the second of them in the shared static scratch array (calls of every size in a row)
        same_dev = device_build(gpu, "rt_bvh_build_device", boxes) == want.tobytes()
        same_noalloc = len(boxes) > NOALLOC_MAX or device_build(gpu, "rt_bvh_build_device_noalloc", boxes) == want.tobytes()
        if not (same and same_dev and same_noalloc):
            bad += 1
            print("case %d: %d boxes, style %d: host-buffer build %s, device build %s, no-alloc build %s" % (case, len(boxes), style, same, same_dev, same_noalloc))
    print("%d box sets: %d mismatches" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 100))
