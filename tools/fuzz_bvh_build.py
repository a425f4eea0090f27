"""GPU box: seeded campaign of rt_bvh_build against the oracle's CPU statement of the same tree (tokens bit for bit) over
box sets of random size and shape: clustered, duplicated, flat, identical, tiny and huge coordinates.
    python tools/fuzz_bvh_build.py [N_CASES]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from raytracer_amd import lib


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
        if got.tobytes() != want.tobytes():
            bad += 1
            print("case %d: %d boxes, style %d: tokens differ" % (case, len(boxes), style))
    print("%d box sets: %d mismatches" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 100))
