"""The N > 1 path on CPU: two processes (gloo) shard the sample range, each renders its share, one
all-reduce sums the accumulation buffers. The renderer is injected; here it is the CPU oracle, so the
test checks the sharding and reduction logic that bench.py runs with the HIP renderer over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from raytracer_amd.distributed import equal_batches, render_sharded, shard_samples


def test_shard_samples_partition():
    for total in (0, 1, 7, 8, 1000, 10000):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_samples(5, 5 + total, r, world) for r in range(world)]
            assert parts[0][0] == 5 and parts[-1][1] == 5 + total
            for a, b in zip(parts, parts[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_samples(0, 8, 2, 2)


def test_library_shards_like_the_python_side(rtlib):
    # rt_shard_samples (what rt_sharded_render uses inside the library) and distributed.shard_samples (what bench.py and
    # the command-line renderer use across processes) are the same partition
    from raytracer_amd import lib
    for begin, total in ((0, 0), (5, 1), (0, 7), (3, 8), (0, 1000), (100, 10000)):
        for world in (1, 2, 3, 4, 8):
            for r in range(world):
                assert lib.shard_samples(rtlib, begin, begin + total, r, world) == shard_samples(begin, begin + total, r, world)
    assert lib.shard_samples(rtlib, 4, 9, 3, 2) == (4, 4)  # a rank outside the world owns nothing
    assert rtlib.lib.rt_sharded_create(None, None, 0, None) == -1
    assert rtlib.lib.rt_sharded_render(None, None, None, None, None) == -1
    assert rtlib.lib.rt_sharded_device_count(None) == 0
    rtlib.lib.rt_sharded_destroy(None)


def test_equal_batches_leave_no_runt():
    # BASELINE configs[3] on 8 GPUs: 10 000 spp / 8 = 1250 per GPU; with at most 1024 per batch that is 2 x 625, not 1024 + 226
    assert equal_batches(0, 1250, 1024) == [(0, 625), (625, 1250)]
    assert equal_batches(10, 10, 5) == []
    for total in (1, 7, 1024, 1025, 10000):
        for mx in (1, 3, 1024):
            parts = equal_batches(3, 3 + total, mx)
            sizes = [b - a for a, b in parts]
            assert parts[0][0] == 3 and parts[-1][1] == 3 + total and max(sizes) <= mx and max(sizes) - min(sizes) <= 1
            assert len(parts) == -(-total // mx)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    o = oracle.get()
    sid, W, H, S = 7, 24, 24, 6
    sc = o.api.scene_build(sid)
    su = o.api.scene_setup(sid)
    cam = o.api.camera(su)

    def render_range(lo, hi, accum):
        acc, _, _ = o.render(sc, cam, o.make_params(W, H, lo, hi, list(su.background)), threads=1)
        accum += torch.from_numpy(acc.astype(np.float32))

    accum = torch.zeros(H, W, 3, dtype=torch.float32)
    accum, (lo, hi) = render_sharded(render_range, 0, S, accum)
    assert (lo, hi) == shard_samples(0, S, rank, world)
    if rank == 0:
        full, _, _ = o.render(sc, cam, o.make_params(W, H, 0, S, list(su.background)), threads=1)
        np.save(out_path, np.stack([accum.numpy().astype(np.float64), full]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_sharding_matches_single_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    sharded, full = np.load(out)
    # identical samples, summed in a different order and through float32: equal to f32 rounding
    assert np.allclose(sharded, full, rtol=1e-6, atol=1e-6)
    assert full.sum() > 0
