"""Multi-GPU sharding of the sample loop: one process per GPU, every rank renders a
contiguous range of the samples of EVERY pixel into its own accumulation buffer, and one
all-reduce (RCCL over xGMI when the tensors live on GPUs; gloo on CPU in the tests) sums
the buffers. The reference has nothing comparable (its 8 threads split image columns,
src/main.rs:193-195); sample ranges balance better than columns or tiles, and because the
RNG is keyed by (pixel, sample, bounce) the result does not depend on the partition except
for the order of the f32 additions."""
import torch
import torch.distributed as dist


def shard_samples(sample_begin, sample_end, rank, world_size):
    """Samples [begin, end) owned by `rank`: contiguous, disjoint, covering, sizes differing by at most 1."""
    total = sample_end - sample_begin
    if total < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad shard request")
    base, rem = divmod(total, world_size)
    lo = sample_begin + rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def equal_batches(lo, hi, max_batch):
    """Cuts samples [lo, hi) into the fewest batches of at most max_batch samples, sizes differing by at most 1: a
    wavefront batch runs well below peak when it is small, so no short batch is left over at the end."""
    total = hi - lo
    if total <= 0:
        return []
    if max_batch < 1:
        raise ValueError("bad batch size")
    n = -(-total // max_batch)
    return [shard_samples(lo, hi, k, n) for k in range(n)]


def allreduce_accum(accum, group=None):
    """In-place sum of the per-rank accumulation buffers (float32 [H, W, 3]) over all ranks."""
    if dist.is_available() and dist.is_initialized():  # also with one rank: the collective then runs through RCCL all the same
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_sharded(render_range, sample_begin, sample_end, accum, group=None):
    """render_range(lo, hi, accum) adds samples [lo, hi) of every pixel into `accum` (a torch tensor on the
    rank's device). Returns accum holding the sum over all ranks and the (lo, hi) this rank rendered."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_samples(sample_begin, sample_end, rank, world)
    if hi > lo:
        render_range(lo, hi, accum)
    allreduce_accum(accum, group)
    return accum, (lo, hi)
