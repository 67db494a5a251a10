"""Static sharding of a batch of independent frames over the GPUs of one node.

Frames share nothing (tables, offset history and window are per frame: framedecompressor.go:42-52),
so rank r of W simply takes a contiguous range of frames; there is NO collective on the data path.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only by callers that want a barrier
or to gather the per-rank status words."""


def frame_range(n_frames: int, rank: int, world: int):
    """Contiguous, balanced split: the first (n % world) ranks get one extra frame."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def balanced_ranges(costs, world: int):
    """Contiguous ranges with roughly equal total cost (e.g. compressed + decompressed bytes per
    frame) for heterogeneous batches.  -> list of (lo, hi)."""
    total = float(sum(costs))
    out, lo, acc, target = [], 0, 0.0, 0.0
    for r in range(world):
        target += total / world
        hi = lo
        while hi < len(costs) and (acc + costs[hi] <= target or hi == lo) and (len(costs) - hi) > (world - r - 1):
            acc += costs[hi]
            hi += 1
        if r == world - 1:
            hi = len(costs)
        out.append((lo, hi))
        lo = hi
    return out
