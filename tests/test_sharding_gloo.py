"""The N>1 path on CPU: world_size 2, gloo.  Each rank plans ITS OWN frame range of the same
deterministic batch (no data-path collective); the ranks only exchange checksums of their
descriptor-interpreted output to prove the split covers every frame exactly once."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import ctypes
    import sparkzstd_amd as z
    from sparkzstd_amd.sharding import frame_range
    from tests.desc_interp import run_batch
    from tools import synth_binding as sb
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frame_range(n_frames, rank, world)
    blob, off, ln, ck, ns = sb.make_batch(4, lo, hi - lo, frame_bytes=4096, threads=1)
    plan = z.Plan()
    assert plan.add_frames(blob, off, ln, threads=1) == 0
    b = plan.finalize()
    outs = run_batch(b, bytes((ctypes.c_uint8 * b.in_size).from_address(b.in_)))
    mine = np.array([sb.checksum64(o) for o in outs], dtype=np.uint64)
    assert (mine == ck).all()
    # gather (frame index, checksum) to rank 0 -- bookkeeping only, not part of the decode path
    t = torch.zeros(n_frames, dtype=torch.int64)
    t[lo:hi] = torch.from_numpy(mine.view(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put(t.numpy().view(np.uint64).copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_cover_the_batch_once():
    from tools import synth_binding as sb
    n_frames = 11  # ragged split: 6 + 5
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, _, _, ck, _ = sb.make_batch(4, 0, n_frames, frame_bytes=4096, threads=1)
    assert (got == ck).all()


def test_frame_range_and_balanced_ranges():
    from sparkzstd_amd.sharding import balanced_ranges, frame_range
    for n in (0, 1, 7, 8, 65536):
        for w in (1, 2, 3, 8):
            rs = [frame_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
    costs = [1] * 10 + [10] * 2
    rs = balanced_ranges(costs, 3)
    assert rs[0][0] == 0 and rs[-1][1] == len(costs) and all(rs[i][1] == rs[i + 1][0] for i in range(2))
