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


def _bench_args(*argv):
    """bench.py's own argument parser on a made-up command line (no GPU is touched: parse() only reads sys.argv)"""
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py", *argv]
    try:
        return bench, bench.parse()
    finally:
        sys.argv = old


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bench_rank_arithmetic_of_a_node(world):
    """What `bench.py --gpus N` does with its ranks, the device calls left out: BASELINE configs[4] is ONE 65 536-frame batch cut into
    contiguous ranges (8 192 per GPU at N = 8), every frame in exactly one range, the host threads of the node shared between the
    ranks, and the line's per-GPU rows / whole-job figures assembled from what the ranks gather (value = the bytes of ALL ranks over
    the SLOWEST rank's clock).  The first 8-GPU lease has to produce a curve, not a traceback: this is its rank arithmetic on CPU."""
    bench, a = _bench_args("--gpus", str(world))
    shares = [bench.rank_share(a, r, world) for r in range(world)]
    assert [s[2] for s in shares] == ["strong"] * world and all(s[3] == 65536 for s in shares)
    assert shares[0][0] == 0 and sum(s[1] for s in shares) == 65536 and all(s[1] == 65536 // world for s in shares)
    assert all(shares[r][0] + shares[r][1] == shares[r + 1][0] for r in range(world - 1))
    # a batch that does not divide: the remainder goes to the first ranks, nothing is lost
    bench, a = _bench_args("--gpus", str(world), "--frames", "65539")
    shares = [bench.rank_share(a, r, world) for r in range(world)]
    assert sum(s[1] for s in shares) == 65539 and max(s[1] for s in shares) - min(s[1] for s in shares) <= 1
    assert all(shares[r][0] + shares[r][1] == shares[r + 1][0] for r in range(world - 1))
    # --weak: the whole batch on every rank, at frame indices of its own
    bench, a = _bench_args("--gpus", str(world), "--weak")
    shares = [bench.rank_share(a, r, world) for r in range(world)]
    assert [s[:3] for s in shares] == [(r * 65536, 65536, "weak") for r in range(world)]
    # the corpus is split by replicas
    bench, a = _bench_args("--gpus", str(world), "--workload", "corpus")
    assert sum(bench.rank_share(a, r, world, corpus_reps=372)[1] for r in range(world)) == 372
    # the node's host threads are shared between its ranks, never zero
    assert bench.gen_threads_for(a, world, 16) == max(1, 16 // world) and bench.gen_threads_for(a, world, 1) == 1
    # the line: the ranks' 8-vectors as all_gather hands them over (rank, device, frames, ms, path ms, GB/s, C bytes, D bytes)
    per = 65536 // world
    c1, d1 = per * 45694, per * 131072
    ms = [2.8 + 0.01 * r for r in range(world)]  # the last rank is the slowest
    rows, c_all, d_all = bench.per_gpu_rows([[r, r, per, ms[r], ms[r] - 0.05, (c1 + d1) / (ms[r] - 0.05) / 1e6, c1, d1] for r in range(world)])
    assert [p["rank"] for p in rows] == list(range(world)) and c_all == world * c1 and d_all == world * d1 == 65536 * 131072
    assert all(p["frames"] == per and 0 < p["hbm_frac"] < 1 for p in rows)
    job = bench.job_figures(rows, c_all, d_all, elapsed=ms[-1] * 1e-3 * 20, steps=20, world=world)
    assert abs(job["ms_per_step"] - ms[-1]) < 1e-9 and abs(job["value"] - d_all / (ms[-1] * 1e-3) / 1e6) < 1e-3
    agg = job["aggregate"]
    assert agg["slowest_rank"] == world - 1 and agg["devices"] == list(range(world)) and agg["devices_distinct"] is True
    assert 0 < agg["hbm_frac_of_all_gpus"] < 1
    # two ranks on ONE device are named as such (the line's assert in bench.py refuses them outside the one-GPU test hook)
    if world > 1:
        rows2, _, _ = bench.per_gpu_rows([[r, 0, per, 1.0, 1.0, 1.0, c1, d1] for r in range(world)])
        assert bench.job_figures(rows2, c_all, d_all, 1.0, 1, world)["aggregate"]["devices_distinct"] is False
