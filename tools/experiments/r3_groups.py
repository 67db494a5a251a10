#!/usr/bin/env python3
"""Experiment: the corpus batch as G sub-batches (replicas dealt round robin) resident side by side, their passes submitted to G
contexts (own streams) back to back -- what a software pipeline over frame groups could give a heterogeneous batch.
usage: python tools/experiments/r3_groups.py [groups ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z

gdir = os.path.join(ROOT, "tests", "golden", "decodecorpus")
manifest = json.load(open(os.path.join(gdir, "manifest.json")))
names = sorted(manifest)
frames = [np.fromfile(os.path.join(gdir, n + ".zst"), dtype=np.uint8) for n in names]
lens = np.array([manifest[n]["length"] for n in names], dtype=np.uint64)
reps = int(4.0 * 2**30 / float(lens.sum()) + 0.5)
one = np.concatenate(frames)
o1 = np.concatenate([[0], np.cumsum([f.size for f in frames])[:-1]]).astype(np.uint64)
l1 = np.array([f.size for f in frames], dtype=np.uint64)
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    ctxs, rbs, plans = [], [], []
    for g in range(G):
        r = len(range(g, reps, G))
        blob = np.tile(one, r)
        off = (np.tile(o1, r) + np.repeat(np.arange(r, dtype=np.uint64) * np.uint64(one.size), len(names))).astype(np.uint64)
        ln = np.tile(l1, r)
        plan = z.Plan(device_tables=True)
        assert plan.add_frames(blob, off, ln, threads=8) == 0
        c = z.Context(0)
        rb = c.upload(plan.finalize())
        ctxs.append(c); rbs.append(rb); plans.append(plan)
    def one_pass():
        for rb in rbs:
            rb.run()
        for c in ctxs:
            c.sync()
    one_pass(); one_pass()
    t0 = time.time()
    K = 5
    for _ in range(K):
        one_pass()
    ms = (time.time() - t0) / K * 1e3
    ok = all(int(s) == 0 for rb in rbs for s in rb.download()[1][:200])
    print(f"{G} group(s): {ms:.2f} ms per pass of {reps} replicas ({float(lens.sum()) * reps / ms / 1e6:.0f} GB/s), statuses ok: {ok}", flush=True)
    for rb in rbs: rb.free()
    for p in plans: p.close()
    for c in ctxs: c.close()
