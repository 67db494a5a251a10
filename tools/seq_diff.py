#!/usr/bin/env python3
"""Records / tile bases / block sums of two sequence-decode variants on corpus frames, entropy stages only
(MZD_DEBUG_SEQ_ONLY=1, a library built with -DMZD_EXPERIMENTS selected through MZD_LIB): first difference.
usage: MZD_LIB=tmp_ab/libmzd_exp.so MZD_DEBUG_SEQ_ONLY=1 seq_diff.py name [name...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
names = sys.argv[1:]
frames = [open(os.path.join(golden, n + ".zst"), "rb").read() for n in names]
res = {}
for sv in (3, 2):
    ctx = z.Context(0, seq_variant=sv, huf_variant=1)
    plan = z.Plan(device_tables=True)
    for f in frames:
        assert plan.add_frame(f)[0] == 0
    b = plan.finalize()
    rb = ctx.upload(b)
    rb.run(); ctx.sync()
    st = rb.stats()
    n = int(st.n_sequences)
    recs = rb.debug_read(_lib.MZD_DEBUG_RECORDS, np.uint64, 0, n)
    blocks = rb.debug_blocks(b.n_blocks)
    nt = sum((int(d.n_seq) + 63) // 64 for d in blocks)
    tiles = rb.debug_read(_lib.MZD_DEBUG_TILES, np.uint32, 0, 2 * nt)
    res[sv] = (recs, tiles, [(int(d.rec_off), int(d.n_seq), int(d.tile_off)) for d in blocks])
    rb.free()
r0, t0, b0 = res[3]
r2, t2, b2 = res[2]
print("sequences", len(r0), "equal records", bool((r0 == r2).all()), "equal tiles", bool((t0 == t2).all()))
if not (r0 == r2).all():
    i = int(np.nonzero(r0 != r2)[0][0])
    blk = max(k for k, (ro, ns, to) in enumerate(b0) if ns and ro <= i)
    print("first difference at record", i, "block", blk, "seq in block", i - b0[blk][0], "of", b0[blk][1])
    for j in range(max(i - 2, 0), min(i + 3, len(r0))):
        def f(r): r = int(r); return (r & 0x1FFFF, (r >> 17) & 0x3FFFF, (r >> 35) & 0x1FFFFFFF)
        print(j, f(r0[j]), f(r2[j]))
    nd = int((r0 != r2).sum())
    print("differing records", nd)
