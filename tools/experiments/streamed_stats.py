#!/usr/bin/env python3
"""GPU box: who executed the frames of a STREAMED pass -- frames claimed per XCD, frames a streamed wavefront executed, frames left to the
plain launch behind (mzd_batch_debug_read MZD_DEBUG_STREAM).  usage: python tools/stream_stats.py [frames=8192]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, 131072, threads=0)
plan = z.Plan(device_tables=True)
assert plan.add_frames(blob, off, ln, threads=0) == 0
ctx = z.Context(0)
rb = ctx.upload(plan.finalize())
for _ in range(2):
    rb.run(); ctx.sync()
print("last_pass", rb.last_pass(), "streamed", bool(rb.last_pass() & _lib.MZD_PASS_STREAMED))
w = rb.debug_read(_lib.MZD_DEBUG_STREAM, np.uint32, 0, n + 8 + n)
prog, claim, done = w[:n], w[n:n + 8], w[n + 8:]
print("claims per XCD", claim.tolist(), "| frames executed by streamed wavefronts", int(done.sum()), "of", n)
print("producer XCD of the first 16 blocks", ((prog[:16] >> 24) & 15).tolist(), "| final flags", int((prog >> 31).sum()))
nd = np.nonzero(done == 0)[0]
print("first frames left to the plain launch", nd[:16].tolist())
