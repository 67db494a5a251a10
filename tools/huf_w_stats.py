#!/usr/bin/env python3
"""Phase cycles of k_huf_w (build with -DMZD_HUF_W_STATS, MZD_LIB=that library): usage huf_w_stats.py [config] [frames] [huf_variant]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
hv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
blob, off, ln, ck, ns = sb.make_batch(cfg, 0, n, 131072, threads=0)
plan = z.Plan(device_tables=True)
assert plan.add_frames(blob, off, ln, threads=0) == 0
ctx = z.Context(0, huf_variant=hv)
rb = ctx.upload(plan.finalize())
L = _lib.load()
st = (ctypes.c_ulonglong * 16)()
rb.run(); ctx.sync()
L.mzd_debug_huf_w_stats(st, 1)
rb.run(); ctx.sync()
L.mzd_debug_huf_w_stats(st, 0)
r = max(st[0], 1)
print(f"config {cfg}, {n} frames: rounds {st[0]}, validation rounds {st[1]}, lanes decoded again {st[2]}, rounds run again with half the segment {st[3]}")
print("wavefront cycles per round: load %.0f, decode %.0f, validation %.0f, compaction %.0f, store %.0f; whole stream %.0f per round" %
      tuple(x / r for x in (st[8], st[9], st[10], st[11], st[12], st[13])))
