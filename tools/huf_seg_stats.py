#!/usr/bin/env python3
"""Validation statistics of k_huf_seg on BASELINE config 3 (build with -DMZD_HUF_SEG_STATS, MZD_LIB=that library)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
blob, off, ln, ck, ns = sb.make_batch(cfg, 0, n, 131072, threads=0)
plan = z.Plan(device_tables=True)
assert plan.add_frames(blob, off, ln, threads=0) == 0
ctx = z.Context(0, huf_variant=2)
rb = ctx.upload(plan.finalize())
L = _lib.load()
st = (ctypes.c_ulonglong * 16)()
L.mzd_debug_huf_seg_stats(st, 1)
rb.run(); ctx.sync()
L.mzd_debug_huf_seg_stats(st, 0)
print("rounds", st[0], "validation rounds", st[1], "lanes recounted", st[2], "active lanes", st[3], "lanes whose symbols did not fit", st[5])
tot = max(st[13], 1)
print("wavefront cycles per round: fill %.0f, approach + count %.0f, validation %.0f, scan %.0f, write %.0f; whole stream %.0f per round" %
      tuple(x / max(st[0], 1) for x in (st[8], st[9], st[10], st[11], st[12], st[13])))
