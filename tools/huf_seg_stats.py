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
st = (ctypes.c_ulonglong * 8)()
L.mzd_debug_huf_seg_stats(st, 1)
rb.run(); ctx.sync()
L.mzd_debug_huf_seg_stats(st, 0)
print("streams", st[0], "validation rounds", st[1], "lanes recounted", st[2], "active lanes", st[3])
