#!/usr/bin/env python3
"""Whole-pass statistics of k_seq_q4's chain wavefronts (build with -DMZD_Q4_STATS, MZD_LIB=that library)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13824
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, 131072, threads=0)
plan = z.Plan(device_tables=True)
assert plan.add_frames(blob, off, ln, threads=0) == 0
ctx = z.Context(0, seq_variant=2, no_split=True)
rb = ctx.upload(plan.finalize())
L = _lib.load()
st = (ctypes.c_ulonglong * 8)()
rb.run(); ctx.sync()
L.mzd_debug_q4_stats(st, 1)
ctx.timing_reset(True)
rb.run(); ctx.sync()
L.mzd_debug_q4_stats(st, 0)
print("kernel ms", ctx.kernel_ms())
w, steps, cyc, qf, rp, gen = st[0], st[1], st[2], st[3], st[4], st[5]
print(f"chain wavefronts {w}, steps/wavefront {steps / w:.0f}, cycles/step {cyc / steps:.1f}, queue-full polls/batch {qf / (steps / 4):.3f}, "
      f"ring polls/batch {rp / (steps / 4):.3f}, general steps/wavefront {gen / w:.1f}, cycles per general step {st[6] / max(gen, 1):.0f} "
      f"({100.0 * st[6] / cyc:.1f} % of stage A's time)")
r = st[7]
print(f"chains in general steps, by reason: last sequence {r & 0xFFFFF}, escape cell {(r >> 20) & 0xFFFFF}, more bits than the window holds {r >> 40}")
_, status, _ = rb.download(want_out=False)
print("status ok", bool((status == 0).all()))
