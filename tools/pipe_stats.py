#!/usr/bin/env python3
"""Debug helper (GPU box, library built with -DMZD_PIPE_STATS, MZD_LIB pointing at it): statistics of stage A of
k_seq_pipe summed over ALL workgroups of whole passes of the bench batch (tools/pipe_prof.py sees workgroup 0 only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=16)
ctx = z.Context(0)
rb = ctx.upload_frames(blob[:int(off[-1] + ln[-1])], off, ln)
L = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
rb.run(); ctx.sync()
L.mzd_debug_pipe_stats(buf, 1)
passes = 3
for _ in range(passes):
    rb.run()
ctx.sync()
L.mzd_debug_pipe_stats(buf, 0)
wg, steps, cyc, qp, rp = (buf[i] / passes for i in range(5))
print(f"per pass: workgroups {wg:.0f}, stage-A steps {steps:.0f}, cycles per step {cyc / steps:.1f}, queue-full polls per batch {qp / (steps / 4):.3f}, "
      f"ring polls per batch {rp / (steps / 4):.3f}")
print("kernel ms:", ctx.kernel_ms())
