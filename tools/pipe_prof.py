#!/usr/bin/env python3
"""Debug helper (GPU box, library built with -DMZD_PIPE_PROF, MZD_LIB pointing at it): per-stage cycle
counts of k_seq_pipe's workgroup 0 for a batch of `n` synthetic config-4 frames (n = 256 * 57 fills every
CU with 57 chains; n = 256 gives one chain per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 57
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=16)
ctx = z.Context(0, no_split=True)
rb = ctx.upload_frames(blob[:int(off[-1] + ln[-1])], off, ln)
rb.run()
ctx.sync()
_, st, _ = rb.download(want_out=False)
assert (st == 0).all() or os.environ.get('MZD_PROF_IGNORE_STATUS')
