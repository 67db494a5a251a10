#!/usr/bin/env python3
"""Debug helper (GPU box, library built with -DMZD_EXEC_STATS): per-tile dataflow statistics of k_exec."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=8)
frames = [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
ctx = z.Context(0, seq_variant=int(os.environ.get("SEQ_VARIANT", "0")), exec_threads=int(os.environ.get("EXEC_THREADS", "0")), exec_chunk=int(os.environ.get("EXEC_CHUNK", "0")))
L = _lib.load()
buf = (ctypes.c_ulonglong * 32)()
L.mzd_debug_exec_stats(buf, 1)
outs, sts = z.decode_frames(frames, ctx)
assert all(s == 0 for s in sts)
L.mzd_debug_exec_stats(buf, 0)
names = ["tiles", "pending matches", "far (a) matches", "dataflow iterations", "fast passes", "fast lanes", "slowb passes",
         "slowb lanes", "long tries", "idle spins", "long literals", "short literal lanes"]
t = buf[0]
for i, nm in enumerate(names):
    print(f"{nm:22s} {buf[i]:12d}  per tile {buf[i] / t:8.3f}")

