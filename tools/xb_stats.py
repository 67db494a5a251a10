#!/usr/bin/env python3
"""Debug helper (GPU box, library built with -DMZD_XB_STATS, selected with MZD_LIB): what k_exec_b's tiles and passes do."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=8)
frames = [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
ctx = z.Context(0, exec_variant=2)
L = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
outs, sts = z.decode_frames(frames, ctx)  # warm
L.mzd_debug_xb_stats(buf, 1)
outs, sts = z.decode_frames(frames, ctx)
assert all(s == 0 for s in sts)
L.mzd_debug_xb_stats(buf, 0)
names = ["tiles", "stretches", "plain passes", "generic passes", "passes with in-pass bytes", "passes that went to memory",
         "staged matches", "matches", "cycles tile setup", "cycles stretch setup", "cycles passes", "cycles total", "frames"]
t = max(buf[0], 1)
for i, nm in enumerate(names):
    print(f"{nm:34s} {buf[i]:14d}  per tile {buf[i] / t:10.3f}")
np_ = max(buf[2] + buf[3], 1)
print("cycles per pass", buf[10] / np_, " per stretch setup", buf[9] / max(buf[1], 1), " per tile setup", buf[8] / t,
      " other per tile", (buf[11] - buf[8] - buf[9] - buf[10]) / t)
