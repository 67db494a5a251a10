#!/usr/bin/env python3
"""Debug helper (GPU box, library built with -DMZD_XC_STATS, selected with MZD_LIB): what k_exec_c's stretches and passes do."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
if len(sys.argv) > 1 and sys.argv[1] == "corpus":  # the reference's 100 frames, replicated
    import json
    golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
    names = sorted(json.load(open(os.path.join(golden, "manifest.json"))))
    frames = [open(os.path.join(golden, nm + ".zst"), "rb").read() for nm in names] * (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
elif len(sys.argv) > 1 and sys.argv[1] == "largest":  # the corpus's k-th largest frame alone, n copies
    import json
    golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
    man = json.load(open(os.path.join(golden, "manifest.json")))
    names = sorted(man, key=lambda nm: -man[nm]["length"])
    nm = names[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
    print("frame", nm, man[nm]["length"], "bytes")
    frames = [open(os.path.join(golden, nm + ".zst"), "rb").read()] * (int(sys.argv[3]) if len(sys.argv) > 3 else 256)
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=8)
    frames = [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
ctx = z.Context(0, exec_variant=5)
L = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
outs, sts = z.decode_frames(frames, ctx)  # warm
L.mzd_debug_xc_stats(buf, 1)
outs, sts = z.decode_frames(frames, ctx)
assert all(s == 0 for s in sts)
L.mzd_debug_xc_stats(buf, 0)
names = ["tiles", "stretches", "passes (128 B)", "extra fixed-point rounds", "passes resolved by pointer jumping", "passes that went to memory",
         "staged matches", "matches", "cycles setup", "cycles plan + flush", "cycles passes", "cycles total", "frames", "stretches on the general path",
         "cycles in blocks without sequences (Raw, RLE, literals only)", "blocks without sequences"]
t = max(buf[0], 1)
for i, nm in enumerate(names):
    print(f"{nm:36s} {buf[i]:14d}  per tile {buf[i] / t:10.3f}")
print("cycles per pass", buf[10] / max(buf[2], 1), " other per tile", (buf[11] - buf[8] - buf[9] - buf[10]) / t)
