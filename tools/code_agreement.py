#!/usr/bin/env python3
"""How often does the device report the SAME error code as the oracle (= the reference's sentinel) on
mutated frames?  Report only: the device plans a whole frame before decoding any of it, so a frame with
two independent defects may name the other one.  usage: python tools/code_agreement.py [n] [seed]"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tests.oracle_binding import load_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = load_oracle()
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
names = sorted(json.load(open(os.path.join(golden, "manifest.json"))))
base = [open(os.path.join(golden, x + ".zst"), "rb").read() for x in names]
base = [b for b in base if 24 <= len(b) <= 100000]
frames = []
for _ in range(n):
    b = bytearray(base[int(rng.integers(len(base)))])
    if rng.random() < 0.15:
        b = b[:int(rng.integers(1, len(b)))]
    else:
        b[int(rng.integers(4, len(b)))] ^= int(rng.integers(1, 256))  # ONE defect
    frames.append(bytes(b))
c = z.Context(0)
outs, sts = z.decode_frames(frames, c, device_plan=True)
pairs = collections.Counter()
for f, s in zip(frames, sts):
    rc = orc.decode_frame(f, cap=4 << 20)[0]
    pairs[(rc, s)] += 1
same = sum(v for (a, b), v in pairs.items() if a == b)
print(f"{n} frames: same code {same} ({100.0 * same / n:.2f} %)")
for (a, b), v in sorted(pairs.items(), key=lambda kv: -kv[1]):
    if a != b:
        print(f"  oracle {a:3d} ({orc.strerror(a) if a else 'ok'})  device {b:3d} ({z.strerror(b) if b else 'ok'})  x{v}")
