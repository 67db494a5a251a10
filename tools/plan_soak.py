#!/usr/bin/env python3
"""GPU soak of the device-side planner (k_parse, mzd_batch_upload_frames): corpus and synthetic
frames, intact / mutated anywhere (headers included) / truncated, decoded once with the host planner
and once with the device planner.  Status and bytes must agree frame for frame.
usage: python tools/plan_soak.py [n_frames] [seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
names = sorted(json.load(open(os.path.join(golden, "manifest.json"))))
base = [open(os.path.join(golden, n + ".zst"), "rb").read() for n in names]
base = [b for b in base if len(b) <= 200000]
for i in range(60):
    base.append(sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 1900 + i, int(rng.integers(1, 300000))))[0])
ctxs = [z.Context(0, verify_checksum=True), z.Context(0, seq_variant=1)]
bad = done = n_ok = 0
t0 = time.time()
while done < n_total:
    frames = []
    for _ in range(min(2000, n_total - done)):
        b = bytearray(base[int(rng.integers(len(base)))])
        r = rng.random()
        if r < 0.15:
            b = b[:int(rng.integers(0, len(b)))]
        elif r < 0.85:
            lo = 0 if rng.random() < 0.3 else 4
            for pos in rng.integers(lo, len(b), size=int(rng.integers(1, 4))):
                b[int(pos)] ^= int(rng.integers(1, 256))
        frames.append(bytes(b))
    c = ctxs[(done // 2000) % 2]
    oh, sh = z.decode_frames(frames, c)
    od, sd = z.decode_frames(frames, c, device_plan=True)
    for i in range(len(frames)):
        if sh[i] != sd[i] or oh[i] != od[i]:
            bad += 1
            print("DISAGREE frame", i, "host", sh[i], "device", sd[i], "len", len(frames[i]), flush=True)
        n_ok += sd[i] == 0
    done += len(frames)
    print(f"{done} frames, {n_ok} decoded, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("PLAN SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(0 if bad == 0 else 1)
