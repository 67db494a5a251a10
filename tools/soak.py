#!/usr/bin/env python3
"""GPU soak (not part of the test suite): thousands of frames of random kind / size / mode, decoded in
batches by both sequence kernels and compared with the content they were made from.
usage: python tools/soak.py [n_frames] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctxs = [z.Context(0, seq_variant=0), z.Context(0, seq_variant=1, exec_threads=256), z.Context(0, seq_variant=0, exec_threads=64, exec_chunk=4096, huf_variant=2),
        z.Context(0, seq_variant=3, huf_variant=1), z.Context(0, huf_variant=3)]
bad = 0
done = 0
t0 = time.time()
while done < n_frames:
    frames, want = [], []
    for i in range(min(1500, n_frames - done)):
        kind = int(rng.choice([sb.TEXT, sb.TEXT, sb.EXP, sb.RANDOM, sb.ZERO]))
        n = int(rng.integers(0, 5000)) if rng.random() < 0.4 else int(rng.integers(0, 400000))
        data = sb.generate(kind, int(rng.integers(1 << 40)), n)
        r = rng.random()
        if n > 64 and r < 0.35:      # periodic / repeated / noisy splices: overlaps, long matches, long literal runs
            k = int(rng.integers(1, 300))
            a, b = sorted(int(x) for x in rng.integers(0, n, 2))
            mid = (data[:k] * ((b - a) // k + 1))[:b - a] if rng.random() < 0.5 else sb.generate(sb.RANDOM, i, b - a)
            data = data[:a] + mid + data[b:]
        mode = int(rng.choice([sb.MODE_FULL] * 6 + [sb.MODE_LITERALS, sb.MODE_RAW, sb.MODE_RLE]))
        if mode == sb.MODE_RLE:
            data = bytes([data[0] if data else 0]) * n
        if mode == sb.MODE_LITERALS and n > 131072:
            mode = sb.MODE_FULL
        if rng.random() < 0.5:
            sb.set_content_checksum(True)
        frames.append(sb.compress(data, mode)[0])
        sb.set_content_checksum(False)
        want.append(data)
    for c in ctxs:
        outs, sts = z.decode_frames(frames, c)
        for j, (o, w, s) in enumerate(zip(outs, want, sts)):
            if s != 0 or o != w:
                bad += 1
                if bad <= 5:
                    print("MISMATCH frame", done + j, "status", s, "len", len(w), flush=True)
    done += len(frames)
    print(f"{done} frames, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", done, "frames x", len(ctxs), "configurations")
sys.exit(0 if bad == 0 else 1)
