#!/usr/bin/env python3
"""GPU box: a batch of MIXED frame sizes (4 KiB ... 16 MiB, text-like) through the library's own choice of execution kernel (0)
and every forced one: the choice should be the best or close to it.  usage: python tools/experiments/r4_mixed.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

def batch(spec):
    frames, cks = [], []
    for count, size in spec:
        blob, off, ln, ck, ns = sb.make_batch(4, 900 + size % 977, count, frame_bytes=size, threads=8)
        frames += [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
        cks += [(size, int(k)) for k in ck]
    return frames, cks

for name, spec in [("4000 x 4 KiB + 2000 x 128 KiB + 200 x 1 MiB + 4 x 16 MiB", [(4000, 4096), (2000, 131072), (200, 1 << 20), (4, 16 << 20)]),
                   ("2000 x 128 KiB + 2 x 64 MiB", [(2000, 131072), (2, 64 << 20)]),
                   ("20000 x 32 KiB + 64 x 4 MiB", [(20000, 32768), (64, 4 << 20)])]:
    frames, cks = batch(spec)
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    ln = np.array([len(f) for f in frames], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    for v in (0, 5, 2, 1, 4):
        ctx = z.Context(0, exec_variant=v)
        plan = z.Plan(device_tables=True)
        plan.add_frames(blob, off, ln, threads=8)
        b = plan.finalize()
        rb = ctx.upload(b)
        rb.run(); ctx.sync()
        ctx.timing_reset(True)
        for _ in range(3):
            rb.run()
        ctx.sync()
        ms = ctx.kernel_ms()
        out, st, ol = rb.download()
        ok = bool((st == 0).all())
        for i in (0, len(frames) // 2, len(frames) - 1):
            o = int(b.frames[i].out_offset)
            ok = ok and sb.checksum64(out[o:o + cks[i][0]].tobytes()) == cks[i][1]
        print(f"{name}: exec_variant {v}: path {ms.get('path', 0):.2f} ms  " + " ".join(f"{k}={x:.2f}" for k, x in ms.items() if k != 'path') + f"  ok={ok}", flush=True)
        rb.free(); plan.close(); ctx.close()
