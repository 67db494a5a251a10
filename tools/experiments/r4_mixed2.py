#!/usr/bin/env python3
"""GPU box: batches that mix BLOCK TYPES (Raw / RLE frames, Huffman-only frames, frames with sequences; BASELINE configs 2, 3, 4)
through the library's own choices and forced ones: the defaults should be the best or close.  usage: python tools/experiments/r4_mixed2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

def batch(spec):
    frames, cks = [], []
    for config, count, size in spec:
        blob, off, ln, ck, ns = sb.make_batch(config, 500 + config, count, frame_bytes=size, threads=8)
        frames += [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
        cks += [(size, int(k)) for k in ck]
    return frames, cks

for name, spec in [("2000 each of configs 2, 3, 4 (128 KiB)", [(2, 2000, 131072), (3, 2000, 131072), (4, 2000, 131072)]),
                   ("8000 config 3 + 8000 config 4 (128 KiB)", [(3, 8000, 131072), (4, 8000, 131072)]),
                   ("1000 config 3 (1 MiB) + 10000 config 4 (32 KiB)", [(3, 1000, 1 << 20), (4, 10000, 32768)])]:
    frames, cks = batch(spec)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(frames))
    frames = [frames[i] for i in perm]
    cks = [cks[i] for i in perm]
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    ln = np.array([len(f) for f in frames], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64)
    for kw in ({}, {"huf_variant": 1}, {"huf_variant": 2}, {"huf_variant": 3}, {"exec_variant": 1}, {"exec_variant": 2}, {"seq_variant": 3}):
        ctx = z.Context(0, **kw)
        plan = z.Plan(device_tables=True)
        plan.add_frames(blob, off, ln, threads=8)
        b = plan.finalize()
        rb = ctx.upload(b)
        rb.run(); ctx.sync()
        ctx.timing_reset(True)
        for _ in range(4):
            rb.run()
        ctx.sync()
        ms = ctx.kernel_ms()
        out, st, ol = rb.download()
        ok = bool((st == 0).all())
        for i in (0, len(frames) // 3, len(frames) - 1):
            o = int(b.frames[i].out_offset)
            ok = ok and sb.checksum64(out[o:o + cks[i][0]].tobytes()) == cks[i][1]
        print(f"{name}: {kw or 'defaults'}: path {ms.get('path', 0):.2f} ms  " + " ".join(f"{k}={x:.2f}" for k, x in ms.items() if k != 'path') + f"  ok={ok}", flush=True)
        rb.free(); plan.close(); ctx.close()
