#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the streaming path (mzd_stream_*): host frames in -> regenerated
frames back in (pinned) host memory, batches pipelined through `depth` device slots.  Not the headline
metric (bench.py measures the resident hot path); this is the number DESIGN.md quotes for SURVEY 8f #4.
usage: python tools/stream_bench.py [frames_per_batch=8192] [n_batches=12] [depths=1,2,3]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

per = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
depths = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
frame_bytes = 131072
blob, off, ln, cks, _ = sb.make_batch(4, 0, per, frame_bytes, threads=os.cpu_count() or 8)
blob = np.ascontiguousarray(blob[:int(off[-1] + ln[-1])])
rc, off2, ln2, ob, total = z.split_frames(blob)  # the caller's view: one buffer of concatenated frames
assert rc == 0 and (off2 == off).all() and (ln2 == ln).all() and total == per * frame_bytes
ctx = z.Context(0)
res = {"frames_per_batch": per, "batches": nb, "compressed_bytes_per_batch": int(blob.size), "out_bytes_per_batch": total, "depth": {}}
for depth in depths:
    pin_in = [z.PinnedBuffer(blob.size) for _ in range(depth)]
    pin_out = [z.PinnedBuffer(total) for _ in range(depth)]
    for p in pin_in:
        p.a[:] = blob
    st = z.Stream(ctx, depth=depth)
    for warm in (True, False):
        t0 = time.perf_counter()
        inflight, ok = [], True
        for k in range(nb if not warm else depth + 1):
            if len(inflight) == depth:
                t, o = inflight.pop(0)
                s, l, oo = st.wait(t)
                ok = ok and bool((s == 0).all() and (l == frame_bytes).all())
            slot = k % depth
            inflight.append((st.submit(pin_in[slot].a, off, ln, pin_out[slot].a), pin_out[slot]))
        for t, o in inflight:
            s, l, oo = st.wait(t)
            ok = ok and bool((s == 0).all() and (l == frame_bytes).all())
        dt = time.perf_counter() - t0
    # content check of the last batch: the synthetic generator's checksum of every frame
    words = frame_bytes // 8
    w = (2 * np.arange(words, dtype=np.uint64) + 1)
    o64 = pin_out[(nb - 1) % depth].a[:per * frame_bytes].view(np.uint64).reshape(per, words)
    got = (o64[:256] * w).sum(axis=1, dtype=np.uint64)
    ok = ok and bool((got == cks[:256].view(np.uint64)).all())
    res["depth"][depth] = {"seconds": round(dt, 4), "out_GBs": round(nb * total / dt / 1e9, 2),
                           "in_GBs": round(nb * blob.size / dt / 1e9, 2), "ms_per_batch": round(dt / nb * 1e3, 2), "ok": ok}
    st.close()
    for p in pin_in + pin_out:
        p.free()
print(json.dumps(res))
