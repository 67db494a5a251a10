#!/usr/bin/env python3
"""GPU box: ONE large frame through mzd_fstream_* (a frame in chunks of whole blocks: the device keeps the frame's window and nothing
else) against the whole-frame path on the same bytes -- host to host, pinned buffers -- with the device memory each of them takes.
usage: python tools/fstream_bench.py [frame MiB = 1024] [window log = 23] [chunk MiB, ... = 16,64,256] [cursor threads = 0]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z  # noqa: E402
from tools import synth_binding as sb  # noqa: E402
from tests.test_gpu_chunks import with_window  # noqa: E402

frame_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wlog = int(sys.argv[2]) if len(sys.argv) > 2 else 23
chunks = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "16,64,256").split(",")]
cursor_threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # (0: the library's default, up to eight; 1: the serial walk)
n = frame_mib << 20
sb.set_max_offset(1 << wlog)
blob, off, ln, ck, ns = sb.make_batch(4, 31, 1, frame_bytes=n, threads=8)
sb.set_max_offset(0)
comp = with_window(blob[int(off[0]):int(off[0] + ln[0])].tobytes(), wlog)
want = int(ck[0])
ctx = z.default_context()


def used_mib():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2 ** 20


src = z.PinnedBuffer(len(comp))
src.a[:] = np.frombuffer(comp, dtype=np.uint8)
base = used_mib()
for c in chunks:
    dst = z.PinnedBuffer(c << 20)
    out = z.PinnedBuffer(n)
    best, peak = None, 0.0
    for rep in range(3):
        fs = z.FrameStream(ctx, c << 20, threads=cursor_threads)
        t0 = time.time()
        pos = made_total = calls = 0
        first = None
        while not fs.done:
            used, made = fs.next(src.a[pos:], dst.a)
            assert used or made
            out.a[made_total:made_total + made] = dst.a[:made]  # (what a reader's Read does with the chunk)
            pos += used
            made_total += made
            calls += 1
            if first is None and made:
                first = time.time() - t0
            peak = max(peak, used_mib() - base)
        t1 = time.time()
        stages = fs.timing()
        fs.close()
        assert made_total == n and sb.checksum64(out.a[:n].tobytes()) == want
        best = min(best or 1e9, t1 - t0)
    print(json.dumps({"bench": "mzd_fstream_next", "frame_MiB": frame_mib, "compressed_MiB": round(len(comp) / 2 ** 20, 1), "window_log": wlog,
                      "chunk_MiB": c, "cursor_threads": cursor_threads, "calls": calls, "seconds": round(best, 4), "out_GBs": round(n / best / 1e9, 2),
                      "first_bytes_after_ms": round(first * 1e3, 2), "device_MiB_between_calls": round(peak, 1),
                      "host_ms_per_stage_last_run": stages}), flush=True)
    dst.free()
    out.free()
if n <= (1 << 31) - (1 << 20):
    z.decode_frames([comp], ctx)
    base = used_mib()
    t0 = time.time()
    outs, sts = z.decode_frames([comp], ctx)
    t1 = time.time()
    assert sts == [0] and sb.checksum64(outs[0]) == want
    print(json.dumps({"bench": "decode_frames, the frame whole", "frame_MiB": frame_mib, "seconds": round(t1 - t0, 4),
                      "out_GBs": round(n / (t1 - t0) / 1e9, 2), "note": "device memory: the frame's output + 3 planes of it + scratch"}))
