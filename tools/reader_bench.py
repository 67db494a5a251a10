#!/usr/bin/env python3
"""GPU box: what a consumer gets that uses the reference-shaped FrameReader ONE FRAME AT A TIME (one device batch,
i.e. plan + upload + six launches + download, per frame) against the batch entry on the same frames.  The reader
mirrors sparkzstd's API; throughput comes from batching (DecodeFrames / mzd_stream_*), and this tool puts the
number on the difference.  usage: python tools/reader_bench.py [n_frames]"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from tools import synth_binding as sb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
blob, off, ln, cks, nseq = sb.make_batch(4, 0, n, 131072, threads=8)
frames = [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
ctx = z.default_context()
z.decode_frames(frames[:4], ctx)  # warm-up: context, kernels
t0 = time.time()
total = 0
for f in frames:
    r = z.FrameReader(io.BytesIO(f))
    total += len(r.read())
t1 = time.time()
outs, sts = z.decode_frames(frames, ctx)
t2 = time.time()
assert all(s == 0 for s in sts) and total == sum(len(o) for o in outs)
print(f"FrameReader, one frame per call: {n / (t1 - t0):.0f} frames/s, {total / (t1 - t0) / 1e6:.0f} MB/s "
      f"({(t1 - t0) / n * 1e3:.2f} ms per frame)")
print(f"decode_frames, one batch of {n}: {n / (t2 - t1):.0f} frames/s, {total / (t2 - t1) / 1e6:.0f} MB/s (host to host, pageable memory)")
# the reader that batches: the same reader-shaped consumer (Reset per frame, Read until EOF), the frames to come known to it
for look in (64, 256):
    r = z.BatchFrameReader((io.BytesIO(f) for f in frames), lookahead=look)
    t3 = time.time()
    tot = 0
    while r.Reset():
        while True:
            d = r.Read(1 << 20)
            if not d:
                break
            tot += len(d)
    t4 = time.time()
    assert tot == total
    print(f"BatchFrameReader, lookahead {look}: {n / (t4 - t3):.0f} frames/s, {tot / (t4 - t3) / 1e6:.0f} MB/s (Reset + Read per frame, host to host)")

# one LARGE frame through the reader (the reference's own usage: cmd/sparkzstd/main.go:59,126): the frame's blocks are executed
# side by side on the device (mzd_exec_blk.hip); usage: python tools/reader_bench.py <n_frames> <big_frame_bytes>
if len(sys.argv) > 2:
    big = int(sys.argv[2])
    data = sb.generate(sb.TEXT, 5, big)
    comp = sb.compress(data, sb.MODE_FULL)[0]
    z.FrameReader(io.BytesIO(comp)).read()  # warm-up: the allocations of this size
    t0 = time.time()
    got = z.FrameReader(io.BytesIO(comp)).read()
    t1 = time.time()
    assert got == data
    print(f"FrameReader, one frame of {big >> 20} MiB ({len(comp) >> 20} MiB compressed): {(t1 - t0) * 1e3:.1f} ms = {big / (t1 - t0) / 1e6:.0f} MB/s "
          f"host to host (planning + upload + device pass + download)")
    # the same frame in CHUNKS of whole blocks (ABI 9: the device keeps the frame's window, not the frame; bytes as the source delivers)
    for chunk in (16 << 20, 64 << 20):
        r = z.FrameReader(io.BytesIO(comp), chunk_bytes=chunk)
        assert r.read() == data  # warm-up: the reader's pinned chunk buffer, the allocations of this size
        r.Reset(io.BytesIO(comp))
        buf = bytearray(chunk)
        t0 = time.time()
        first = None
        n = 0
        while True:
            k = r.readinto(buf)
            if not k:
                break
            n += k
            if first is None:
                first = time.time() - t0
        t1 = time.time()
        assert n == big
        print(f"FrameReader(chunk_bytes={chunk >> 20} MiB), readinto: {(t1 - t0) * 1e3:.1f} ms = {big / (t1 - t0) / 1e6:.0f} MB/s host to host, first bytes after "
              f"{first * 1e3:.1f} ms; stream stages (host ms): {getattr(r, 'stream_timing', None)}")
        r.close()
    ctx.timing_reset(True)
    outs, sts = z.decode_frames([comp], ctx)
    ctx.sync()
    assert sts == [0] and outs[0] == data
    print("device pass of that frame, per kernel (ms):", {k: round(v, 3) for k, v in ctx.kernel_ms().items()})
