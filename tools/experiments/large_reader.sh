# one LARGE frame end to end: the reader mirror, decode_frames with either planning route, the blob form
timeout 900 python - <<'PY'
import sys, time, io
sys.path.insert(0, '.')
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import api
from tools import synth_binding as sb
for mib in (64, 256):
    data = sb.generate(sb.TEXT, 5, mib << 20)
    comp = sb.compress(data, sb.MODE_FULL)[0]
    ctx = z.Context(0)
    for dp in (False, True):
        z.decode_frames([comp], ctx, device_plan=dp)  # warm
        t0 = time.time()
        outs, sts = z.decode_frames([comp], ctx, device_plan=dp)
        t1 = time.time()
        assert sts == [0] and outs[0] == data
        t2 = time.time()
        out, lay, ol, sts = api.decode_frames_blob([comp], ctx, device_plan=dp)
        t3 = time.time()
        assert sts == [0] and out[int(lay[0]):int(lay[0]) + int(ol[0])].tobytes() == data
        print(f"{mib} MiB frame, device_plan={dp}: decode_frames {1e3 * (t1 - t0):.1f} ms, decode_frames_blob {1e3 * (t3 - t2):.1f} ms host to host "
              f"({(mib << 20) / (t3 - t2) / 1e6:.0f} MB/s)", flush=True)
    r = z.FrameReader(io.BytesIO(comp), ctx)
    r.Read(1)
    t0 = time.time()
    r = z.FrameReader(io.BytesIO(comp), ctx)
    first = r.Read(1 << 20)
    t1 = time.time()
    rest = r.read()
    t2 = time.time()
    assert first + rest == data
    print(f"{mib} MiB frame through FrameReader: first MiB after {1e3 * (t1 - t0):.1f} ms, all of it after {1e3 * (t2 - t0):.1f} ms", flush=True)
    t0 = time.time()
    got = z.FrameReader(io.BytesIO(comp), ctx).read()
    t1 = time.time()
    assert got == data
    buf = bytearray(mib << 20)  # (zero-filled, i.e. touched: a consumer's reused buffer)
    tt = []
    for _ in range(3):
        t2 = time.time()
        r = z.FrameReader(io.BytesIO(comp), ctx)
        k = r.readinto(buf)
        t3 = time.time()
        assert k == len(buf)
        tt.append(1e3 * (t3 - t2))
    assert buf == data
    print(f"{mib} MiB frame through FrameReader: read() {1e3 * (t1 - t0):.1f} ms, readinto(caller's buffer) " + " / ".join(f"{x:.1f}" for x in tt) + " ms", flush=True)
    ctx.timing_reset(True)
    api.decode_frames_blob([comp], ctx)
    ctx.sync()
    print("   device pass per kernel (ms):", {k: round(v, 3) for k, v in ctx.kernel_ms().items()})
    ctx.close()
PY
