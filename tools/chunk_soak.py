#!/usr/bin/env python3
"""GPU fuzz soak of the chunk path (mzd_fstream_*; not part of the test suite): corpus and synthetic frames with random byte flips /
truncations, each through a FrameStream with a random chunk size and the source arriving in random pieces.  The device must never
fault; a frame the stream decodes must be one the oracle decodes to the same bytes; a frame the oracle rejects must end in an error
or (a truncation) in a stream that wants more bytes.  Where the oracle accepts and the stream does not, only the checks the
reference does not make: a declared content size that is wrong (15), a block that regenerates more than 128 KiB (12), the documented
limits (16), and an offset that reaches behind the frame's declared window (14: the whole-frame path and the oracle have all of the
frame to copy from, a decoder that keeps the window does not -- ringbuffer.go:198-225 wraps there).
usage: python tools/chunk_soak.py [n_mutations] [seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb
from tests.oracle_binding import load_oracle

n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = load_oracle()
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
manifest = json.load(open(os.path.join(golden, "manifest.json")))
pool = []
for nm in sorted(manifest):
    b = open(os.path.join(golden, nm + ".zst"), "rb").read()
    if 24 <= len(b) <= 400000:
        pool.append(b)
for i in range(24):
    pool.append(sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 5900 + i, int(rng.integers(200, 1500000))))[0])
ctxs = [("default", z.Context(0)), ("k_exec_c", z.Context(0, exec_variant=5)), ("block mode, jobs of four", z.Context(0, exec_variant=4))]
dst = np.empty(1 << 20, dtype=np.uint8)


def declared_window(f):
    if len(f) < 6 or f[:4] != b"\x28\xb5\x2f\xfd":
        return None
    fhd = f[4]
    if fhd & 0x20:
        # single segment: the window is the declared content size (frame.go:49-61; 1, 2, 4 or 8 bytes behind the dictionary id)
        n = (1, 2, 4, 8)[fhd >> 6]
        p = 5 + (0, 1, 2, 4)[fhd & 3]
        if len(f) < p + n:
            return None
        return int.from_bytes(f[p:p + n], "little") + (256 if n == 2 else 0)
    wd = f[5]
    base = 1 << (10 + (wd >> 3))
    return base + (base >> 3) * (wd & 7)


def through_stream(f, c, chunk, piece):
    """-> (status, bytes or None, wants_more)"""
    fs = z.FrameStream(c, chunk)
    src = np.frombuffer(f, dtype=np.uint8)
    out = bytearray()
    pos, have = 0, min(len(f), piece)
    try:
        while not fs.done:
            used, made = fs.next(src[pos:have], dst)
            pos += used
            out += dst[:made].tobytes()
            if used == 0 and made == 0 and not fs.done:
                if have >= len(f):
                    return 0, None, True
                have = min(len(f), have + piece)
            elif pos >= have:
                have = min(len(f), have + piece)
        return 0, bytes(out), False
    except z.MzdError as e:
        return e.code, None, False
    finally:
        fs.close()


bad = done = n_ok = n_err = n_more = n_chunks_gt1 = 0
t0 = time.time()


def keep(f, what):
    """a frame the stream and the oracle disagree on goes to gpurun_out/ with what was asked of the stream"""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"chunk_soak_bad_{seed}_{done}.zst"), "wb") as fh:
        fh.write(f)
    with open(os.path.join(d, f"chunk_soak_bad_{seed}_{done}.json"), "w") as fh:
        json.dump(what, fh)


while done < n_mut:
    b = bytearray(pool[int(rng.integers(len(pool)))])
    r = rng.random()
    if r < 0.1:
        b = b[:int(rng.integers(1, len(b)))]
    elif r < 0.95:
        for p in rng.integers(min(4, len(b) - 1), len(b), size=int(rng.integers(1, 4))):
            b[int(p)] ^= int(rng.integers(1, 256))
    f = bytes(b)
    rc, ref, _, _ = orc.decode_frame(f, cap=8 << 20)
    name, c = ctxs[int(rng.integers(len(ctxs)))]
    chunk = int(rng.choice([131072, 262144, 1 << 20]))
    piece = int(rng.choice([1 << 30, 50000, 4000]))
    w = declared_window(f)
    if w is not None and w > (64 << 20):
        done += 1
        continue  # (a flipped window descriptor: gigabytes of slab per mutation say nothing new)
    st, out, more = through_stream(f, c, chunk, piece)
    done += 1
    if more:
        n_more += 1
        if rc == 0:
            bad += 1
            print(f"DISAGREE [{name}]: oracle ok, the stream wants more bytes, len", len(f), flush=True)
            keep(f, {"ctx": name, "chunk": chunk, "piece": piece, "what": "wants more"})
    elif st == 0:
        n_ok += 1
        n_chunks_gt1 += 1 if len(out) > chunk else 0
        if rc != 0 or out != ref:
            bad += 1
            print(f"DISAGREE [{name}]: stream ok ({len(out)} bytes), oracle rc", rc, "len", len(f), "chunk", chunk, flush=True)
            keep(f, {"ctx": name, "chunk": chunk, "piece": piece, "what": "stream ok", "oracle_rc": rc, "out_len": len(out)})
    else:
        n_err += 1
        if rc == 0:
            ok = st in (12, 15, 16) or (st == 14 and w is not None and w < len(ref))
            if st == 12:
                _, _, _, tr = orc.decode_frame(f, cap=8 << 20, want_trace=True)
                ok = max((bb["out_end"] - bb["out_begin"] for bb in tr["blocks"]), default=0) > 131072
            if not ok:
                bad += 1
                print(f"DISAGREE [{name}]: oracle ok, stream status", st, "len", len(f), "chunk", chunk, "window", w, flush=True)
                keep(f, {"ctx": name, "chunk": chunk, "piece": piece, "what": "stream status", "status": st, "window": w})
    if done % 500 == 0:
        print(f"{done} mutations: {n_ok} decoded ({n_chunks_gt1} in more than one chunk), {n_err} errors, {n_more} cut, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"{done} mutations: {n_ok} decoded ({n_chunks_gt1} in more than one chunk), {n_err} errors, {n_more} cut, {bad} bad, {time.time() - t0:.0f} s")
print("CHUNK SOAK", "OK" if bad == 0 and n_chunks_gt1 > 0 else "FAILED")
sys.exit(0 if bad == 0 and n_chunks_gt1 > 0 else 1)
