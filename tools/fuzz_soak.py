#!/usr/bin/env python3
"""GPU fuzz soak (not part of the test suite): corpus and synthetic frames with random byte flips /
truncations, decoded in batches.  The device must never fault; a frame it reports as decoded must
be one the oracle decodes to the same bytes; a frame the oracle rejects must carry a status.

Two pools of base frames, batches alternate between them:
  small   frames that regenerate at most 128 KiB (what BASELINE's configs are made of): the library's own choice for such a
          batch is k_huf first, k_seq_q4, k_exec_c; contexts: the default, k_seq / k_seq_pipe with k_exec_c (exec_variant 5),
          k_exec_b, k_exec, k_huf_w for every stream (round 6; the variants that are second implementations come from libmzd_test.so)
  mixed   everything, frames of up to 1.2 MiB in several blocks: 8-byte records, the serial walk and block mode (exec_variant
          3, 4; 4 also with mzd_debug_force_fixup_bail: the rescue launch of the fix-up walk on damaged input)
Both pools hold PERIODIC content too (runs that feed themselves at periods of 1 to 40 bytes: the in-pass resolver of
k_exec_c -- ringbuffer.go:242-277, the overlap case -- and sequences with a literal length of 0 in front of repeat
codes, sequence_execution.go:84-101).
usage: python tools/fuzz_soak.py [n_mutations] [seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
from tests.oracle_binding import load_oracle

n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# hunting a hang: batches up to `first_batch` are generated (the random sequence stays the soak's) but not decoded, and from then on
# every batch is left in gpurun_out/fuzz_soak_batch.pkl with the context that is about to take it, so that a run cut by `timeout`
# leaves behind what it was decoding.  usage: fuzz_soak.py <n> <seed> <first_batch>
first_batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
if first_batch is not None:
    import faulthandler
    faulthandler.dump_traceback_later(100, repeat=True)  # a stall: where the host is (which library call, or the oracle)
rng = np.random.default_rng(seed)
orc = load_oracle()
L = _lib.load()
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
manifest = json.load(open(os.path.join(golden, "manifest.json")))
names = sorted(manifest)


def periodic(seed_, n):
    """runs of a short period between stretches of text: matches with offset < length, repeat codes behind a zero literal length"""
    r = np.random.default_rng(seed_)
    out = bytearray()
    text = sb.generate(sb.TEXT, seed_, n)
    while len(out) < n:
        p = int(r.integers(1, 41))
        unit = bytes(r.integers(97, 123, size=p, dtype=np.uint8))
        out += unit * int(r.integers(2, 400))
        a = int(r.integers(0, max(1, n - 300)))
        out += text[a:a + int(r.integers(0, 300))]
    return bytes(out[:n])


small, mixed = [], []
for nm in names:
    b = open(os.path.join(golden, nm + ".zst"), "rb").read()
    if 24 <= len(b) <= 200000:
        mixed.append(b)
        if manifest[nm]["length"] <= 131072:
            small.append(b)
for i in range(40):
    f = sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 900 + i, int(rng.integers(200, 131072))))[0]
    small.append(f)
    mixed.append(f)
for i in range(16):
    f = sb.compress(periodic(2900 + i, int(rng.integers(2000, 131072))))[0]
    small.append(f)
    mixed.append(f)
for i in range(12):  # multi-block frames: the scan / pattern passes / fix-up walk of block mode on damaged input
    mixed.append(sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 1900 + i, int(rng.integers(300000, 1200000))))[0])
for i in range(4):
    mixed.append(sb.compress(periodic(3900 + i, int(rng.integers(300000, 900000))))[0])

ctx_small = [("default", z.Context(0)), ("k_seq + k_exec_c", z.Context(0, seq_variant=1, exec_variant=5)),
             ("k_seq_pipe + k_huf_seg + k_exec_c", z.Context(0, seq_variant=3, huf_variant=2, exec_variant=5)),
             ("k_exec_b", z.Context(0, exec_variant=2)), ("k_exec + checksum", z.Context(0, exec_variant=1, verify_checksum=True)),
             ("k_huf_w for every stream", z.Context(0, huf_variant=4))]
ctx_mixed = [("default + checksum", z.Context(0, verify_checksum=True)), ("k_seq", z.Context(0, seq_variant=1)), ("k_huf first", z.Context(0, huf_variant=3)),
             ("k_exec_c", z.Context(0, exec_variant=5)), ("k_exec_b", z.Context(0, exec_variant=2)), ("block mode", z.Context(0, exec_variant=3)),
             ("block mode, jobs of four", z.Context(0, exec_variant=4, huf_variant=2)), ("k_huf_w for every stream", z.Context(0, huf_variant=4)),
             ("block mode, fix-up rescue", z.Context(0, exec_variant=4))]
bail_ctx = ctx_mixed[-1][1]


def decode(frames, c):
    """z.decode_frames, and which kernels the pass took (ResidentBatch.last_pass)"""
    rb, lay, out_len, sts = z.api.decode_frames_resident(frames, c)
    try:
        lp = rb.pass_flags
        out, _, _ = rb.download()
    finally:
        rb.free()
    return [out[int(lay[i]):int(lay[i]) + int(out_len[i])].tobytes() if sts[i] == 0 else None for i in range(len(frames))], sts, lp


n_pass = {"block mode": 0, "k_exec_c": 0, "k_exec_b": 0}
bad = done = n_ok = n_batches = 0
t0 = time.time()
while done < n_mut:
    pool, ctxs = (small, ctx_small) if n_batches % 2 == 0 else (mixed, ctx_mixed)
    n_batches += 1
    frames = []
    lo = 5
    for _ in range(min(1000, n_mut - done)):
        b = bytearray(pool[int(rng.integers(len(pool)))])
        r = rng.random()
        if r < 0.1:
            b = b[:int(rng.integers(1, len(b)))]
        else:
            for pos in rng.integers(min(lo, len(b) - 1), len(b), size=int(rng.integers(1, 4))):
                b[int(pos)] ^= int(rng.integers(1, 256))
        frames.append(bytes(b))
    if first_batch is not None and n_batches < first_batch:
        for name, c in ctxs:
            if c is bail_ctx:
                rng.integers(1, 6)
        done += len(frames)
        continue
    want = [orc.decode_frame(f, cap=4 << 20) for f in frames]
    for name, c in ctxs:
        if c is bail_ctx:
            bail_step = int(rng.integers(1, 6))
            assert L.mzd_debug_force_fixup_bail(c._c, bail_step) == 0
        if first_batch is not None:
            import pickle
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            if name == ctxs[0][0]:  # (the frames once per batch, the context that is about to take them every time)
                with open(os.path.join(ROOT, "gpurun_out", "fuzz_soak_batch.pkl"), "wb") as fh:
                    pickle.dump({"batch": n_batches, "frames": frames}, fh)
            with open(os.path.join(ROOT, "gpurun_out", "fuzz_soak_now.json"), "w") as fh:
                json.dump({"batch": n_batches, "ctx": name, "bail_step": bail_step if c is bail_ctx else 0, "t": time.time() - t0}, fh)
        outs, sts, lp = decode(frames, c)
        for k, bit in (("block mode", _lib.MZD_PASS_BLOCK_MODE), ("k_exec_c", _lib.MZD_PASS_EXEC_C), ("k_exec_b", _lib.MZD_PASS_EXEC_B)):
            n_pass[k] += 1 if lp & bit else 0
        for f, o, s, (rc, ref, _, _) in zip(frames, outs, sts, want):
            if s == 0:
                n_ok += 1
                if rc != 0 or o != ref:
                    bad += 1
                    print(f"DISAGREE [{name}]: device ok, oracle rc", rc, "len", len(f), flush=True)
            elif rc != 0:
                pass
            elif s == 12:
                # MZD_ERR_CORRUPT_SIZES with an oracle that accepts: legitimate only when a block regenerates more
                # than Block_Maximum_Size (128 KiB) -- the reference has no such check, the device path does
                _, _, _, tr = orc.decode_frame(f, cap=4 << 20, want_trace=True)
                if max((b["out_end"] - b["out_begin"] for b in tr["blocks"]), default=0) <= 131072:
                    bad += 1
                    print(f"DISAGREE [{name}]: oracle ok, device status 12 without an oversized block, len", len(f), flush=True)
            elif s not in (15, 16, 18):  # oracle accepts, device rejects: only the content-size / checksum checks the reference
                # does not make (a corrupted Frame_Content_Size, MZD_ERR_DST_FULL) and the documented limits may do that
                bad += 1
                print(f"DISAGREE [{name}]: oracle ok, device status", s, "len", len(f), flush=True)
    done += len(frames)
    print(f"{done} mutations x {len(ctxs)} contexts ({'small' if pool is small else 'mixed'} pool), {n_ok} decoded, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("passes by kernel choice:", n_pass)
print("FUZZ SOAK", "OK" if bad == 0 and min(n_pass.values()) > 0 else "FAILED")
sys.exit(0 if bad == 0 and min(n_pass.values()) > 0 else 1)
