#!/usr/bin/env python3
"""GPU fuzz soak (not part of the test suite): corpus and synthetic frames with random byte flips /
truncations, decoded in batches.  The device must never fault; a frame it reports as decoded must
be one the oracle decodes to the same bytes; a frame the oracle rejects must carry a status.
usage: python tools/fuzz_soak.py [n_mutations] [seed]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb
from tests.oracle_binding import load_oracle

n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = load_oracle()
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
names = sorted(json.load(open(os.path.join(golden, "manifest.json"))))
base = [open(os.path.join(golden, n + ".zst"), "rb").read() for n in names]
base = [b for b in base if 24 <= len(b) <= 200000]
for i in range(40):
    base.append(sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 900 + i, int(rng.integers(200, 150000))))[0])
for i in range(12):  # multi-block frames: the scan / pattern passes / fix-up walk of block mode on damaged input
    base.append(sb.compress(sb.generate(int(rng.choice([sb.TEXT, sb.EXP])), 1900 + i, int(rng.integers(300000, 1200000))))[0])
ctxs = [z.Context(0, seq_variant=0, verify_checksum=True), z.Context(0, seq_variant=1), z.Context(0, seq_variant=3, huf_variant=2), z.Context(0, huf_variant=3),
        z.Context(0, exec_variant=2), z.Context(0, exec_variant=3), z.Context(0, exec_variant=4, huf_variant=2)]
bad = done = n_ok = 0
t0 = time.time()
while done < n_mut:
    frames = []
    for _ in range(min(1000, n_mut - done)):
        b = bytearray(base[int(rng.integers(len(base)))])
        r = rng.random()
        if r < 0.1:
            b = b[:int(rng.integers(1, len(b)))]
        else:
            for pos in rng.integers(5, len(b), size=int(rng.integers(1, 4))):
                b[int(pos)] ^= int(rng.integers(1, 256))
        frames.append(bytes(b))
    for ci, c in enumerate(ctxs):
        outs, sts = z.decode_frames(frames, c)
        for f, o, s in zip(frames, outs, sts):
            rc, want, _, _ = orc.decode_frame(f, cap=4 << 20)
            if s == 0:
                n_ok += 1
                if rc != 0 or o != want:
                    bad += 1
                    print("DISAGREE: device ok, oracle rc", rc, "len", len(f), flush=True)
            elif rc != 0:
                pass
            elif s == 12:
                # MZD_ERR_CORRUPT_SIZES with an oracle that accepts: legitimate only when a block regenerates more
                # than Block_Maximum_Size (128 KiB) -- the reference has no such check, the device path does
                _, _, _, tr = orc.decode_frame(f, cap=4 << 20, want_trace=True)
                if max((b["out_end"] - b["out_begin"] for b in tr["blocks"]), default=0) <= 131072:
                    bad += 1
                    print("DISAGREE: oracle ok, device status 12 without an oversized block, len", len(f), flush=True)
            elif s not in (15, 16, 18):  # oracle accepts, device rejects: only the content-size / checksum checks the reference
                # does not make (a corrupted Frame_Content_Size, MZD_ERR_DST_FULL) and the documented limits may do that
                bad += 1
                print("DISAGREE: oracle ok, device status", s, "len", len(f), flush=True)
    done += len(frames)
    print(f"{done} mutations, {n_ok} decoded, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("FUZZ SOAK", "OK" if bad == 0 else "FAILED")
sys.exit(0 if bad == 0 else 1)
