#!/usr/bin/env python3
"""Debug helper (GPU box): decode the golden corpus on the device, compare with the oracle and
print where the first difference of every failing frame falls (block, kind, sequence)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from tests.oracle_binding import load_oracle

orc = load_oracle()
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
manifest = json.load(open(os.path.join(golden, "manifest.json")))
names = sorted(manifest)
sv = int(os.environ.get("SEQ_VARIANT", "0")); et = int(os.environ.get("EXEC_THREADS", "0"))
ctx = z.Context(0, seq_variant=sv, exec_threads=et)
frames = [open(os.path.join(golden, n + ".zst"), "rb").read() for n in names]
outs, sts = z.decode_frames(frames, ctx)
nbad = 0
for n, f, o, s in zip(names, frames, outs, sts):
    rc, want, _, tr = orc.decode_frame(f, cap=manifest[n]["length"] + 64, want_trace=True)
    if s == 0 and o == want:
        continue
    nbad += 1
    if s != 0:
        print(n, "status", s, z.strerror(s)); continue
    d = next((i for i in range(min(len(o), len(want))) if o[i] != want[i]), None)
    ndiff = sum(1 for a, b in zip(o, want) if a != b)
    print(f"{n}: len {len(o)}/{len(want)} first diff at {d}, {ndiff} bytes differ")
    seq_base = 0
    for bi, b in enumerate(tr["blocks"]):
        if b["out_begin"] <= d < b["out_end"]:
            print("   block", bi, b)
            # locate the sequence
            pos = b["out_begin"]
            for si in range(seq_base, seq_base + b["n_seq"]):
                ll, ml, ofv, off = tr["seqs"][si]
                if pos + ll + ml > d:
                    print(f"   seq {si - seq_base}/{b['n_seq']}: LL={ll} ML={ml} ofv={ofv} off={off} starts at {pos} "
                          f"(block-rel {pos - b['out_begin']}), diff is in {'literals' if d < pos + ll else 'match'}")
                    break
                pos += ll + ml
            else:
                print("   diff in trailing literals at block-rel", d - b["out_begin"])
            print("   got ", o[max(0, d - 8):d + 24].hex())
            print("   want", want[max(0, d - 8):d + 24].hex())
        if b["block_type"] == 2:
            seq_base += b["n_seq"]
print("failing frames:", nbad, "of", len(names))
