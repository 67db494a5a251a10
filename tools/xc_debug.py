#!/usr/bin/env python3
"""Debug helper (GPU box): k_exec_c (exec_variant 5) against k_exec_b (2) on a small synthetic batch and the corpus -- where the
first differing byte of a frame lies, and which sequence made it (from the oracle's trace)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparkzstd_amd as z
from tools import synth_binding as sb
from tests.oracle_binding import load_oracle

orc = load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=4)
frames = [bytes(blob[int(o):int(o + l)]) for o, l in zip(off, ln)]
golden = os.path.join(ROOT, "tests", "golden", "decodecorpus")
names = sorted(json.load(open(os.path.join(golden, "manifest.json"))))
frames += [open(os.path.join(golden, nm + ".zst"), "rb").read() for nm in names]
labels = ["synth%d" % i for i in range(n)] + names
ref, rs = z.decode_frames(frames, z.Context(0, exec_variant=2))
got, gs = z.decode_frames(frames, z.Context(0, exec_variant=5))
bad = 0
for i, (a, b, sa, sg) in enumerate(zip(ref, got, rs, gs)):
    if sa != sg or a != b:
        bad += 1
        if bad > 6:
            continue
        print(labels[i], "status", sa, sg, "len", len(a or b""), len(b or b""))
        if a is None or b is None:
            continue
        A, B = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
        m = min(len(A), len(B))
        d = np.nonzero(A[:m] != B[:m])[0]
        print("  differing bytes", len(d), "first", d[:12], "last", d[-3:] if len(d) else [])
        rc, out, cons, tr = orc.decode_frame(frames[i], cap=len(a) + 64, want_trace=True)
        pos = 0
        first = int(d[0]) if len(d) else -1
        blocks = tr["blocks"]
        # walk the sequences to find the one covering `first`
        bi = 0
        seqs = tr["seqs"]
        p = 0
        k = 0
        for blk in blocks:
            p = int(blk["out_begin"])
            nsq = int(blk["n_seq"])
            for j in range(nsq):
                ll, ml, _, ro = seqs[k + j]
                if p <= first < p + ll:
                    print("  block", bi, "seq", j, "tile", j // 64, "lane", j % 64, "LITERAL run at", p, "LL", ll, "ML", ml, "off", ro, "byte", first - p)
                if p + ll <= first < p + ll + ml:
                    print("  block", bi, "seq", j, "tile", j // 64, "lane", j % 64, "MATCH at", p + ll, "LL", ll, "ML", ml, "off", ro, "byte", first - p - ll,
                          "blockstart", int(blk["out_begin"]))
                p += ll + ml
            k += nsq
            if p <= first < int(blk["out_end"]):
                print("  block", bi, "trailing literals at", p)
            bi += 1
        print("  want", A[max(first - 4, 0):first + 12], "got", B[max(first - 4, 0):first + 12])
print("frames", len(frames), "bad", bad)
