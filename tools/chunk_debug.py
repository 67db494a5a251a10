"""Debugging aid for frames in chunks (mzd_fstream_*): every corpus frame through FrameStream against the whole-frame path; for a frame
that differs: the first differing byte, the chunk it lies in and that chunk's blocks.   python tools/chunk_debug.py [chunk_bytes] [max frames]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparkzstd_amd as z  # noqa: E402
from tests.test_gpu_chunks import stream_decode  # noqa: E402


def main():
    chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "decodecorpus")
    names = sorted(json.load(open(os.path.join(d, "manifest.json"))))[:limit]
    ctx = z.Context(0)
    bad = 0
    for nm in names:
        comp = open(os.path.join(d, nm + ".zst"), "rb").read()
        (want,), sts = z.decode_frames([comp], ctx)
        marks = []
        try:
            got, chunks, used, w = stream_decode(comp, ctx, chunk, on_chunk=marks.append)
        except Exception as e:  # noqa: BLE001
            print(nm, "EXCEPTION", e, "after", marks[-3:])
            bad += 1
            continue
        if got == want:
            continue
        bad += 1
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        n = min(len(a), len(b))
        diff = np.nonzero(a[:n] != b[:n])[0]
        first = int(diff[0]) if len(diff) else n
        ci = next(i for i, m in enumerate(marks) if m > first) if first < len(got) else len(marks)
        print(nm, "len", len(got), len(want), "window", w, "chunks", chunks, "first diff at", first, "of", len(diff), "in chunk", ci,
              "chunk range", marks[ci - 1] if ci else 0, marks[ci] if ci < len(marks) else None)
        # the blocks of the frame
        p = z.Plan()
        p.add_frame(comp)
        bt = p.finalize()
        desc = [(bt.blocks[i].type, bt.blocks[i].lit_type, bt.blocks[i].lit_regen, bt.blocks[i].n_seq, bt.blocks[i].size) for i in range(bt.n_blocks)]
        print("   blocks (type, lit_type, lit_regen, n_seq, size):", desc[max(0, ci - 2):ci + 2], "n_blocks", bt.n_blocks)
        print("   got ", bytes(a[first:first + 16]).hex(), " want", bytes(b[first:first + 16]).hex(), "diff positions", diff[:12].tolist())
        p.close()
    print("frames that differ:", bad, "of", len(names))


main()
