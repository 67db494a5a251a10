#!/usr/bin/env python3
"""Research probe (CPU, host planner only): does zstd's sequence decoding SELF-SYNCHRONISE?
A decoder started at an arbitrary bit position of a block's sequence bitstream with arbitrary FSE states
follows its own trajectory; if it ever reaches a (bit position, LL/ML/OF state) point of the TRUE trajectory
it is synchronised for good.  If that happens within a few hundred sequences, one block's chain can be
cut into segments decoded by several lanes that SHARE the block's tables in LDS -- the lever against the
LDS capacity x time bound of k_seq_pipe.  usage: python tools/research/sync_probe.py"""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb

LL_BITS = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
ML_BITS = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]


def tables_of(batch, bd):
    def tab(idx):
        d = batch.fse_tables[idx]
        n = 1 << d.acc_log
        cells = [(batch.fse_entries[d.entries_off + i].baseline, batch.fse_entries[d.entries_off + i].nbits,
                  batch.fse_entries[d.entries_off + i].symbol) for i in range(n)]
        return d.acc_log, cells
    return tab(bd.ll_table), tab(bd.of_table), tab(bd.ml_table)


def probe(frame, trials, rng, max_steps=3000):
    p = z.Plan(device_tables=False)
    assert p.add_frame(frame)[0] == 0
    b = p.finalize()
    res = []
    blob = bytes((ctypes.c_uint8 * b.in_size).from_address(ctypes.cast(b.in_, ctypes.c_void_p).value))
    for bi in range(b.n_blocks):
        bd = b.blocks[bi]
        if bd.type != 2 or bd.n_seq < 2000:
            continue
        (alL, cL), (alO, cO), (alM, cM) = tables_of(b, bd)
        data = blob[bd.seq_off:bd.seq_off + bd.seq_size]
        val = int.from_bytes(data, "little")
        nbit = len(data) * 8

        def read(pos, n):  # n bits ending at bit `pos` (pos = index of the next bit to read), MSB first
            if n == 0:
                return 0
            lo = pos - n + 1
            if lo < 0:
                return ((val & ((1 << (pos + 1)) - 1)) << (-lo)) if pos >= 0 else 0
            return (val >> lo) & ((1 << n) - 1)

        def step(pos, sL, sM, sO):
            bl, nl, yl = cL[sL]
            bm, nm, ym = cM[sM]
            bo, no, yo = cO[sO]
            pos -= yo + ML_BITS[min(ym, 52)] + LL_BITS[min(yl, 35)]
            aL = read(pos, nl); pos -= nl
            aM = read(pos, nm); pos -= nm
            aO = read(pos, no); pos -= no
            return pos, bl + aL, bm + aM, bo + aO

        pos = nbit - 1
        while read(pos, 1) == 0:
            pos -= 1
        pos -= 1
        sL = read(pos, alL); pos -= alL
        sO = read(pos, alO); pos -= alO
        sM = read(pos, alM); pos -= alM
        true_at = {}
        traj = []
        for i in range(bd.n_seq - 1):
            true_at[pos] = (sL, sM, sO, i)
            traj.append(pos)
            pos, sL, sM, sO = step(pos, sL, sM, sO)
        # the probe's own decoder is right if the last sequence's extra bits end exactly at the stream's first bit
        yl, ym, yo = cL[sL][2], cM[sM][2], cO[sO][2]
        assert pos - (yo + ML_BITS[min(ym, 52)] + LL_BITS[min(yl, 35)]) == -1, "probe decoder out of step with the format"
        for _ in range(trials):
            # a segment boundary: an arbitrary BYTE position of the stream, arbitrary states
            start = rng.randrange(nbit // 8 // 8, nbit // 8 * 7 // 8) * 8 + 7
            q, a, m_, o = start, rng.randrange(1 << alL), rng.randrange(1 << alM), rng.randrange(1 << alO)
            synced = None
            for s in range(max_steps):
                t = true_at.get(q)
                if t is not None and t[:3] == (a, m_, o):
                    synced = s
                    break
                if q < 64:
                    break
                q, a, m_, o = step(q, a, m_, o)
            res.append(synced)
    p.close()
    return res


if __name__ == "__main__":
    rng = random.Random(7)
    blob, off, ln, _, _ = sb.make_batch(4, 0, 3, threads=2)
    frames = [("synthetic config 4 #%d" % i, bytes(blob[o:o + l])) for i, (o, l) in enumerate(zip(off, ln))]
    import json
    g = os.path.join(ROOT, "tests", "golden", "decodecorpus")
    for n in sorted(json.load(open(os.path.join(g, "manifest.json")))):
        f = open(os.path.join(g, n + ".zst"), "rb").read()
        if len(f) > 60000:
            frames.append(("corpus " + n, f))
    for name, f in frames[:8]:
        r = probe(f, 60, rng)
        if not r:
            continue
        ok = sorted(x for x in r if x is not None)
        print(f"{name}: {len(ok)}/{len(r)} synchronised; steps to sync: median {ok[len(ok) // 2] if ok else None}, "
              f"p90 {ok[int(len(ok) * 0.9)] if ok else None}, max {ok[-1] if ok else None}")
