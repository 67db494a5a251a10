"""Test-only numpy/Python restatement of fse/fse.go:136-230 BuildDecodingTable, used to check the
count-form tables (MZD_FSE_FROM_COUNTS) of the planner and the device-side build against the host
planner's cells.  Not imported by the product."""
import numpy as np

from sparkzstd_amd import _lib


def counts_of(batch, ti):
    """Normalised counts packed into the first cells of a from-counts table."""
    d = batch.fse_tables[ti]
    assert d.build & _lib.MZD_FSE_FROM_COUNTS
    nsym = d.build & 0xFF
    out = []
    for s in range(nsym):
        e = batch.fse_entries[d.entries_off + (s >> 1)]
        raw = (e.nbits | (e.symbol << 8)) if (s & 1) else e.baseline
        out.append(raw - 65536 if raw >= 32768 else raw)
    return out


def build_cells(counts, acc_log):
    """-> uint32 cells baseline | nbits << 16 | symbol << 24 in table order."""
    size = 1 << acc_log
    sym = [0] * size
    nxt = []
    high = size - 1
    for s, c in enumerate(counts):
        if c == -1:
            sym[high] = s
            high -= 1
            nxt.append(1)
        else:
            nxt.append(c)
    step, mask, pos = (size >> 1) + (size >> 3) + 3, size - 1, 0
    for s, c in enumerate(counts):
        for _ in range(max(c, 0)):
            sym[pos] = s
            pos = (pos + step) & mask
            while pos > high:
                pos = (pos + step) & mask
    assert pos == 0
    out = np.zeros(size, dtype=np.uint32)
    for i in range(size):
        n = nxt[sym[i]]
        nxt[sym[i]] += 1
        nb = acc_log - (n.bit_length() - 1)
        out[i] = (((n << nb) - size) & 0xFFFF) | (nb << 16) | (sym[i] << 24)
    return out


def host_cells(batch, ti):
    d = batch.fse_tables[ti]
    n = 1 << d.acc_log
    return np.array([batch.fse_entries[d.entries_off + i].baseline | (batch.fse_entries[d.entries_off + i].nbits << 16) |
                     (batch.fse_entries[d.entries_off + i].symbol << 24) for i in range(n)], dtype=np.uint32)
