"""TEST-ONLY pure-Python interpreter of the mzd.h batch descriptors.

It executes a planned batch exactly as the device contract says (Huffman cells, FSE cells,
sequence records, offset history, raw/RLE blocks) so that the host planner can be checked on CPU
against the golden vectors without a GPU.  It follows the same reference lines as the kernels:
huffman.go:221-264, sequences.go:64-206, sequence_execution.go:14-114."""

LL_BASE = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64,
           0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000]
LL_EXTRA = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
ML_BASE = list(range(3, 35)) + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099,
                                8195, 16387, 32771, 65539]
ML_EXTRA = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]


class RevBits:
    def __init__(self, data: bytes):
        self.v = int.from_bytes(data, "little")
        self.cursor = len(data) * 8 - 1

    def read(self, n):
        if n == 0:
            return 0
        lo = self.cursor - n + 1
        if lo >= 0:
            r = (self.v >> lo) & ((1 << n) - 1)
        elif self.cursor >= 0:
            r = (self.v & ((1 << (self.cursor + 1)) - 1)) << (-lo)
        else:
            r = 0
        self.cursor -= n
        return r

    def skip_padding(self):
        for _ in range(8):
            if self.read(1):
                return True
        return False


def huf_decode(cells, max_bits, data, want):
    rb = RevBits(data)
    if not rb.skip_padding():
        raise ValueError("bad padding")
    state = rb.read(max_bits)
    out = bytearray()
    mask = (1 << max_bits) - 1
    while rb.cursor + 1 > -max_bits:
        sym, nb = cells[state]
        out.append(sym)
        state = ((state << nb) + rb.read(nb)) & mask
    if rb.cursor + 1 != -max_bits or len(out) != want:
        raise ValueError("huffman stream length/bits mismatch")
    return bytes(out)


def run_batch(batch, blob: bytes, prefixes=None, hists_out=None):
    """-> list of output bytes per frame.  A frame description that is a CHUNK of a frame (MZD_FRAME_CONTINUES, ABI 9) starts behind
    `prefixes[f]` -- the `start` bytes its slab begins with -- from the history it names; what is returned is the chunk's own bytes.
    hists_out: a list that receives the offset history behind every frame's last block."""
    outs = []
    for f in range(batch.n_frames):
        fd = batch.frames[f]
        out = bytearray()
        hist = [1, 4, 8]
        start = 0
        if fd.flags & 2:
            out += prefixes[f]
            assert len(out) == fd.start, (len(out), fd.start)
            start = fd.start
            hist = list(fd.hist)
        for bi in range(fd.first_block, fd.first_block + fd.n_blocks):
            b = batch.blocks[bi]
            if b.type == 0:
                out += blob[b.src_off:b.src_off + b.size]
                continue
            if b.type == 1:
                out += blob[b.src_off:b.src_off + 1] * b.size
                continue
            # literals
            if b.lit_type == 0:
                lits = blob[b.lit_off:b.lit_off + b.lit_regen]
            elif b.lit_type == 1:
                lits = blob[b.lit_off:b.lit_off + 1] * b.lit_regen
            else:
                ht = batch.huf_tables[b.huf_table]
                cells = [(batch.huf_entries[ht.entries_off + i].symbol, batch.huf_entries[ht.entries_off + i].nbits)
                         for i in range(1 << ht.max_bits)]
                if b.lit_streams == 4:
                    normal = (b.lit_regen + 3) // 4
                    wants = [normal, normal, normal, b.lit_regen - 3 * normal]
                    off = b.lit_off
                    lits = b""
                    for s in range(4):
                        lits += huf_decode(cells, ht.max_bits, blob[off:off + b.lit_stream_size[s]], wants[s])
                        off += b.lit_stream_size[s]
                else:
                    lits = huf_decode(cells, ht.max_bits, blob[b.lit_off:b.lit_off + b.lit_stream_size[0]], b.lit_regen)
            # sequences
            lit_pos = 0
            if b.n_seq == 0 and b.seq_status:
                raise ValueError(f"sequence section of no sequences: status {b.seq_status}")
            if b.n_seq:
                tabs = []
                for ti in (b.ll_table, b.of_table, b.ml_table):
                    td = batch.fse_tables[ti]
                    tabs.append((td.acc_log, [batch.fse_entries[td.entries_off + i] for i in range(1 << td.acc_log)]))
                rb = RevBits(blob[b.seq_off:b.seq_off + b.seq_size])
                if not rb.skip_padding():
                    raise ValueError("bad padding")
                (all_, tl), (alo, to), (alm, tm) = tabs
                sl = rb.read(all_)
                so = rb.read(alo)
                sm = rb.read(alm)
                for i in range(b.n_seq):
                    el, eo, em = tl[sl], to[so], tm[sm]
                    ofv = (1 << eo.symbol) + rb.read(eo.symbol)
                    ml = ML_BASE[em.symbol] + rb.read(ML_EXTRA[em.symbol])
                    ll = LL_BASE[el.symbol] + rb.read(LL_EXTRA[el.symbol])
                    if i < b.n_seq - 1:
                        sl = el.baseline + rb.read(el.nbits)
                        sm = em.baseline + rb.read(em.nbits)
                        so = eo.baseline + rb.read(eo.nbits)
                    # offset history
                    if ofv > 3:
                        off = ofv - 3
                        hist = [off, hist[0], hist[1]]
                    else:
                        idx = ofv - 1 + (1 if ll == 0 else 0)
                        if idx == 0:
                            off = hist[0]
                        elif idx == 1:
                            off = hist[1]
                            hist = [off, hist[0], hist[2]]
                        elif idx == 2:
                            off = hist[2]
                            hist = [off, hist[0], hist[1]]
                        else:
                            off = hist[0] - 1
                            hist = [off, hist[0], hist[1]]
                    out += lits[lit_pos:lit_pos + ll]
                    lit_pos += ll
                    if off <= 0 or off > len(out):
                        raise ValueError("bad offset")
                    if off >= ml:
                        out += out[len(out) - off:len(out) - off + ml]
                    else:
                        for _ in range(ml):
                            out.append(out[-off])
                if rb.cursor != -1:
                    raise ValueError("sequence bits left over")
            out += lits[lit_pos:]
        outs.append(bytes(out[start:]))
        if hists_out is not None:
            hists_out.append(hist)
    return outs
