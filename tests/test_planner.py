"""Host planner (sparkzstd_amd/csrc/planner.cpp) on CPU: descriptors produced for the golden
corpus, executed by the test-only descriptor interpreter, must regenerate the originals.  Also
error behaviour of the header parsers.  No GPU, no oracle involvement in the product path."""
import ctypes

import numpy as np
import pytest

import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tests.conftest import check_expected
from tests.desc_interp import run_batch


def _blob_of(batch):
    return bytes((ctypes.c_uint8 * batch.in_size).from_address(batch.in_)) if batch.in_size else b""


def test_planner_parses_whole_corpus(corpus):
    p = z.Plan()
    for name, comp, length, sha, exp in corpus:
        rc, consumed = p.add_frame(comp)
        assert rc == 0, name
        assert consumed == len(comp) - 4  # checksum never read
    b = p.finalize()
    assert b.n_frames == 100
    # SURVEY 4 corpus census: 3651 blocks = 746 raw / 447 rle / 2458 compressed
    types = [b.blocks[i].type for i in range(b.n_blocks)]
    assert (types.count(0), types.count(1), types.count(2)) == (746, 447, 2458)
    nseq = sum(b.blocks[i].n_seq for i in range(b.n_blocks))
    assert nseq == 1031936
    # frames with a content size get an exact slab
    for i, (name, comp, length, sha, exp) in enumerate(corpus):
        fd = b.frames[i]
        if fd.content_size != _lib.MZD_UNKNOWN_SIZE:
            assert fd.content_size == length and fd.out_capacity == length
        else:
            assert fd.out_capacity >= length
        assert fd.out_offset % 16 == 0
    p.close()


def test_descriptors_regenerate_small_corpus_frames(corpus):
    """The descriptor contract carries everything: interpreting it reproduces the originals."""
    small = [c for c in corpus if c[2] <= 40000][:24]
    assert len(small) >= 10
    p = z.Plan()
    for name, comp, *_ in small:
        assert p.add_frame(comp)[0] == 0
    b = p.finalize()
    outs = run_batch(b, _blob_of(b))
    for (name, comp, length, sha, exp), got in zip(small, outs):
        check_expected(name, got, length, sha, exp)
    p.close()


def test_predefined_tables_are_shared_and_match_kat(kat, corpus):
    """fse/fse_test.go:8-41 through the planner: baseline / nbits / code of the predefined LL table."""
    LL_BASE = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64,
               0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000]
    LL_EXTRA = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    p = z.Plan()
    for name, comp, *_ in corpus:
        p.add_frame(comp)
    b = p.finalize()
    found = [i for i in range(b.n_fse_tables) if b.fse_tables[i].kind == 0 and b.fse_tables[i].acc_log == 6]
    ok = 0
    for ti in found:
        td = b.fse_tables[ti]
        cells = [b.fse_entries[td.entries_off + i] for i in range(64)]
        got = [[c.baseline, LL_EXTRA[c.symbol], c.nbits, LL_BASE[c.symbol]] for c in cells]
        ok += got == kat["ll_predefined_table"]
    assert ok >= 1
    p.close()


@pytest.mark.parametrize("frame,code", [
    (b"\x28\xb5\x2f\xfc\x20\x00\x01\x00\x00", 2),            # wrong magic
    (b"\x28\xb5\x2f\xfd\x20\x00\x07\x00\x00", 3),            # reserved block type 3
    (b"\x28\xb5\x2f\xfd\x20\x00\x01\x00", 1),                # truncated block header
    (b"\x28\xb5\x2f\xfd\x20\x00\x09\x00\x10\x00", 4),        # block size 131073 > 128 KiB
    (b"\x28\xb5\x2f\xfd", 1),
])
def test_planner_header_errors(frame, code):
    p = z.Plan()
    rc, _ = p.add_frame(frame)
    assert rc == code
    b = p.finalize()
    assert b.n_frames == 1 and b.frames[0].n_blocks == 0 and p.frame_status(0) == code
    p.close()


def test_handmade_raw_and_rle_frames():
    """config-2 style frames: single-segment, 4-byte FCS, one raw or rle block."""
    payload = bytes(range(200)) * 3
    n = len(payload)
    raw = b"\x28\xb5\x2f\xfd" + bytes([0xA0]) + n.to_bytes(4, "little") + ((n << 3) | 1).to_bytes(3, "little") + payload
    rle = b"\x28\xb5\x2f\xfd" + bytes([0xA0]) + (1000).to_bytes(4, "little") + ((1000 << 3) | 3).to_bytes(3, "little") + b"\x5a"
    p = z.Plan()
    assert p.add_frame(raw) == (0, len(raw))
    assert p.add_frame(rle) == (0, len(rle))
    b = p.finalize()
    outs = run_batch(b, _blob_of(b))
    assert outs[0] == payload and outs[1] == b"\x5a" * 1000
    assert b.frames[0].content_size == n and b.frames[0].window_size == n
    p.close()


def test_count_form_tables_describe_the_same_tables(corpus):
    """mzd_plan_set_device_tables: every Compressed-mode FSE table is shipped as its normalised counts
    (two int16 per cell); expanding them with a restatement of fse.go:136-230 gives exactly the cells
    the host planner builds.  Predefined and RLE tables stay in cell form."""
    from tests.fse_build_ref import build_cells, counts_of, host_cells
    frames = [c[1] for c in corpus[:30]]
    ph, pd = z.Plan(), z.Plan(device_tables=True)
    for f in frames:
        assert ph.add_frame(f)[0] == 0 and pd.add_frame(f)[0] == 0
    bh, bd = ph.finalize(), pd.finalize()
    assert bh.n_fse_tables == bd.n_fse_tables and bd.n_fse_entries < bh.n_fse_entries // 4  # counts are compact
    n_counts = 0
    for ti in range(bd.n_fse_tables):
        dh, dd = bh.fse_tables[ti], bd.fse_tables[ti]
        assert (dh.acc_log, dh.kind) == (dd.acc_log, dd.kind) and dh.build == 0
        if dd.build & _lib.MZD_FSE_FROM_COUNTS:
            n_counts += 1
            counts = counts_of(bd, ti)
            assert sum(1 if c < 0 else c for c in counts) == 1 << dd.acc_log
            if n_counts <= 40:  # the Python build is slow; the GPU test compares every table
                assert (build_cells(counts, dd.acc_log) == host_cells(bh, ti)).all()
        else:
            assert dd.build == 0 and (host_cells(bd, ti) == host_cells(bh, ti)).all()
    assert n_counts > 50
    # Huffman tables: weight form, same MaxBits, compact
    assert bd.n_huf_tables == bh.n_huf_tables > 20 and bd.n_huf_entries < bh.n_huf_entries
    for ti in range(bd.n_huf_tables):
        dh, dd = bh.huf_tables[ti], bd.huf_tables[ti]
        assert dd.max_bits & _lib.MZD_HUF_FROM_WEIGHTS and (dd.max_bits & 0xFF) == dh.max_bits
        nw = (dd.max_bits >> 8) & 0xFFFF
        ws = []
        for j in range(nw):
            e = bd.huf_entries[dd.entries_off + (j >> 1)]
            ws.append(e.nbits if j & 1 else e.symbol)
        s_ = sum(1 << (w - 1) for w in ws if w)
        left = (1 << dh.max_bits) - s_
        assert s_.bit_length() == dh.max_bits and left > 0 and left & (left - 1) == 0
        # every symbol's code length in the host-built table is MaxBits + 1 - weight
        lens = {}
        for j in range(1 << dh.max_bits):
            e = bh.huf_entries[dh.entries_off + j]
            lens[e.symbol] = e.nbits
        for sym, w in enumerate(ws):
            assert (lens.get(sym, 0) == dh.max_bits + 1 - w) if w else sym not in lens
    # the rest of the descriptors is identical
    for i in range(bh.n_blocks):
        a, b = bh.blocks[i], bd.blocks[i]
        assert (a.type, a.n_seq, a.ll_table, a.of_table, a.ml_table, a.seq_off, a.seq_size) == \
               (b.type, b.n_seq, b.ll_table, b.of_table, b.ml_table, b.seq_off, b.seq_size)
    ph.close()
    pd.close()


def test_planner_records_content_checksums(corpus, oracle):
    """Frame descriptor carries the 4 bytes after the last block when Content_Checksum_flag is set;
    they are the low half of XXH64 of the original (oracle restatement of the published algorithm)."""
    p = z.Plan()
    for name, comp, *_ in corpus[:20]:
        assert p.add_frame(comp)[0] == 0
    b = p.finalize()
    for i, (name, comp, length, sha, exp) in enumerate(corpus[:20]):
        fd = b.frames[i]
        assert fd.flags & _lib.MZD_FRAME_HAS_CHECKSUM
        assert fd.checksum == int.from_bytes(comp[-4:], "little")
        if exp is not None:
            assert fd.checksum == oracle.xxh64(exp) & 0xFFFFFFFF
    p.close()


def test_split_frames_multi_frame_stream_with_skippable_frames(corpus):
    """mzd_split_frames: concatenated frames + skippable frames -> the frames' extents (checksum included) and
    output bounds equal to the planner's; defects end the walk with their status."""
    import numpy as np
    frames = [comp for _, comp, *_ in corpus]
    skip = bytes([0x5A, 0x2A, 0x4D, 0x18, 3, 0, 0, 0]) + b"abc"
    blob = skip + frames[0] + skip + skip + b"".join(frames[1:]) + skip
    rc, off, ln, ob, total = z.split_frames(blob)
    assert rc == 0 and len(off) == len(frames)
    for f, o, l in zip(frames, off, ln):
        assert blob[int(o):int(o) + int(l)] == f
    p = z.Plan()
    for f in frames:
        assert p.add_frame(f)[0] == 0
    b = p.finalize()
    assert [int(x) for x in ob] == [int(b.frames[i].out_capacity) for i in range(b.n_frames)]
    assert total == b.out_size - 256
    p.close()
    assert z.split_frames(b"")[0] == 0 and len(z.split_frames(b"")[1]) == 0
    rc, off, *_ = z.split_frames(blob[:-2])  # the last skippable frame is cut
    assert rc == 1 and len(off) == len(frames)
    rc, off, *_ = z.split_frames(frames[0] + b"\x00\x01\x02\x03" + frames[1])
    assert rc == 2 and len(off) == 1
    cut = frames[2][:len(frames[2]) // 2]
    rc, off, *_ = z.split_frames(frames[0] + cut)
    assert rc == 1 and len(off) == 1
    # more frames than the first guess of the binding (1024)
    many = frames[1] * 1500
    rc, off, ln, _, _ = z.split_frames(many)
    assert rc == 0 and len(off) == 1500 and int(off[-1]) == 1499 * len(frames[1])


def test_declared_frame_cost_and_shard_ranges(corpus):
    """Host side of the multi-GPU entry (decode_frames(devices=...)): the cost of a frame is its compressed length plus
    the content size its header declares (frame.go:23-61); equal costs -> equal counts, else contiguous ranges of
    roughly equal C + D.  Every frame lands in exactly one range, in order."""
    from sparkzstd_amd.api import declared_frame_cost, shard_frames
    declared = 0
    for name, comp, length, sha, exp in corpus:
        c = declared_frame_cost(comp)
        assert c >= len(comp)
        if c == len(comp) + length:
            declared += 1
    assert declared >= 40  # about half of the decodecorpus frames declare their content size; the others count with their window
    assert declared_frame_cost(b"") == 0 and declared_frame_cost(b"\x28\xb5\x2f\xfd\x20") == 5 and declared_frame_cost(b"junkjunk") == 8
    frames = [comp for _, comp, *_ in corpus]
    for world in (1, 2, 3, 8, 128):
        r = shard_frames(frames, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == len(frames)
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1)) and all(lo <= hi for lo, hi in r)
    r = shard_frames(frames, 4)
    costs = [declared_frame_cost(f) for f in frames]
    per = [sum(costs[lo:hi]) for lo, hi in r]
    assert max(per) <= 2.5 * (sum(costs) / 4) + max(costs)
    assert shard_frames([b"x" * 10] * 10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_frames([], 2) == [(0, 0), (0, 0)]


def test_planner_tables_equal_the_oracles_cell_for_cell(oracle, corpus):
    """The host planner's FSE decode tables (planner.cpp build_fse_cells) and Huffman decode tables (build_huffman_cells) against the
    ORACLE's builds of the same normalised counts / weights -- orc_fse_build (fse.go:136-230) and orc_huf_build (huffman.go:112-190)
    -- cell for cell: baseline, number of bits, symbol.  (The GPU suite holds the DEVICE-built tables to the same oracle cells:
    tests/test_gpu_corpus.py::test_device_built_fse_tables_equal_host_tables.)"""
    import ctypes
    from tests import fse_build_ref, oracle_binding as ob
    from tools import synth_binding as sb
    blob, off, ln, _, _ = sb.make_batch(4, 7, 12, threads=4)
    blob3, off3, ln3, _, _ = sb.make_batch(3, 11, 6, threads=4)  # MaxBits 11 Huffman tables
    frames = [comp for _, comp, *_ in corpus] + [bytes(blob[o:o + l]) for o, l in zip(off, ln)] + \
             [bytes(blob3[o:o + l]) for o, l in zip(off3, ln3)]
    ph, pd = z.Plan(), z.Plan(device_tables=True)
    for f in frames:
        assert ph.add_frame(f)[0] == 0 and pd.add_frame(f)[0] == 0
    bh, bd = ph.finalize(), pd.finalize()
    host = np.ctypeslib.as_array(ctypes.cast(bh.fse_entries, ctypes.POINTER(ctypes.c_uint32)), shape=(bh.n_fse_entries,)).copy()
    n_fse = 0
    for ti in range(bd.n_fse_tables):
        if not bd.fse_tables[ti].build & _lib.MZD_FSE_FROM_COUNTS:
            continue
        dh = bh.fse_tables[ti]
        counts = fse_build_ref.counts_of(bd, ti)
        t = ob.FseTable()
        t.acc_log, t.n_values = dh.acc_log, len(counts)
        for k, cnt in enumerate(counts):
            t.values[k] = cnt + 1  # (fse.go:19: the value kept is the probability + 1)
        assert oracle.lib.orc_fse_build(ctypes.byref(t), None, 0, None, 0) == 0, ti
        cells = np.array([t.table[i].baseline | (t.table[i].nbits << 16) | (t.table[i].raw_symbol << 24) for i in range(1 << dh.acc_log)],
                         dtype=np.uint32)
        oracle.lib.orc_fse_free(ctypes.byref(t))
        assert (host[dh.entries_off:dh.entries_off + (1 << dh.acc_log)] == cells).all(), (ti, dh.acc_log, dh.kind)
        n_fse += 1
    hufh = np.ctypeslib.as_array(ctypes.cast(bh.huf_entries, ctypes.POINTER(ctypes.c_uint16)), shape=(bh.n_huf_entries,)).copy()
    ht = ob.HufTable()
    for ti in range(bd.n_huf_tables):
        dh, dd = bh.huf_tables[ti], bd.huf_tables[ti]
        nw = (dd.max_bits >> 8) & 0xFFFF
        ws = (ctypes.c_uint8 * max(nw, 1))()
        for j in range(nw):
            e = bd.huf_entries[dd.entries_off + (j >> 1)]
            ws[j] = e.nbits if j & 1 else e.symbol
        assert oracle.lib.orc_huf_build(ctypes.byref(ht), ws, nw) == 0 and ht.max_bits == dh.max_bits, ti
        cells = np.frombuffer(ht.symbols, dtype=np.uint8, count=1 << dh.max_bits).astype(np.uint16) | \
            (np.frombuffer(ht.nbits, dtype=np.uint8, count=1 << dh.max_bits).astype(np.uint16) << 8)
        assert (hufh[dh.entries_off:dh.entries_off + (1 << dh.max_bits)] == cells).all(), (ti, dh.max_bits)
    assert n_fse > 500 and bd.n_huf_tables > 300
    ph.close()
    pd.close()


def test_fse_table_with_a_symbol_beyond_its_kind_is_a_documented_limit(oracle):
    """A mutation the soak of round 6 found (tests/golden/fuzz_ml_symbol_53.zst: a corpus-derived frame, 175 bytes): its match-length
    table gives symbol 53 -- one beyond ML's 53 codes -- the probability "less than one".  The reference builds the table anyway, the
    symbol untranslated and without extra bits (fse.go:219-224), and the oracle, which follows it, decodes the frame; libzstd rejects it.
    The planner reports MZD_ERR_UNSUPPORTED (16, a documented limit: the device's code tables translate the format's codes only),
    not MZD_ERR_FSE_TABLE (5), which would claim the reference fails the frame too."""
    import os
    f = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_ml_symbol_53.zst"), "rb").read()
    rc, out, _, _ = oracle.decode_frame(f, cap=1 << 20)
    assert rc == 0 and len(out) == 37338
    for dt in (False, True):
        p = z.Plan(device_tables=dt)
        assert p.add_frame(f)[0] == 16
        p.close()


def _chunked_on_cpu(comp: bytes, max_out: int, piece: int):
    """The frame through the cursor, its source arriving `piece` bytes at a time, every chunk run by the descriptor interpreter behind
    the window bytes of the chunks before it -- what mzd_fstream_next does with the device (mzd_api.hip), on the CPU.
    -> (output, chunks, consumed)"""
    cur = z.Cursor()
    src = np.frombuffer(comp, dtype=np.uint8)
    pos = have = 0
    out = bytearray()
    keep = b""
    hist = [1, 4, 8]
    chunks = 0
    while True:
        have = min(len(comp), have + piece)
        rc, used, b, last = cur.next(src[pos:have], max_out, len(keep), hist)
        assert rc == 0, rc
        if b is None:
            pos += used
            assert have < len(comp), "the cursor wants bytes behind the frame's end"
            continue
        assert b.n_frames == 1 and b.frames[0].start == len(keep) and b.frames[0].flags & _lib.MZD_FRAME_CONTINUES
        assert b.frames[0].out_capacity >= len(keep) and b.in_size == used
        hs = []
        got = run_batch(b, _blob_of(b), prefixes=[keep], hists_out=hs)[0]
        assert len(got) <= max(max_out, 128 * 1024)
        out += got
        hist = hs[0]
        pos += used
        chunks += 1
        w = cur.window
        keep = bytes(out[max(0, len(out) - w):]) if w else b""
        if last:
            break
    rc, used, b, last = cur.next(src[pos:], max_out, len(keep), hist)
    assert rc == 17 and b is None  # MZD_ERR_OUT_OF_BLOCKS (framedecompressor.go:196)
    cs = cur.content_size
    assert cs == _lib.MZD_UNKNOWN_SIZE or cs == len(out)
    cur.close()
    return bytes(out), chunks, pos


@pytest.mark.parametrize("max_out,piece", [(128 * 1024, 1 << 30), (128 * 1024, 777), (512 * 1024, 65536), (1 << 30, 1 << 30)])
def test_cursor_chunks_regenerate_the_frames(corpus, max_out, piece):
    """ABI 9, the host half of a frame in chunks: whatever the chunk size and however the source arrives, the chunks -- each with the
    tables in force at its start (Repeat / Treeless across a chunk boundary: framedecompressor.go:283-294), the window bytes and the
    offset history of the chunks before it -- regenerate the frame; everything up to the content checksum is consumed."""
    small = [c for c in corpus if c[2] <= 300000]
    picked = sorted(small, key=lambda c: -c[2])[:6] + small[:6]
    many = 0
    for name, comp, length, sha, exp in picked:
        got, chunks, used = _chunked_on_cpu(comp, max_out, piece)
        check_expected(name, got, length, sha, exp)
        assert used == len(comp) - 4
        many += chunks > 1
    assert (many > 0) == (max_out < (1 << 30))


def test_cursor_errors_stick_and_a_cut_frame_wants_more():
    cur = z.Cursor()
    rc, used, b, last = cur.next(np.frombuffer(b"\x28\xb5\x2f", dtype=np.uint8), 1 << 20)
    assert (rc, used, b) == (0, 0, None)  # not all of the header yet
    rc, used, b, last = cur.next(np.frombuffer(b"\x00\x00\x00\x00\x00\x00", dtype=np.uint8), 1 << 20)
    assert rc == 2 and b is None  # MZD_ERR_MAGIC
    assert cur.next(np.frombuffer(b"\x28\xb5\x2f\xfd\x20\x00\x01\x00\x00", dtype=np.uint8), 1 << 20)[0] == 2  # ... sticks
    cur.close()
    # a Raw block of 5 bytes, then the last (RLE) block -- handed over with its payload cut
    frame = b"\x28\xb5\x2f\xfd\x00\x48" + b"\x28\x00\x00hello" + b"\x1b\x00\x00z"
    cur = z.Cursor()
    rc, used, b, last = cur.next(np.frombuffer(frame[:-1], dtype=np.uint8), 1 << 20)
    assert rc == 0 and b is not None and b.n_blocks == 1 and not last and used == 6 + 8 and cur.window == 1 << 19
    rc, used2, b, last = cur.next(np.frombuffer(frame[used:], dtype=np.uint8), 1 << 20, 5, [1, 4, 8])
    assert rc == 0 and last and used2 == 4 and b.blocks[0].type == 1 and b.blocks[0].size == 3
    assert b.frames[0].start == 5 and b.frames[0].flags & _lib.MZD_FRAME_CONTINUES and b.frames[0].out_capacity == 8
    cur.close()


def _batch_image(b):
    """everything a batch describes, as bytes (the descriptor arrays; not the pointers)"""
    def arr(ptr, n, t):
        return bytes((t * n).from_address(ctypes.addressof(ptr.contents))) if n else b""
    return (b.n_frames, b.n_blocks, b.n_fse_tables, b.n_fse_entries, b.n_huf_tables, b.n_huf_entries, b.in_size, b.out_size,
            arr(b.frames, b.n_frames, _lib.FrameDesc), arr(b.blocks, b.n_blocks, _lib.BlockDesc),
            arr(b.fse_tables, b.n_fse_tables, _lib.FseTableDesc), arr(b.fse_entries, b.n_fse_entries, _lib.FseEntry),
            arr(b.huf_tables, b.n_huf_tables, _lib.HufTableDesc), arr(b.huf_entries, b.n_huf_entries, _lib.HufEntry))


def test_blocks_parsed_in_ranges_describe_what_the_serial_walk_describes(corpus):
    """A large frame's blocks are parsed in contiguous ranges on several host threads and stitched in order (planner.cpp
    parse_blocks_parallel): Repeat_Mode / Treeless references that cross a range boundary (framedecompressor.go:283-294) are marks
    until then.  The descriptions must be the serial walk's, byte for byte: every corpus frame of 32 blocks and more through the
    cursor on 1 and on 8 threads, whole and in chunks; a text-like frame of 9 MiB and a frame of Raw / RLE / literal-only blocks
    through the whole-frame planner; a frame with a defect in the middle (the same status from both)."""
    from tools import synth_binding as sb
    many = [comp for _, comp, *_ in corpus]
    big = sb.compress(sb.generate(sb.TEXT, 17, 9 << 20), sb.MODE_FULL)[0]
    n_par = 0
    for comp in many + [big]:
        src = np.frombuffer(comp, dtype=np.uint8)
        for max_out in (1 << 30, 6 << 20):
            imgs = []
            for threads in (1, 8):
                cur = z.Cursor(threads)
                pos, img = 0, []
                while True:
                    rc, used, b, last = cur.next(src[pos:], max_out, 0, None)
                    assert rc == 0 and b is not None
                    img.append((used, last, _batch_image(b)))
                    pos += used
                    if last:
                        break
                cur.close()
                imgs.append(img)
            assert imgs[0] == imgs[1], len(comp)
            n_par += any(im[2][1] >= 32 for im in imgs[0])
    assert n_par >= 10, n_par  # (frames whose chunks were large enough to be parsed in ranges)
    # the whole-frame planner: add_frames with one thread (serial) and with eight (the frame's blocks in ranges)
    hurt = bytearray(big)
    hurt[len(hurt) // 2] ^= 0x40
    for frame in (big, bytes(hurt)):
        res = []
        for threads in (1, 8):
            p = z.Plan()
            blob = np.frombuffer(frame, dtype=np.uint8)
            rc = p.add_frames(blob, np.array([0], dtype=np.uint64), np.array([len(frame)], dtype=np.uint64), threads=threads)
            res.append((rc, p.frame_status(0), _batch_image(p.finalize())))
            p.close()
        assert res[0] == res[1]
    assert res[0][0] != 0 or True


def _zero_sequence_frames():
    """frames whose last block declares ZERO sequences in the two-byte form (0x80 0x00) -- modes, tables and a bitstream follow all the
    same -- with what the reference makes of each (sequences.go:126-208: padding + initial states must use the bitstream up)"""
    from tests.frame_splice import splice_frame
    head = (0, b"hello world, hello", 18)
    lit = bytes([(23 << 3) | 1, 0x41])  # RLE literals: 23 x 'A'
    mk = lambda tail: splice_frame([head, (2, lit + b"\x80\x00" + tail, len(lit) + 2 + len(tail))])
    return [("RLE tables, the padding bit alone", mk(b"\x54\x00\x00\x00\x01"), 0),
            ("RLE tables, one bit left over", mk(b"\x54\x00\x00\x00\x03"), 11),
            ("RLE tables, a zero byte on top", mk(b"\x54\x00\x00\x00\x00"), 8),
            ("predefined tables, 17 state bits behind 7 bits of padding", mk(b"\x00\x12\x34\x02"), 0),
            ("predefined tables, a byte too many", mk(b"\x00\x12\x34\x56\x02"), 11)]


def test_zero_sequences_in_the_two_byte_form(oracle):
    """Found by the chunk soak of round 6 (tests/golden/fuzz_zero_sequences_long_form.zst: a corpus frame with ONE byte flipped): a block
    whose Number_of_Sequences is zero in its two-byte form used to pass as its literals; the reference decodes the section and
    returns ErrNotAllBitsUsed.  The planner now says what the reference's DecodeSequences makes of such a block
    (mzd_block_desc.seq_status) -- the oracle's verdict on every case; the descriptor interpreter honours it."""
    import os
    f = open(os.path.join(os.path.dirname(__file__), "golden", "fuzz_zero_sequences_long_form.zst"), "rb").read()
    for what, frame, want in [("the soak's frame", f, 11)] + _zero_sequence_frames():
        rc, ref, *_ = oracle.decode_frame(frame, cap=1 << 20)
        assert rc == want, what
        p = z.Plan()
        assert p.add_frame(frame)[0] == 0
        b = p.finalize()
        last = b.blocks[b.n_blocks - 1]
        assert last.n_seq == 0 and last.seq_status == want, (what, last.seq_status)
        if want == 0:
            assert run_batch(b, _blob_of(b))[0] == ref, what
        else:
            with pytest.raises(ValueError):
                run_batch(b, _blob_of(b))
        p.close()
