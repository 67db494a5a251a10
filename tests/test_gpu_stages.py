"""-m gpu: the STAGE boundaries of the hot path against the oracle, the device bit
reader against the reference's own bit-reader vectors, the BASELINE configs at their stated sizes, and the
self-launching multi-rank bench.  Everything goes through the C-ABI."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import sparkzstd_amd as z
from sparkzstd_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    return z.Context(0)


# ---- B0: the device's backward bit reader, driven like bitstream/reversebitstream_test.go

def test_device_bit_reader_edge_vector(ctx, kat):
    """bitstream/reversebitstream_test.go:172-229: {64,58,169,224}, reads that end on / cross byte borders, a
    zero-bit read, and the last read running past the start of the stream."""
    k = kat["rbs_edges"]
    vals, left = ctx.backbits(bytes(k["data"]), k["reads"])
    assert vals == k["expect"]
    want_left, cur = [], 8 * len(k["data"]) - 1
    for n in k["reads"]:
        cur -= n
        want_left.append(cur)
    assert left == want_left


def test_device_bit_reader_ramp_patterns(ctx, oracle):
    """bitstream/reversebitstream_test.go:7-170: the 256-byte ramp read in the test's bit patterns; every value
    equals the oracle's reader (itself pinned on the same vectors in tests/test_oracle.py) and the bit string."""
    from tests.oracle_binding import Rbs
    data = bytes(range(256))
    bits = []
    for b in reversed(data):
        bits += [(b >> i) & 1 for i in range(7, -1, -1)]
    for pattern in ([8], [4], [5] * 8 + [8], [3] * 8 + [8], [6, 3, 3, 3, 3, 6, 8], [7, 7, 7, 3, 8], [32, 1, 31, 0, 17], [11]):
        reads, pos = [], 0
        while pos + pattern[len(reads) % len(pattern)] <= len(bits):
            reads.append(pattern[len(reads) % len(pattern)])
            pos += reads[-1]
        vals, left = ctx.backbits(data, reads)
        r = Rbs()
        oracle.lib.orc_rbs_init(ctypes.byref(r), data, len(data))
        pos = 0
        for n, v in zip(reads, vals):
            want = 0
            for b in bits[pos:pos + n]:
                want = (want << 1) | b
            assert v == want == oracle.lib.orc_rbs_read(ctypes.byref(r), n), (pattern, pos, n)
            pos += n
        assert left[-1] == 8 * len(data) - 1 - pos


def test_device_bit_reader_reads_zeros_below_the_start(ctx, oracle):
    """reversebitstream.go:23-27,67-75: past the start of the stream Read returns zero bits and the cursor keeps
    decrementing.  The hook surrounds the stream with non-zero bytes, so the masking is the reader's own."""
    from tests.oracle_binding import Rbs
    rng = np.random.default_rng(5)
    for ln in (1, 2, 7, 8, 9, 15, 16, 17, 40):
        data = bytes(rng.integers(1, 256, ln, dtype=np.uint8))
        reads = [int(x) for x in rng.integers(0, 33, 6 + ln)] + [32, 32, 7, 0, 1]
        vals, left = ctx.backbits(data, reads)
        r = Rbs()
        oracle.lib.orc_rbs_init(ctypes.byref(r), data, len(data))
        for i, n in enumerate(reads):
            assert vals[i] == oracle.lib.orc_rbs_read(ctypes.byref(r), n), (ln, i, n)
            assert left[i] == r.offset, (ln, i)
        assert left[-1] < -1  # the sequence really over-read


# ---- stage boundaries: regenerated literals and (LL, ML, resolved offset) per block, against the oracle's trace

def _next_offset(hist, value, ll):
    """sequence_execution.go:65-114 (the test's own restatement, used only to turn a symbolic record into a number)"""
    if value > 3:
        off = value - 3
        hist[:] = [off, hist[0], hist[1]]
        return off
    idx = value - 1 + (1 if ll == 0 else 0)
    if idx == 0:
        return hist[0]
    off = hist[idx] if idx < 3 else hist[0] - 1
    if idx == 1:
        hist[:] = [off, hist[0], hist[2]]
    else:
        hist[:] = [off, hist[0], hist[1]]
    return off


@pytest.mark.parametrize("seq_variant,huf_variant", [(0, 1), (1, 1), (0, 2), (3, 1), (0, 3), (0, 4)])
def test_literals_and_sequences_per_block_equal_the_oracle_trace(corpus, oracle, seq_variant, huf_variant):
    """After one pass over the whole corpus: for every compressed block, the literal bytes the Huffman stage
    regenerated (literals.go:283-361 LiteralSection.Data) and every sequence's (LiteralLength, MatchLength,
    resolved offset) (sequences.go:11-15 + sequence_execution.go:65-114) as the device holds them between its
    stages equal the oracle's trace, block by block -- not only the final bytes."""
    c = z.Context(0, seq_variant=seq_variant, huf_variant=huf_variant)
    n = _check_stage_boundaries(c, corpus, oracle)
    assert n["blocks"] > 1500 and n["seqs"] > 1000000 and n["lits"] > 2000000
    assert n["symbolic"] > 0 or seq_variant == 1  # later blocks of a frame carry repeat offsets relative to the block start


def test_stage_boundaries_of_a_batch_of_small_frames(corpus, oracle):
    """The same check on a batch shaped like BASELINE's configs -- no frame above 128 KiB, so the library's own choices are the
    split pass's (k_huf first, k_seq_q4, k_exec_c): the corpus's frames of up to 128 KiB (real data: long literal runs, long
    matches, multi-block frames whose later blocks start with symbolic repeat offsets) beside text-like synthetic frames."""
    from tools import synth_binding as sb
    c = z.Context(0)
    small = [it for it in corpus if it[2] <= 131072]
    blob, off, ln, _, _ = sb.make_batch(4, 77, 6, 131072, threads=2)
    synth = []
    for i in range(len(off)):
        comp = bytes(blob[int(off[i]):int(off[i]) + int(ln[i])])
        rc, ref, _, _ = oracle.decode_frame(comp, cap=1 << 18)
        assert rc == 0
        synth.append((f"synth{i}", comp, len(ref), hashlib.sha256(ref).hexdigest(), ref))
    # the planner's bound of a frame that declares no content size is its window or 128 KiB per block: keep what qualifies
    plan = z.Plan(device_tables=True)
    for _, comp, *_ in small:
        assert plan.add_frame(comp)[0] == 0
    b = plan.finalize()
    items = [it for i, it in enumerate(small) if int(b.frames[i].out_capacity) <= 131072] + synth
    plan.close()
    assert len(items) > 40
    n = _check_stage_boundaries(c, items, oracle)
    assert n["pass"] & _lib.MZD_PASS_EXEC_C and not n["pass"] & _lib.MZD_PASS_BLOCK_MODE
    assert n["seqs"] > 100000 and n["long"] > 100 and n["symbolic"] > 0, n


def _check_stage_boundaries(c, corpus, oracle):
    """one pass over the frames of `corpus` ((name, compressed, length, ...) items) on context `c` (closed here): literals,
    sequence records and tile bases of every block against the oracle's trace, the frames' bytes against its output.
    -> counts of what was seen"""
    frames = [comp for _, comp, *_ in corpus]
    plan = z.Plan(device_tables=True)
    for f in frames:
        assert plan.add_frame(f)[0] == 0
    b = plan.finalize()
    rb = c.upload(b)
    try:
        rb.run()
        last_pass = rb.last_pass()
        out_blob, status, out_len = rb.download()
        assert (status == 0).all()
        st = rb.stats()
        dblocks = rb.debug_blocks(b.n_blocks)
        n_rec = int(st.n_sequences)
        recs = rb.debug_read(_lib.MZD_DEBUG_RECORDS, np.uint64, 0, n_rec)
        lit_bytes = max((int(d.lit_src) + int(d.lit_regen) for d in dblocks if d.type == 2 and d.lit_type == 2 and not d.lit_in_place),
                        default=0)
        lits = rb.debug_read(_lib.MZD_DEBUG_LITERALS, np.uint8, 0, lit_bytes)
        tiles = rb.debug_read(_lib.MZD_DEBUG_TILES, np.uint32, 0, 2 * sum((int(d.n_seq) + 63) // 64 for d in dblocks))
        blob = bytes(np.ctypeslib.as_array(ctypes.cast(b.in_, ctypes.POINTER(ctypes.c_uint8)), shape=(b.in_size,)))
        n_blocks_seen = n_seq_seen = n_lit_seen = n_symbolic = n_in_place = n_long = 0
        for fi, f in enumerate(frames):
            rc, ref, _, tr = oracle.decode_frame(f, cap=corpus[fi][2] + 64, want_trace=True)
            assert rc == 0
            fd = b.frames[fi]
            assert out_blob[int(fd.out_offset):int(fd.out_offset) + int(out_len[fi])].tobytes() == ref, corpus[fi][0]
            assert fd.n_blocks == len(tr["blocks"])
            lit_at = seq_at = 0
            hist = [1, 4, 8]  # framedecompressor.go:48,59
            for k, ob in enumerate(tr["blocks"]):
                d = dblocks[fd.first_block + k]
                assert d.type == ob["block_type"], (fi, k)
                if ob["block_type"] != 2:
                    continue
                n_blocks_seen += 1
                # literals
                regen = ob["lit_regen"]
                want_l = tr["literals"][lit_at:lit_at + regen]
                lit_at += regen
                assert d.lit_regen == regen
                if d.lit_type == 2 and d.lit_in_place:  # no sequences: the Huffman stage wrote the block's output itself
                    assert ob["n_seq"] == 0
                    got_l = out_blob[int(d.lit_src):int(d.lit_src) + regen].tobytes()
                    n_in_place += 1
                elif d.lit_type == 2:
                    got_l = lits[int(d.lit_src):int(d.lit_src) + regen].tobytes()
                elif d.lit_type == 0:
                    got_l = blob[int(d.lit_src):int(d.lit_src) + regen]
                else:
                    got_l = blob[int(d.lit_src):int(d.lit_src) + 1] * regen
                assert got_l == want_l, (corpus[fi][0], k, "literals")
                n_lit_seen += regen
                # sequences
                ns = ob["n_seq"]
                assert d.n_seq == ns
                h0 = list(hist)  # history at the start of the block: what a symbolic record refers to
                lit_pos = out_pos = 0
                for j in range(ns):
                    ll, ml, raw, resolved = tr["seqs"][seq_at + j]
                    r = int(recs[int(d.rec_off) + j])
                    g_ll, g_ml, g_of = r & 0x1FFFF, (r >> 17) & 0x3FFFF, (r >> 35) & 0x1FFFFFFF
                    if g_of & (1 << 28):
                        u = g_of & 0x0FFFFFFF
                        g_of = h0[u & 3] - (u >> 2)
                        n_symbolic += 1
                    n_long += 1 if (ll > 127 or ml > 255 or g_of >= (1 << 17)) else 0  # long literal runs / matches, far offsets
                    assert (g_ll, g_ml, g_of) == (ll, ml, resolved), (corpus[fi][0], k, j)
                    assert _next_offset(hist, raw, ll) == resolved  # the test's history agrees with the oracle's
                    if j % 64 == 0:
                        t = int(d.tile_off) + j // 64
                        assert (int(tiles[2 * t]), int(tiles[2 * t + 1])) == (lit_pos, out_pos), (corpus[fi][0], k, j)
                    lit_pos += ll
                    out_pos += ll + ml
                seq_at += ns
                n_seq_seen += ns
            assert lit_at == len(tr["literals"]) and seq_at == len(tr["seqs"])
        assert n_seq_seen == n_rec
        return {"blocks": n_blocks_seen, "seqs": n_seq_seen, "lits": n_lit_seen, "symbolic": n_symbolic, "long": n_long, "pass": last_pass}
    finally:
        rb.free()
        plan.close()
        c.close()


def test_release_library_refuses_the_test_kernels(corpus):
    """The release library has one kernel per stage: a context that forces a second implementation (k_seq, k_seq_pipe, k_huf_seg,
    k_exec_b) on it gets MZD_ERR_UNSUPPORTED from the pass -- not a silent substitute --; the same context on libmzd_test.so (what
    z.Context picks for such variants by itself) decodes, and the test library's DEFAULT path gives the release library's bytes."""
    frames = [comp for _, comp, *_ in corpus[:12]]
    for kw in ({"seq_variant": 1}, {"seq_variant": 3}, {"huf_variant": 2}, {"exec_variant": 2}, {"exec_variant": 3}):
        c = z.Context(0, library="release", **kw)
        with pytest.raises(z.MzdError) as e:
            z.decode_frames(frames, c)
        assert e.value.code == 16, kw
        c.close()
        c = z.Context(0, **kw)
        assert c._L is _lib.load_test()
        outs, sts = z.decode_frames(frames, c)
        assert sts == [0] * len(frames)
        c.close()
    ct, cr = z.Context(0, library="test"), z.Context(0)
    assert z.decode_frames(frames, ct) == z.decode_frames(frames, cr)
    ct.close()
    cr.close()


# ---- k_huf_seg: one wavefront per Huffman stream, segments decoded in parallel (self-synchronising codes)

@pytest.mark.parametrize("huf_variant", [2, 3, 4])
def test_huf_seg_corpus_bit_exact_all_maxbits(corpus, huf_variant):
    """The whole corpus (Huffman tables with MaxBits 1..11, 1- and 4-stream sections, Treeless tables, streams from
    a few bytes to 40 KB) through k_huf_seg (2), k_huf_w (4: every stream; round 6) and through k_huf with its transposed bulk phase (3: the wavefronts
    whose tables have MaxBits <= 5 take it, lanes with other streams' chunks to load and symbols to store): golden bytes."""
    from tests.conftest import check_expected
    c = z.Context(0, huf_variant=huf_variant)
    outs, sts = z.decode_frames([comp for _, comp, *_ in corpus], c)
    assert sts == [0] * len(corpus)
    for (name, comp, length, sha, exp), got in zip(corpus, outs):
        check_expected(name, got, length, sha, exp)
    outs_d, sts_d = z.decode_frames([comp for _, comp, *_ in corpus], c, device_plan=True)
    assert sts_d == sts and outs_d == outs
    c.close()


def test_huf_seg_literal_heavy_frames_of_every_size(oracle):
    """Literals-only and mixed frames whose streams range from one symbol to 32 768 (1 to 64 segments per stream,
    segment borders everywhere), skewed (short codes, many symbols per segment) and flat (MaxBits 11) alphabets."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(11)
    frames, want = [], []
    sizes = [1, 2, 5, 63, 64, 65, 200, 511, 512, 513, 1000, 4097, 9000, 20000, 65535, 65536, 100000, 131071, 131072]
    for i, n in enumerate(sizes + [int(x) for x in rng.integers(1, 131073, 40)]):
        kind = [sb.EXP, sb.TEXT, sb.EXP][i % 3]
        data = sb.generate(kind, 500 + i, n)
        mode = sb.MODE_LITERALS if i % 2 == 0 else sb.MODE_FULL
        frames.append(sb.compress(data, mode)[0])
        want.append(data)
    for hv in (2, 1, 3, 4):
        c = z.Context(0, huf_variant=hv)
        outs, sts = z.decode_frames(frames, c)
        assert sts == [0] * len(frames), (hv, sts)
        assert outs == want, hv
        c.close()
    for f, w in list(zip(frames, want))[::9]:
        rc, ref, _, _ = oracle.decode_frame(f, cap=len(w) + 64)
        assert rc == 0 and ref == w


def test_huf_w_segments_that_make_more_symbols_than_a_lane_keeps(oracle):
    """k_huf_w sizes a stream's segments by its AVERAGE code length (a lane keeps 104 symbols in registers); a stream whose code
    lengths change along the way -- incompressible bytes, then one byte over and over: 8 bits a symbol, then 1 -- makes segments with
    several times that: the round is run again with half the segment until every lane's symbols fit.  Literal-only and full frames,
    the long stretch first / last / in the middle, streams of one and of several rounds; against the oracle and the content."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(77)
    datas = []
    for n, order in ((120000, 0), (131072, 1), (60000, 2), (9000, 0), (131072, 3)):
        noise = rng.integers(0, 256, n // 3, dtype=np.uint8).tobytes()
        flat = bytes([65]) * (n - len(noise) - n // 10) + bytes(rng.integers(65, 68, n // 10, dtype=np.uint8))
        parts = [noise, flat] if order == 0 else [flat, noise] if order == 1 else [flat[:len(flat) // 2], noise, flat[len(flat) // 2:]]
        if order == 3:
            parts = [sb.generate(sb.TEXT, 5, n // 2), flat[:n - n // 2]]
        datas.append(b"".join(parts)[:n])
    frames = [sb.compress(d, sb.MODE_LITERALS)[0] for d in datas] + [sb.compress(d, sb.MODE_FULL)[0] for d in datas]
    want = datas + datas
    for hv in (4, 0, 3):
        c = z.Context(0, huf_variant=hv)
        outs, sts = z.decode_frames(frames, c)
        assert sts == [0] * len(frames), (hv, sts)
        assert outs == want, hv
        c.close()
    for f, w in zip(frames, want):
        rc, ref, _, _ = oracle.decode_frame(f, cap=len(w) + 64)
        assert rc == 0 and ref == w


def test_huf_seg_reports_the_lane_kernels_status_on_damaged_streams(corpus, oracle):
    """End conditions of huffman.go:248-261 / literals.go:320-366 under damage: every corpus frame and literal-heavy
    synthetic frames, mutated (seeded byte flips) -- k_huf_seg and k_huf must give the SAME status and bytes frame for
    frame, a frame either decodes under the oracle to the same bytes or is rejected by it too."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(20261002)
    base = [comp for _, comp, *_ in corpus if len(comp) >= 24]
    for i in range(30):
        base.append(sb.compress(sb.generate(sb.EXP, 900 + i, int(rng.integers(2000, 131073))), sb.MODE_LITERALS)[0])
    frames = []
    for comp in base:
        for k in range(4):
            b = bytearray(comp)
            for pos in rng.integers(8, len(b), size=1 + k % 3):
                b[int(pos)] ^= int(rng.integers(1, 256))
            frames.append(bytes(b))
    res = {}
    for hv in (1, 2, 3, 4):
        c = z.Context(0, huf_variant=hv)
        res[hv] = z.decode_frames(frames, c)
        c.close()
    assert res[1][1] == res[4][1], [(i, a, b) for i, (a, b) in enumerate(zip(res[1][1], res[4][1])) if a != b][:20]
    assert res[1][0] == res[4][0]
    assert res[1][1] == res[2][1], [(i, a, b) for i, (a, b) in enumerate(zip(res[1][1], res[2][1])) if a != b][:20]
    assert res[1][0] == res[2][0]
    assert res[1][1] == res[3][1], [(i, a, b) for i, (a, b) in enumerate(zip(res[1][1], res[3][1])) if a != b][:20]
    assert res[1][0] == res[3][0]
    n_huf_err = sum(1 for s in res[2][1] if s in (8, 9, 10))
    assert n_huf_err > 20  # the damage really reached the Huffman end conditions
    for f, o, s in list(zip(frames, res[2][0], res[2][1]))[::5]:
        rc, ref, _, _ = oracle.decode_frame(f, cap=4 << 20)
        # the device fails exactly the frames the reference's algorithm fails; what it may add are the checks the reference does
        # not make (a block of more than 128 KiB: 12; the declared content size: 15) and its documented limits (16)
        assert (s == 0) == (rc == 0) or (rc == 0 and s in (12, 15, 16)), (s, rc)
        if s == 0:
            assert o == ref


# ---- BASELINE.json configs at their stated sizes: size-independent properties on every frame

def _run_full_config(config, n_frames, ctx, frame_bytes=131072, oracle=None):
    import torch
    from tools import synth_binding as sb
    blob, off, ln, cks, nseq = sb.make_batch(config, 0, n_frames, frame_bytes, threads=0)
    plan = z.Plan(device_tables=True)
    assert plan.add_frames(blob, off, ln, threads=0) == 0
    batch = plan.finalize()
    assert batch.n_frames == n_frames
    rb = ctx.upload(batch)
    try:
        rb.run()
        # a resident batch is re-runnable: the second pass must leave the same bytes -- by itself.  The output blob is
        # POISONED between the two (0xA5 everywhere), so a second pass that wrote nothing would be caught.
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        hip.hipDeviceSynchronize.argtypes = []
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemset(rb.device_out_ptr(), 0xA5, n_frames * frame_bytes) == 0
        assert hip.hipDeviceSynchronize() == 0
        rb.run()
        _, status, out_len = rb.download(want_out=False)
        assert (status == 0).all(), np.unique(status, return_counts=True)
        assert (out_len == frame_bytes).all()
        # position-weighted checksum of every frame, computed on the device from the regenerated bytes and
        # compared with the one the generator took from the ORIGINAL content (tools/synth)
        n = n_frames * frame_bytes
        dptr = rb.device_out_ptr()
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert hip.hipMemcpy(out.data_ptr(), dptr, n, 3) == 0  # device to device; frames are laid out back to back
        words = frame_bytes // 8
        wts = 2 * torch.arange(words, dtype=torch.int64, device="cuda") + 1
        exp = torch.from_numpy(cks.view(np.int64)).cuda()
        o64 = out.view(torch.int64).view(n_frames, words)
        for c0 in range(0, n_frames, 4096):
            got = (o64[c0:c0 + 4096] * wts).sum(dim=1)
            assert bool((got == exp[c0:c0 + 4096]).all()), (config, c0)
        if oracle is not None:
            # ... and the generator's figure pinned to the ORACLE: ~2 048 frames spread over the batch go through the C restatement
            # of the reference, whose regenerated bytes must give the same word sums (device == generator == oracle; bench.py's
            # cpu_baseline leg does this for every frame of the headline batch)
            idx = np.arange(0, n_frames, max(1, n_frames // 2048))
            st, ol, ws = oracle.decode_frames_wsum(blob, off[idx], ln[idx], frame_bytes, threads=os.cpu_count() or 1)
            assert (st == 0).all() and (ol == frame_bytes).all() and (ws == cks[idx]).all(), config
        return rb.stats()
    finally:
        rb.free()
        plan.close()


def test_config2_raw_rle_4096_frames_full_size(ctx, oracle):
    """BASELINE configs[1]: 4096 single-block Raw / RLE frames of 128 KiB (framedecompressor.go:211-215,229-241)."""
    st = _run_full_config(2, 4096, ctx, oracle=oracle)
    assert list(st.n_blocks) == [2048, 2048, 0]


def test_config3_huffman_only_4096_frames_full_size(ctx, oracle):
    """BASELINE configs[2]: 4096 frames, 4-stream Huffman literals (MaxBits 11), no sequences."""
    st = _run_full_config(3, 4096, ctx, oracle=oracle)
    assert st.n_huf_streams == 4 * 4096 and st.n_sequences == 0


def test_config4_full_frames_65536_full_size(ctx, oracle):
    """BASELINE configs[3]: 65536 text-like single-block frames, Huffman literals + FSE sequences + match copy."""
    st = _run_full_config(4, 65536, ctx, oracle=oracle)
    assert st.n_huf_streams == 4 * 65536 and st.n_sequences > 65536 * 10000


def test_config4_shard_of_eight_gpus_8192_frames_full_size(ctx, oracle):
    """BASELINE configs[4]: what ONE of eight GPUs runs -- the 8 192-frame shard of the 65 536-frame batch, at full size
    (a single round of the sequence stage: 32 chains per CU, the Huffman kernel beside it)."""
    st = _run_full_config(4, 8192, ctx, oracle=oracle)
    assert st.n_huf_streams == 4 * 8192 and st.n_sequences > 8192 * 10000


def test_decode_frames_over_two_contexts_splits_and_stitches(corpus):
    """The multi-GPU entry of the product (decode_frames(devices=[...]) / DecodeFrames): one host thread and one context per
    listed device, contiguous frame ranges, results in frame order.  On a one-GPU box both contexts sit on device 0.
    Heterogeneous frames (the corpus: balanced by C + D) and frames of one size (equal counts), against the goldens /
    the generator's checksums."""
    import torch
    from tests.conftest import check_expected
    from tools import synth_binding as sb
    devs = [0, 1] if torch.cuda.device_count() >= 2 else [0, 0]
    frames = [comp for _, comp, *_ in corpus]
    ranges = z.shard_frames(frames, 2)
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == len(frames) and 0 < ranges[0][1] < len(frames)
    for kw in ({}, {"device_plan": True}):
        outs, sts = z.decode_frames(frames, devices=devs, **kw)
        assert sts == [0] * len(frames)
        for (name, comp, length, sha, exp), got in zip(corpus, outs):
            check_expected(name, got, length, sha, exp)
    blob, off, ln, ck, ns = sb.make_batch(4, 5, 257, frame_bytes=16384, threads=4)
    fr = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
    r3 = z.shard_frames(fr, 3)  # (frames of one regenerated size, compressed sizes that differ a little: balanced by C + D)
    assert r3[0][0] == 0 and r3[2][1] == 257 and r3[0][1] == r3[1][0] and r3[1][1] == r3[2][0] and all(80 <= hi - lo <= 92 for lo, hi in r3)
    outs, errs = z.DecodeFrames(fr, devices=devs + [0])  # three contexts
    assert errs == [None] * len(fr)
    for o, k in zip(outs, ck):
        assert len(o) == 16384 and sb.checksum64(o) == int(k)
    # a damaged frame keeps its place and its status
    bad = list(fr[:9])
    bad[7] = bad[7][:100]
    outs, sts = z.decode_frames(bad, devices=devs)
    assert [s == 0 for s in sts] == [True] * 7 + [False] + [True] and outs[7] is None and outs[8] is not None


def test_small_frames_small_sequence_tables_two_workgroups_per_cu(ctx):
    """131072 frames of 4 KiB: ~400 sequences each, so the batch's largest LL / ML / OF tables are small, k_seq_q4 sizes
    its chains' LDS slots to them and two of its workgroups share a CU (and every chain's last step, the loop variant
    near the end of a chain and the per-workgroup staging weigh far more than in the 128 KiB configs)."""
    st = _run_full_config(4, 131072, ctx, frame_bytes=4096)
    assert st.n_sequences > 131072 * 200


# ---- BASELINE configs[4]: the sharded bench, launched by bench.py itself

@pytest.mark.parametrize("extra", [[], ["--weak"]])
def test_bench_launches_its_own_ranks(extra):
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns two ranks before touching the GPU;
    each decodes its own contiguous frame range (framedecompressor.go:42-52: frames share nothing), rank 0 prints
    one line that names both ranks.  Default = BASELINE configs[4]: ONE batch split over the ranks ("strong");
    --weak keeps the batch per GPU.  The ranks meet over gloo (no RCCL anywhere); on a one-GPU box both share device 0."""
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env.update(MZD_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--frames-per-gpu", "2048", "--cpu-seconds", "0", "--no-ceiling"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["bit_exact"] is True
    assert sorted(p["rank"] for p in line["per_gpu"]) == [0, 1]
    per = 2048 if extra else 1024
    assert all(p["frames"] == per for p in line["per_gpu"]) and line["scaling"] == ("weak" if extra else "strong")
    assert line["config"]["rendezvous"] == "gloo" and line["config"]["frames"] == 2 * per
    assert line["value"] > 0 and all(p["algorithmic_GBs"] > 0 for p in line["per_gpu"])
    assert line["aggregate"]["devices_distinct"] is (torch.cuda.device_count() >= 2) and line["aggregate"]["hbm_frac_of_all_gpus"] > 0


def test_two_distinct_gpus_product_entry_and_bench(corpus):
    """The N > 1 path on DISTINCT devices (framedecompressor.go:42-52: frames share nothing, so a device takes a contiguous range and
    there is no collective): the product's entry `decode_frames(devices=[0, 1])` against the goldens, and `bench.py --gpus 2` with a
    rank per physical GPU -- the rows must name two devices.  Skipped, with the reason, on a one-GPU box: there the same code runs
    with both contexts / ranks on device 0 (the two tests above), and `hipSetDevice(1)` never executes."""
    import torch
    from tests.conftest import check_expected
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU on this box: the N > 1 path on distinct devices needs two")
    frames = [comp for _, comp, *_ in corpus]
    for kw in ({}, {"device_plan": True}):
        outs, sts = z.decode_frames(frames, devices=[0, 1], **kw)
        assert sts == [0] * len(frames)
        for (name, comp, length, sha, exp), got in zip(corpus, outs):
            check_expected(name, got, length, sha, exp)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "MZD_BENCH_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--frames-per-gpu", "4096", "--cpu-seconds", "0", "--no-ceiling"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["bit_exact"] is True
    assert sorted(p["device"] for p in line["per_gpu"]) == [0, 1] and line["aggregate"]["devices_distinct"] is True
    assert all(p["frames"] == 2048 and p["hbm_frac"] > 0 for p in line["per_gpu"])


def test_bench_real_data_line_small():
    """`bench.py --workload corpus`: the reference's own decodecorpus frames (real zstd output: every block type, table mode, window
    size; multi-block frames), replicated at distinct addresses, every frame checked by status and length and a sample by the
    manifest's sha256.  A heterogeneous batch: its work lists are ordered by size at upload and its frames executed largest first."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "corpus", "--corpus-gib", "0.06", "--steps", "2",
                        "--warmup", "1", "--cpu-seconds", "0", "--no-ceiling"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["bit_exact"] is True and line["data"].startswith("real") and line["config"]["frames"] >= 500
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
