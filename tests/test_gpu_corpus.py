"""-m gpu: the HIP path, called through the C-ABI, against the reference's golden corpus and the
CPU oracle (bit-exact)."""
import ctypes
import io
import os

import numpy as np
import pytest

import sparkzstd_amd as z
from tests.conftest import check_expected

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1, 2, 3, 4, 5], ids=["exec_auto", "k_exec", "k_exec_b", "blocks_k_exec_b", "blocks_k_exec_c_jobs_of_four", "k_exec_c"])
def ctx(request):
    """every test that takes `ctx` runs with the three execution kernels, the default choice between them, and with the blocks of
    every frame executed side by side (block mode: what a batch of few large frames takes by default) -- with k_exec_b's passes as
    jobs of one block, and with k_exec_c's passes, all in one launch, as jobs of four consecutive blocks"""
    return z.Context(0, exec_variant=request.param)


def _decode(frames, ctx):
    outs, sts = z.decode_frames(frames, ctx)
    return outs, sts


@pytest.mark.parametrize("seq_variant,exec_threads,exec_variant",
                         [(0, 256, 1), (1, 256, 1), (3, 256, 1), (0, 128, 1), (1, 128, 1), (3, 128, 1), (0, 64, 1), (1, 64, 1), (3, 64, 1),
                          (0, 0, 2), (1, 0, 2), (3, 0, 2), (0, 0, 0), (0, 0, 3), (3, 0, 3), (0, 0, 4), (0, 0, 5), (1, 0, 5), (3, 0, 5)])
def test_decodecorpus_bit_exact_on_gpu(corpus, seq_variant, exec_threads, exec_variant):
    """All 100 golden frames in ONE device batch: multi-block frames, cross-block matches,
    Repeat/Treeless tables, RLE modes, 1-stream literals, windows < 128 KiB."""
    c = z.Context(0, seq_variant=seq_variant, exec_threads=exec_threads, exec_variant=exec_variant)
    outs, sts = _decode([comp for _, comp, *_ in corpus], c)
    bad = [(corpus[i][0], sts[i]) for i in range(len(corpus)) if sts[i] != 0]
    assert not bad, bad
    for (name, comp, length, sha, exp), got in zip(corpus, outs):
        check_expected(name, got, length, sha, exp)
    c.close()


@pytest.mark.parametrize("assume_cus", [1, 3, 20])
def test_split_batch_two_streams(corpus, assume_cus):
    """k_seq(tail) on the caller's stream overlapped with k_exec(head) on the library's second
    stream: force the split on a small batch by pretending the device has few CUs.  The split is for batches whose work
    lists are in frame order (homogeneous frames: here 300 text-like ones of 8 KiB); the corpus, whose lists are ordered by
    size at upload, runs its stages one after the other through the same context."""
    from tools import synth_binding as sb
    c = z.Context(0, assume_cus=assume_cus)
    blob, off, ln, ck, ns = sb.make_batch(4, 11, 300, frame_bytes=8192, threads=4)
    frames = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
    for _ in range(2):  # twice: the resident state is reset per run
        outs, sts = _decode(frames, c)
        assert sts == [0] * len(frames)
        for o, k in zip(outs, ck):
            assert len(o) == 8192 and sb.checksum64(o) == int(k)
        outs, sts = _decode([comp for _, comp, *_ in corpus], c)
        assert sts == [0] * len(corpus)
        for (name, comp, length, sha, exp), got in zip(corpus, outs):
            check_expected(name, got, length, sha, exp)
    c.close()


def test_synthetic_configs_small(ctx, oracle):
    """BASELINE configs 2-4 at small size: device output == original content == oracle output."""
    from tools import synth_binding as sb
    for config in (2, 3, 4):
        blob, off, ln, ck, ns = sb.make_batch(config, 7, 24, frame_bytes=131072, threads=4)
        frames = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
        outs, sts = _decode(frames, ctx)
        assert sts == [0] * len(frames), (config, sts)
        for f, o, c in zip(frames, outs, ck):
            assert sb.checksum64(o) == int(c)
            rc, want, _, _ = oracle.decode_frame(f, cap=131072 + 64)
            assert rc == 0 and o == want


def test_multi_block_synthetic_frames(ctx):
    """frames of up to 1 MiB (8 blocks): offset history carried across blocks (symbolic in k_seq),
    raw blocks in between, block buffers reused."""
    from tools import synth_binding as sb
    frames, want = [], []
    for kind, n in [(sb.TEXT, 1 << 20), (sb.EXP, 400000), (sb.TEXT, 131073), (sb.RANDOM, 300000), (sb.TEXT, 262144)]:
        d = sb.generate(kind, n, n)
        frames.append(sb.compress(d, sb.MODE_FULL)[0])
        want.append(d)
    outs, sts = _decode(frames, ctx)
    assert sts == [0] * len(frames)
    assert outs == want


from tests.frame_splice import frame_blocks as _frame_blocks, splice_frame as _splice_frame, literal_block as _literal_block  # noqa: E402


def test_heterogeneous_batch_in_two_groups_of_frames(corpus):
    """A batch whose sequence stage is as long as its longest chain (the reference's corpus: one chain of 42 k sequences among 2 445)
    is decoded and executed in two groups of frames on two streams -- the frames that hold the long chains, and the others beside
    them (mzd_batch_upload groups them, MZD_PASS_TWO_GROUPS says so); the same bytes and statuses as with the frames in one group
    (a batch of frames of one kind takes no grouping), planned on the host and on the device, damaged frames included, run after run."""
    from sparkzstd_amd import _lib, api
    from tools import synth_binding as sb
    frames = [comp for _, comp, *_ in corpus] * 3
    rng = np.random.default_rng(5)
    for k in (7, 33, 91, 97, 133):  # (damaged copies, some of them of the frames with the long chains)
        b = bytearray(frames[k])
        b[len(b) // 2 + int(rng.integers(0, 64))] ^= 0x5A
        frames[k] = bytes(b)
    c = z.Context(0)
    rb, _, _, sts_r = api.decode_frames_resident(frames, c)
    assert rb.pass_flags & _lib.MZD_PASS_TWO_GROUPS and rb.pass_flags & _lib.MZD_PASS_EXEC_C
    rb.free()
    rb3, _, _, sts_r3 = api.decode_frames_resident(frames, c, device_plan=True)
    assert rb3.pass_flags & _lib.MZD_PASS_TWO_GROUPS  # (planned on the device: the same grouping from the keys the device hands back)
    rb3.free()
    outs, sts = z.decode_frames(frames, c)
    outs2, sts2 = z.decode_frames(frames, c)  # (run after run: the third stream's work is ordered behind the pass before)
    outs3, sts3 = z.decode_frames(frames, c, device_plan=True)
    assert list(sts) == list(sts2) == list(sts3) == list(sts_r) == list(sts_r3)
    assert any(s != 0 for s in sts) and sum(1 for s in sts if s == 0) >= len(frames) - 5
    for i, (a, b, d) in enumerate(zip(outs, outs2, outs3)):
        if sts[i] == 0:
            assert bytes(a) == bytes(b) == bytes(d), i
    for i in range(100):
        if i not in (7, 33, 91, 97):
            name, comp, length, sha, exp = corpus[i]
            check_expected(name, bytes(outs[i]), length, sha, exp)
    # frames of one kind: one group
    blob, off, ln, ck, ns = sb.make_batch(4, 0, 512, 131072, threads=4)
    fr = [bytes(blob[int(o):int(o) + int(l)]) for o, l in zip(off, ln)]
    rb, _, _, sts = api.decode_frames_resident(fr, c)
    assert not rb.pass_flags & _lib.MZD_PASS_TWO_GROUPS and list(sts) == [0] * 512
    rb.free()
    c.close()


def test_small_blocks_without_sequences_between_blocks_with_matches(ctx, oracle):
    """k_exec_c takes small Raw / RLE / literal-only blocks through its ring like a run of literals (up to 2 KiB; larger ones are
    copied to the slab and the ring reloaded).  Frames spliced from the blocks of multi-block text frames -- their matches and
    their offset history reach back over what is put between them -- with such blocks of every size around the path's
    boundaries (a window unit is 512 bytes, a lane carries 8) in between, several in a row, first and last; against the oracle."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(11)
    sizes = [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 500, 511, 512, 513, 1000, 1023, 1024, 1025, 1536, 2047, 2048, 2049, 3000, 5000]
    huf_only = []  # compressed blocks with Huffman literals and no sequences (the Huffman stage writes them in place)
    for k, n in enumerate([300, 700, 1500, 2048, 2500, 6000]):
        huf_only += [b for b in _frame_blocks(sb.compress(sb.generate(sb.TEXT, 900 + k, n), sb.MODE_LITERALS)[0]) if b[0] == 2]
    assert huf_only

    def small(n):
        kind = int(rng.integers(0, 5))
        data = bytes(rng.integers(0, 256, size=max(n, 1), dtype=np.uint8))[:n]
        if kind == 0 or n == 0:
            return (0, data, n)                      # Raw
        if kind == 1:
            return (1, data[:1], n)                  # RLE
        if kind == 2:
            return _literal_block(data)              # raw literals, no sequences
        if kind == 3:
            return _literal_block(data, rle=True)    # RLE literals, no sequences
        return huf_only[int(rng.integers(0, len(huf_only)))]

    frames = []
    for i in range(10):
        src = _frame_blocks(sb.compress(sb.generate(sb.TEXT, 700 + i, (3 + i % 3) * 131072 - 1000 * i), sb.MODE_FULL)[0])
        assert len(src) >= 3 and all(t == 2 for t, _, _ in src)
        blocks = [small(sizes[(5 * i + j) % len(sizes)]) for j in range(i % 3)]  # (some frames start with small blocks)
        for j, b in enumerate(src):
            blocks.append(b)
            blocks += [small(sizes[int(rng.integers(0, len(sizes)))]) for _ in range(1 + (i + j) % 4)]
        frames.append(_splice_frame(blocks))
    # every size once, in one frame, between two blocks with sequences
    src = _frame_blocks(sb.compress(sb.generate(sb.TEXT, 799, 2 * 131072), sb.MODE_FULL)[0])
    for kind_blocks in ([(0, bytes(rng.integers(0, 256, size=max(n, 1), dtype=np.uint8))[:n], n) for n in sizes],
                        [(1, b"\x5a", n) for n in sizes if n],
                        [_literal_block(bytes(rng.integers(0, 256, size=n, dtype=np.uint8))) for n in sizes if n]):
        frames.append(_splice_frame([src[0]] + kind_blocks + [src[1]]))
    want = []
    for f in frames:
        rc, ref, *_ = oracle.decode_frame(f, cap=8 << 20)
        assert rc == 0, rc
        want.append(ref)
    outs, sts = _decode(frames, ctx)
    assert sts == [0] * len(frames), sts
    for i, (o, w) in enumerate(zip(outs, want)):
        assert o == w, i
    outs, sts = z.decode_frames(frames, ctx, device_plan=True)
    assert sts == [0] * len(frames) and all(o == w for o, w in zip(outs, want))


@pytest.mark.parametrize("exec_variant", [0, 3, 4])
def test_large_frames_blocks_side_by_side(exec_variant):
    """Few large frames: the blocks of a frame are executed side by side (mzd_exec_blk.hip) -- three passes for frames below
    8 MiB, four above.  Text-like frames of 1, 9 and 20 MiB (up to 160 blocks, matches reaching back over many block starts,
    repeat offsets carried across blocks) against the generator's content; by default (exec_variant 0) such a batch takes
    block mode by itself, 3 forces it with jobs of one block, 4 with jobs of four consecutive blocks."""
    from tools import synth_binding as sb
    frames, want = [], []
    for kind, n in [(sb.TEXT, 9 << 20), (sb.TEXT, 1 << 20), (sb.TEXT, (20 << 20) + 12345), (sb.EXP, 3 << 20), (sb.RANDOM, 700000)]:
        d = sb.generate(kind, n ^ 0x5bd1, n)
        frames.append(sb.compress(d, sb.MODE_FULL)[0])
        want.append(d)
    c = z.Context(0, exec_variant=exec_variant)
    for _ in range(2):
        outs, sts = _decode(frames, c)
        assert sts == [0] * len(frames)
        for i, (o, w) in enumerate(zip(outs, want)):
            assert len(o) == len(w), i
            if o != w:
                a = np.frombuffer(o, np.uint8)
                b = np.frombuffer(w, np.uint8)
                bad = np.nonzero(a != b)[0]
                raise AssertionError((i, len(bad), bad[:8].tolist()))
    c.close()


@pytest.mark.parametrize("exec_variant", [0, 3, 4])
def test_block_mode_matches_that_reach_back_more_than_8_mib(exec_variant):
    """Block mode spells a derived byte's origin with three passes (23 bits of the position) when the frame's matches reach back
    less than 8 MiB -- k_seq_q4 reports every block's largest offset, k_blk_scan decides per frame -- and with a fourth pass for
    the bits above otherwise.  A frame whose incompressible 200 kB head is matched again 10 MiB and 22 MiB later (zeros in
    between leave the compressor's hash table alone) needs the fourth; a text-like frame of 20 MiB beside it in the same batch
    does not; both must come out byte for byte, together and alone."""
    from tools import synth_binding as sb
    A = sb.generate(sb.RANDOM, 99, 200000)
    T = sb.generate(sb.TEXT, 98, 3 << 20)
    far = A + bytes(10 << 20) + A[:150000] + T + bytes(9 << 20) + A[50000:] + T[:1 << 20]
    near = sb.generate(sb.TEXT, 97, (20 << 20) + 777)
    cfar, cnear = sb.compress(far, sb.MODE_FULL)[0], sb.compress(near, sb.MODE_FULL)[0]
    assert len(cfar) < 1400000  # (the later copies of A were found: 1.2 MB; 1.8 MB without them)
    c = z.Context(0, exec_variant=exec_variant)
    for frames, want in ([cfar, cnear], [far, near]), ([cnear], [near]), ([cfar], [far]), ([cnear, cfar, cnear], [near, far, near]):
        outs, sts = _decode(frames, c)
        assert sts == [0] * len(frames)
        for i, (o, w) in enumerate(zip(outs, want)):
            assert len(o) == len(w), i
            if o != w:
                a, b = np.frombuffer(o, np.uint8), np.frombuffer(w, np.uint8)
                bad = np.nonzero(a != b)[0]
                raise AssertionError((i, len(bad), bad[:8].tolist()))
    c.close()


@pytest.mark.parametrize("count,frame_bytes,what", [
    (1, 64 << 20, "one frame of 512 blocks: jobs of two blocks, every pass in one launch, 64 fix-up workgroups over all XCDs"),
    (17, 4 << 20, "17 frames: jobs of two blocks, sixteen fix-up workgroups per frame over all XCDs"),
    (70, 3 << 20, "more than 64 frames: a frame's fix-up workgroups on one XCD, a launch with eight times the workgroups"),
    (5, 24 << 20, "five frames of 192 blocks (four planes): jobs of two, 64 fix-up workgroups per frame"),
])
def test_block_mode_policy_branches(count, frame_bytes, what):
    """exec_variant 0 on batches that take block mode by themselves, one per branch of its launch policy (blocks per job, passes in
    one launch or one per pass, fix-up workgroups per frame, on one XCD or on all): every frame against the generator's checksum."""
    from tools import synth_binding as sb
    blob, off, ln, ck, ns = sb.make_batch(4, 700 + count, count, frame_bytes=frame_bytes, threads=8)
    frames = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
    c = z.Context(0)
    for _ in range(2):
        outs, sts = _decode(frames, c)
        assert sts == [0] * count, what
        for o, k in zip(outs, ck):
            assert len(o) == frame_bytes and sb.checksum64(o) == int(k), what
    c.close()


def test_block_mode_mixed_batch_walks_only_the_multi_block_frames():
    """A batch of many single-block frames and a few large ones: the fix-up walk is launched for the frames of more than one block
    (k_blk_scan lists them) with the workgroup count a batch of THOSE frames would get -- 1 200 frames of 128 KiB and two of 24 MiB
    are a batch of two frames for it, not of 1 202 (one workgroup per frame: ten times slower) -- in any order of the frames."""
    from tools import synth_binding as sb
    blob, off, ln, ck, ns = sb.make_batch(4, 31, 1200, frame_bytes=131072, threads=8)
    small = [(blob[int(o):int(o + l)].tobytes(), 131072, int(k)) for o, l, k in zip(off, ln, ck)]
    blob2, off2, ln2, ck2, ns2 = sb.make_batch(4, 77, 2, frame_bytes=24 << 20, threads=2)
    large = [(blob2[int(o):int(o + l)].tobytes(), 24 << 20, int(k)) for o, l, k in zip(off2, ln2, ck2)]
    c = z.Context(0)
    for order in (small[:600] + large[:1] + small[600:] + large[1:], large + small, small + large):
        outs, sts = _decode([f for f, _, _ in order], c)
        assert sts == [0] * len(order)
        for o, (_, n, k) in zip(outs, order):
            assert len(o) == n and sb.checksum64(o) == k
    c.close()


def test_block_mode_many_frames_one_fixup_workgroup_each():
    """1 100 two-block frames with blocks side by side: with that many frames the fix-up walk runs ONE workgroup per frame
    (no waiting between workgroups); fewer frames take several per frame (the tests above)."""
    from tools import synth_binding as sb
    blob, off, ln, ck, ns = sb.make_batch(4, 23, 1100, frame_bytes=163840, threads=4)
    frames = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
    c = z.Context(0, exec_variant=3)
    outs, sts = _decode(frames, c)
    assert sts == [0] * len(frames)
    for o, k in zip(outs, ck):
        assert len(o) == 163840 and sb.checksum64(o) == int(k)
    c.close()


def test_block_mode_fixup_rescue_when_workgroups_give_up():
    """The fix-up walk of block mode lets G workgroups of a frame wait for each other; nothing promises that they are all
    resident (another stream or process on the GPU).  mzd_debug_force_fixup_bail makes workgroup 1 of every frame give up at a
    step of the walk as it would after its bounded wait; the rescue launch (one workgroup per such frame, redoing exactly the
    chunk sets that were not gathered) then has to deliver the frames whole -- never MZD_ERR_DEVICE, never a wrong byte."""
    from tools import synth_binding as sb
    from sparkzstd_amd import _lib
    L = _lib.load()
    frames, datas = [], []
    for k, n in enumerate([20 << 20, 9 << 20, 3 << 20 | 12345]):
        d = sb.generate(sb.TEXT, 40 + k, n)
        frames.append(sb.compress(d, sb.MODE_FULL)[0])
        datas.append(d)
    for variant in (3, 4):
        for step in (1, 2, 7):
            c = z.Context(0, exec_variant=variant)
            assert L.mzd_debug_force_fixup_bail(c._c, step) == 0
            outs, sts = _decode(frames, c)
            assert sts == [0, 0, 0], (variant, step, sts)
            for o, d in zip(outs, datas):
                assert o == d, (variant, step)
            assert L.mzd_debug_force_fixup_bail(c._c, 0) == 0
            outs, sts = _decode(frames, c)
            assert sts == [0, 0, 0] and all(o == d for o, d in zip(outs, datas))
            c.close()


def test_block_mode_reports_status_and_length_of_the_serial_walk(corpus, oracle):
    """Damaged multi-block frames: block mode (a scan over the block summaries + per-block execution) names the same status
    and the same produced length per frame as k_exec_b walking the blocks in order -- the first failing block ends the frame
    (framedecompressor.go:246-254), whatever later blocks would have reported -- and both agree with the oracle: a frame the
    device reports as decoded is one the oracle decodes to the same bytes, a frame the oracle rejects carries a status."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(77)
    base = []
    for kind, n in [(sb.TEXT, 700000), (sb.EXP, 500000), (sb.TEXT, 300000)]:
        base.append(sb.compress(sb.generate(kind, n + 3, n), sb.MODE_FULL)[0])
    base += [bytes(comp) for _, comp, length, *_ in corpus if length > 200000][:6]
    frames = list(base)
    for f in base:
        for k in range(12):
            b = bytearray(f)
            for pos in rng.integers(len(b) // 8, len(b), size=1 + k % 2):
                b[int(pos)] ^= int(rng.integers(1, 256))
            frames.append(bytes(b))
    res = []
    for variant in (2, 3, 4, 5):
        c = z.Context(0, exec_variant=variant)
        plan = z.Plan()
        for f in frames:
            plan.add_frame(f)
        batch = plan.finalize()
        offs = [int(batch.frames[i].out_offset) for i in range(len(frames))]
        rb = c.upload(batch)
        rb.run()
        out, status, out_len = rb.download()
        res.append((np.array(status).copy(), np.array(out_len).copy(), np.array(out).copy()))
        rb.free()
        plan.close()
        c.close()
    (s2, l2, o2) = res[0]
    assert (s2 != 0).any() and (s2 == 0).any()
    for i, f in enumerate(frames):
        rc, want, _, _ = oracle.decode_frame(f, cap=4 << 20)
        if s2[i] == 0:
            assert rc == 0, (i, oracle.strerror(rc))
            assert int(l2[i]) == len(want) and bytes(o2[offs[i]:offs[i] + len(want)]) == want, i
        if rc != 0:
            assert s2[i] != 0, i
    for (s3, l3, o3) in res[1:]:
        assert (s2 == s3).all(), np.nonzero(s2 != s3)[0][:8]
        assert (l2 == l3).all(), np.nonzero(l2 != l3)[0][:8]
        for i in np.nonzero(s2 == 0)[0]:
            a, b = offs[i], offs[i] + int(l2[i])
            assert (o2[a:b] == o3[a:b]).all(), i


def test_randomized_differential(ctx):
    """400 frames of random kind / size / mode in ONE batch (ragged, empty, multi-block, mixed block
    types), checked byte for byte against the content they were made from."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(20261001)
    frames, want = [], []
    for i in range(400):
        kind = int(rng.choice([sb.TEXT, sb.EXP, sb.RANDOM, sb.ZERO]))
        n = int(rng.choice([0, 1, 2, 3, 7, 63, 64, 65, 255, 1000, 4095, 4096, 4097, 20000, 65535, 131071, 131072,
                            131073, 200000, 300000]))
        if rng.random() < 0.3:
            n = int(rng.integers(0, 150000))
        data = sb.generate(kind, 1000 + i, n)
        if kind == sb.TEXT and n > 100 and rng.random() < 0.3:  # splice in periodic / repeated regions
            k = int(rng.integers(1, 40))
            data = data[:n // 3] + (data[:k] * (n // k + 1))[:n // 3] + data[2 * (n // 3):]
            data = data[:n] + b"\0" * (n - len(data[:n]))
        mode = int(rng.choice([sb.MODE_FULL, sb.MODE_FULL, sb.MODE_LITERALS, sb.MODE_RAW, sb.MODE_RLE]))
        if mode == sb.MODE_RLE:
            data = bytes([data[0] if data else 0]) * n
        if mode == sb.MODE_LITERALS and n > 131072:
            mode = sb.MODE_FULL
        frames.append(sb.compress(data, mode)[0])
        want.append(data)
    outs, sts = _decode(frames, ctx)
    bad = [(i, s) for i, s in enumerate(sts) if s != 0]
    assert not bad, bad[:10]
    for i, (o, w) in enumerate(zip(outs, want)):
        assert o == w, f"frame {i}: {len(o)} vs {len(w)} bytes"


def test_long_matches_and_rle_literals(ctx, oracle):
    """all-zero / periodic content: one sequence with a 128 KiB overlapping match (oversized tile
    path), RLE literals, offsets 1..7."""
    from tools import synth_binding as sb
    frames, want = [], []
    for period in (1, 2, 3, 5, 7, 64, 100, 4097):
        data = (bytes(range(1, period + 1)) * (131072 // period + 1))[:131072] if period <= 255 else \
            (sb.generate(sb.RANDOM, period, period) * (131072 // period + 1))[:131072]
        frames.append(sb.compress(data, sb.MODE_FULL)[0])
        want.append(data)
    outs, sts = _decode(frames, ctx)
    assert sts == [0] * len(frames)
    assert outs == want


def test_gpu_matches_oracle_per_frame(corpus, oracle, ctx):
    frames = [comp for _, comp, *_ in corpus[:20]]
    outs, sts = _decode(frames, ctx)
    for comp, got, st in zip(frames, outs, sts):
        rc, want, _, _ = oracle.decode_frame(comp, cap=2 << 20)
        assert rc == 0 and st == 0
        assert got == want


def test_frame_reader_mirror(corpus, ctx):
    """FrameReader.Read semantics (framereader.go:51-109) over the device path."""
    name, comp, length, sha, exp = corpus[3]
    fr = z.NewFrameReader(io.BytesIO(comp), ctx)
    got = bytearray()
    while True:
        chunk = fr.Read(4096)
        if not chunk:
            break
        got += chunk
    check_expected(name, bytes(got), length, sha, exp)
    sink = io.BytesIO()
    fd = z.NewFrameDecompressor(io.BytesIO(comp), sink, ctx)
    fd.Decompress()
    check_expected(name, sink.getvalue(), length, sha, exp)


def test_frame_reader_serves_reads_from_device_memory(ctx):
    """The reader keeps the decoded frame in HBM (mzd_batch_read_out): large Reads and readinto go straight from there into the
    bytes they hand out, small ones through a 4 MiB window.  A 9.5 MiB frame read in every mixture of the two -- window
    refills, a large Read that starts inside the window, readinto into small and large buffers, readall after partial reads,
    EOF twice, Reset half way -- returns the frame's bytes, and the device memory is gone once the last byte is out."""
    from tools import synth_binding as sb
    data = sb.generate(sb.TEXT, 314, (9 << 20) + 500001)
    comp = sb.compress(data, sb.MODE_FULL)[0]
    rng = np.random.default_rng(5)
    for mode in range(5):
        r = z.NewFrameReader(io.BytesIO(comp), ctx)
        got = bytearray()
        if mode == 0:      # everything at once
            got += r.read()
        elif mode == 1:    # small Reads only: the window is refilled three times
            while True:
                d = r.Read(int(rng.integers(1, 200000)))
                if not d:
                    break
                got += d
        elif mode == 2:    # small, large, small ...: a large Read inside the window is served from it up to its end (a short read)
            sizes = [100, 2 << 20, 7, 3 << 20, 1 << 20, 50000, 5 << 20, 1 << 30]
            for n in sizes:
                d = r.Read(n)
                assert len(d) <= n
                got += d
            got += r.readall()
        elif mode == 3:    # readinto: a large buffer (straight from the device), then small ones
            big = bytearray(3 << 20)
            k = r.readinto(big)
            assert k == len(big)
            got += big[:k]
            small = bytearray(65536)
            while True:
                k = r.readinto(small)
                if k == 0:
                    break
                got += small[:k]
        else:              # Reset half way: the old frame's memory is released, the new one starts from its first byte
            got += r.Read(3 << 20)
            assert r._rb is not None
            r.Reset(io.BytesIO(comp))
            assert r._rb is None
            got = bytearray(r.read())
        assert bytes(got) == data, mode
        assert r.Read(10) == b"" and r.Read(1 << 21) == b"" and r.readinto(bytearray(8)) == 0
        assert r._rb is None  # drained: the batch was freed
    # an empty frame and a damaged one
    empty = sb.compress(b"", sb.MODE_FULL)[0]
    r = z.NewFrameReader(io.BytesIO(empty), ctx)
    assert r.read() == b"" and r._rb is None
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x55
    r = z.NewFrameReader(io.BytesIO(bytes(bad)), ctx)
    # (a flipped byte may still decode -- a raw literal, say: the reader then returns what the reference's algorithm makes of the
    # damaged frame, and raises exactly when that fails)
    from tests.oracle_binding import load_oracle
    rc, ref, _, _ = load_oracle().decode_frame(bytes(bad), cap=len(data) + (1 << 20))
    try:
        out = r.read()
        assert rc == 0 and out == ref
    except z.ZstdError:
        assert rc != 0 and r._rb is None


@pytest.mark.parametrize("flags", [[], ["--device-plan"], ["--devices", "0,0"], ["--device-plan", "--devices", "0,0,0"], ["--batch-reader", "7"],
                                   ["--chunk", "131072"], ["--chunk", "1000000"], ["--decompressor", "262144"]])
def test_cpp_frame_reader_verify_cli(corpus, flags):
    """The C++ mirror of FrameReader (include/sparkzstd_frame.hpp) driven like the reference's own
    harness cmd/sparkzstd/main.go: decode x.zst, compare byte for byte with x.  --chunk: the shared reader in chunk mode (ABI 9:
    every frame through the device in chunks of whole blocks); --decompressor: FrameDecompressor.DecodeNextBlock, a chunk per call."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "verify", "sparkzstd_verify")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tools", "verify")])
    d = os.path.join(root, "tests", "golden", "decodecorpus")
    files = [os.path.join(d, name + ".zst") for name, _, _, _, exp in corpus if exp is not None]
    assert len(files) >= 30
    r = subprocess.run([exe] + flags + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Found no diffs in any files" in r.stdout and "Found no unexpected errors" in r.stdout


@pytest.mark.parametrize("lookahead", [1, 7, 64])
def test_cpp_batch_frame_reader_on_the_streaming_path(corpus, lookahead):
    """sparkzstd::BatchFrameReader (round 6: a worker thread, pinned buffers, two batches in flight on mzd_stream_*) beyond the happy
    path, in C++ (`sparkzstd_verify --reader-selftest`): Read and View against DecodeFrames' bytes, a frame cut in half (its error at
    ITS read, the frames behind it served), a wrong magic number (Reset's error, framereader.go:35-49), sources enqueued while the
    reader is being served, a reader destroyed with batches in flight."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "verify", "sparkzstd_verify")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "tools", "verify")])
    d = os.path.join(root, "tests", "golden", "decodecorpus")
    files = [os.path.join(d, name + ".zst") for name, *_ in corpus]
    r = subprocess.run([exe, "--reader-selftest", str(lookahead)] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "reader selftest ok" in r.stdout, r.stdout + r.stderr


def test_batch_frame_reader_serves_frames_in_order(corpus):
    """BatchFrameReader: the reference harness's pattern (ONE reader, Reset per frame: framereader.go:35,
    cmd/sparkzstd/main.go:59,126) with the frames to come known to the reader -- it reads ahead, decodes `lookahead` frames per
    device batch and serves them in order.  Ragged batches (lookahead 7 over 100 frames), sources as file objects and as
    bytes, a damaged frame in the middle (its error surfaces at ITS read; the frames behind it are unaffected), an
    explicit Reset(source), and EOF."""
    frames = [comp for _, comp, *_ in corpus]
    srcs = [io.BytesIO(f) if i % 2 else f for i, f in enumerate(frames)]
    bad_at = 13
    srcs[bad_at] = io.BytesIO(frames[bad_at][:len(frames[bad_at]) // 2])
    r = z.BatchFrameReader(iter(srcs), lookahead=7)
    n = 0
    while r.Reset():
        if n == bad_at:
            with pytest.raises(z.ZstdError):
                r.Read(10)
        else:
            name, comp, length, sha, exp = corpus[n]
            got = bytearray()
            if n % 3 == 2:  # the frame's bytes lent in place (a memoryview of the pinned output buffer: round 6), then EOF
                got += r.View()
                assert r.Read(10) == b""
            while n % 3 != 2:
                d = r.Read(50000)  # short reads, like a caller with a fixed buffer
                if not d:
                    break
                got += d
            check_expected(name, bytes(got), length, sha, exp)
        n += 1
    assert n == len(frames) and r.frames_served == len(frames) and r.Read(1) == b""
    # Enqueue as the frames become known; Reset(source) puts a frame in front; a wrong magic number is Reset's error
    r = z.NewBatchFrameReader(lookahead=4)
    for f in frames[:6]:
        r.Enqueue(f)
    assert r.Reset(io.BytesIO(frames[50]))
    check_expected(corpus[50][0], r.read(), corpus[50][2], corpus[50][3], corpus[50][4])
    r.Enqueue(b"\x00\x01\x02\x03 not a frame")
    for k in range(6):
        assert r.Reset()
        check_expected(corpus[k][0], r.read(), corpus[k][2], corpus[k][3], corpus[k][4])
    with pytest.raises(z.ZstdError):
        r.Reset()
    assert not r.Reset()


def test_raw_rle_frames(ctx):
    """BASELINE config 2 shape, small: raw / rle single-block frames incl. sizes 0, 1, 15, 16, 17, 131072."""
    frames, want = [], []
    rng = np.random.default_rng(7)
    for i, n in enumerate([0, 1, 15, 16, 17, 255, 4096, 65537, 131072]):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        frames.append(b"\x28\xb5\x2f\xfd\xa0" + n.to_bytes(4, "little") + ((n << 3) | 1).to_bytes(3, "little") + payload)
        want.append(payload)
        b = bytes([(37 * i + 11) & 255])
        frames.append(b"\x28\xb5\x2f\xfd\xa0" + n.to_bytes(4, "little") + ((n << 3) | 3).to_bytes(3, "little") + b)
        want.append(b * n)
    outs, sts = _decode(frames, ctx)
    assert sts == [0] * len(frames)
    assert outs == want


def test_corrupt_input_reports_status_not_fault(corpus, ctx):
    """Device code never faults on bad input: per-frame status mirrors the reference sentinels."""
    name, comp, length, sha, exp = corpus[10]
    bad = bytearray(comp)
    frames = [bytes(comp)]
    for pos in (len(comp) // 2, len(comp) // 3, len(comp) - 9):
        b2 = bytearray(comp)
        b2[pos] ^= 0x5A
        frames.append(bytes(b2))
    frames.append(bytes(comp[:len(comp) // 2]))  # truncated
    frames.append(b"\x00" * 16)
    outs, sts = _decode(frames, ctx)
    assert sts[0] == 0 and outs[0] is not None
    assert sts[-1] == 2 and sts[-2] != 0
    # corrupted frames either fail with a status or (rarely) still decode; never crash
    for o, s in zip(outs[1:], sts[1:]):
        assert (o is None) == (s != 0)
    # and the context is still usable afterwards
    outs2, sts2 = _decode([bytes(comp)], ctx)
    assert sts2 == [0]
    check_expected(name, outs2[0], length, sha, exp)


@pytest.mark.parametrize("seq_variant,exec_variant", [(0, 1), (1, 1), (3, 1), (0, 2), (1, 2), (0, 3), (0, 4), (0, 5), (1, 5)])
def test_fuzzed_frames_never_fault_and_agree_with_oracle(corpus, oracle, seq_variant, exec_variant):
    """Every corpus frame, mutated 6 times (random byte flips past the frame header, seeded), all
    in ONE device batch.  The device must not fault; a frame it reports as decoded must be one
    the oracle decodes to the same bytes, and a frame the oracle rejects must carry a status."""
    rng = np.random.default_rng(20260101)
    frames = []
    for _, comp, *_ in corpus:
        if len(comp) < 24:
            continue
        for k in range(6):
            b = bytearray(comp)
            nflip = 1 + (k % 3)
            for pos in rng.integers(8, len(b), size=nflip):
                b[int(pos)] ^= int(rng.integers(1, 256))
            frames.append(bytes(b))
    c = z.Context(0, seq_variant=seq_variant, exec_variant=exec_variant)
    outs, sts = _decode(frames, c)
    n_ok = 0
    for f, o, s in zip(frames, outs, sts):
        rc, want, _, _ = oracle.decode_frame(f, cap=4 << 20)
        if s == 0:
            assert rc == 0, (oracle.strerror(rc), len(f))
            assert o == want
            n_ok += 1
        if rc != 0:
            assert s != 0
    assert 0 < n_ok < len(frames)  # some mutations are harmless (unread checksum, literals), most are not
    # the context survives
    outs2, sts2 = _decode([corpus[3][1]], c)
    assert sts2 == [0]


def test_device_built_fse_tables_equal_host_tables(corpus, ctx, oracle):
    """k_fse_build / k_huf_build: every FSE table built on the device from its normalised
    counts and every Huffman table filled from its weights is cell-for-cell the table the host planner
    builds (fse.go:136-230, huffman.go:112-190), over the whole corpus plus synthetic config-3/4
    frames; and the batch decodes to the same bytes in both forms."""
    from sparkzstd_amd import _lib
    from tools import synth_binding as sb
    from tests import fse_build_ref, oracle_binding as ob
    blob, off, ln, ck, ns = sb.make_batch(4, 7, 96, threads=4)
    blob3, off3, ln3, _, _ = sb.make_batch(3, 11, 32, threads=4)  # MaxBits 11 Huffman tables
    frames = [comp for _, comp, *_ in corpus] + [bytes(blob[o:o + l]) for o, l in zip(off, ln)] + \
             [bytes(blob3[o:o + l]) for o, l in zip(off3, ln3)]
    ph, pd = z.Plan(), z.Plan(device_tables=True)
    for f in frames:
        assert ph.add_frame(f)[0] == 0 and pd.add_frame(f)[0] == 0
    bh, bd = ph.finalize(), pd.finalize()
    assert bh.n_fse_tables == bd.n_fse_tables
    host = np.ctypeslib.as_array(ctypes.cast(bh.fse_entries, ctypes.POINTER(ctypes.c_uint32)), shape=(bh.n_fse_entries,)).copy()
    rb = ctx.upload(bd)
    try:
        st = rb.stats()
        n_counts = sum(1 for i in range(bd.n_fse_tables) if bd.fse_tables[i].build & _lib.MZD_FSE_FROM_COUNTS)
        assert st.n_fse_built == n_counts and n_counts > 2000
        n_vs_oracle = 0
        for ti in range(bd.n_fse_tables):
            dh = bh.fse_tables[ti]
            want = host[dh.entries_off:dh.entries_off + (1 << dh.acc_log)]
            got = rb.read_fse_table(ti)
            assert got.shape == want.shape and (got == want).all(), (ti, dh.acc_log, dh.kind)
            # ... and the ORACLE's: orc_fse_build (fse.go:136-230 restated) on the same normalised counts, cell for cell
            # (baseline, number of bits, the untranslated symbol) -- the device against the checker, not only against the host planner
            if bd.fse_tables[ti].build & _lib.MZD_FSE_FROM_COUNTS:
                counts = fse_build_ref.counts_of(bd, ti)
                t = ob.FseTable()
                t.acc_log, t.n_values = dh.acc_log, len(counts)
                for k, cnt in enumerate(counts):
                    t.values[k] = cnt + 1  # (fse.go:19: the value kept is the probability + 1)
                assert oracle.lib.orc_fse_build(ctypes.byref(t), None, 0, None, 0) == 0, ti
                cells = np.array([t.table[i].baseline | (t.table[i].nbits << 16) | (t.table[i].raw_symbol << 24) for i in range(1 << dh.acc_log)], dtype=np.uint32)
                oracle.lib.orc_fse_free(ctypes.byref(t))
                assert (got == cells).all(), ("oracle", ti, dh.acc_log, dh.kind)
                n_vs_oracle += 1
        assert n_vs_oracle == n_counts
        # the same for the Huffman decode tables filled from their weights (k_huf_build, huffman.go:112-190)
        hufh = np.ctypeslib.as_array(ctypes.cast(bh.huf_entries, ctypes.POINTER(ctypes.c_uint16)), shape=(bh.n_huf_entries,)).copy()
        ht = ob.HufTable()
        assert bd.n_huf_tables == bh.n_huf_tables and st.n_huf_built == bd.n_huf_tables > 1000
        for ti in range(bd.n_huf_tables):
            dh, dd = bh.huf_tables[ti], bd.huf_tables[ti]
            assert dd.max_bits & _lib.MZD_HUF_FROM_WEIGHTS and (dd.max_bits & 0xFF) == dh.max_bits
            want = hufh[dh.entries_off:dh.entries_off + (1 << dh.max_bits)]
            got = rb.read_huf_table(ti)
            assert got.shape == want.shape and (got == want).all(), (ti, dh.max_bits)
            # ... and the ORACLE's: orc_huf_build (huffman.go:112-190 restated) on the same weights
            nw = (dd.max_bits >> 8) & 0xFFFF
            ws = (ctypes.c_uint8 * max(nw, 1))()
            for j in range(nw):
                e = bd.huf_entries[dd.entries_off + (j >> 1)]
                ws[j] = e.nbits if j & 1 else e.symbol
            assert oracle.lib.orc_huf_build(ctypes.byref(ht), ws, nw) == 0 and ht.max_bits == dh.max_bits, ti
            cells = np.frombuffer(ht.symbols, dtype=np.uint8, count=1 << dh.max_bits).astype(np.uint16) | \
                (np.frombuffer(ht.nbits, dtype=np.uint8, count=1 << dh.max_bits).astype(np.uint16) << 8)
            assert (got == cells).all(), ("oracle", ti, dh.max_bits)
        rb.run()
        out_d, status_d, len_d = rb.download()
    finally:
        rb.free()
    rb = ctx.upload(bh)
    try:
        rb.run()
        out_h, status_h, len_h = rb.download()
    finally:
        rb.free()
    assert (status_d == 0).all() and (status_h == 0).all() and (len_d == len_h).all()
    assert bytes(out_d) == bytes(out_h)
    ph.close()
    pd.close()


def test_content_checksum_verification_on_device(corpus):
    """k_xxh64 (an extension: the reference never reads the checksum): with
    verify_checksum the 100 corpus frames (all carry one) pass, a frame whose stored checksum is
    altered is reported as MZD_ERR_CHECKSUM (and still decodes without the option), synthetic
    frames with and without checksums mix in one batch."""
    from sparkzstd_amd import _lib
    from tools import synth_binding as sb
    c = z.Context(0, verify_checksum=True)
    frames = [comp for _, comp, *_ in corpus]
    outs, sts = _decode(frames, c)
    assert sts == [0] * len(frames)
    for (name, comp, length, sha, exp), got in zip(corpus, outs):
        check_expected(name, got, length, sha, exp)
    bad = []
    for k in (3, 17, 42, 99):
        f = bytearray(frames[k])
        f[-1 - (k % 4)] ^= 0x40
        bad.append(bytes(f))
    data = [sb.generate(sb.TEXT, 100 + i, 131072 - 37 * i) for i in range(6)]  # lengths with every tail branch
    try:
        sb.set_content_checksum(True)
        with_ck = [sb.compress(d)[0] for d in data]
    finally:
        sb.set_content_checksum(False)
    without = [sb.compress(d)[0] for d in data[:2]]
    broken = bytearray(with_ck[0]); broken[-2] ^= 1
    batch = bad + with_ck + without + [bytes(broken)]
    outs, sts = _decode(batch, c)
    assert sts[:4] == [_lib.MZD_ERR_CHECKSUM] * 4
    assert sts[4:4 + 6 + 2] == [0] * 8 and outs[4:10] == data and outs[10:12] == data[:2]
    assert sts[-1] == _lib.MZD_ERR_CHECKSUM
    # without the option nothing is verified (the reference's behaviour)
    outs2, sts2 = _decode(bad, z.Context(0))
    assert sts2 == [0] * 4


@pytest.mark.parametrize("seq_variant", [0, 1, 3])
def test_escape_codes_long_literal_runs_and_long_matches(oracle, seq_variant):
    """k_seq_pipe keeps literal-length codes >= 32 (runs >= 8192 bytes) and match-length codes >= 45
    (>= 1027 bytes) out of its 2-byte LDS cell ("escape": next = 0) and serves them in the general
    step from the host cell.  Frames built to hit exactly those: incompressible stretches of 8 KiB
    to 70 KiB between repeated text (long literal runs), and long repeats (long matches), several
    of each per block so that escapes meet ordinary sequences in one wavefront."""
    from tools import synth_binding as sb
    rng = np.random.default_rng(7)
    frames, want = [], []
    for i in range(24):
        text = sb.generate(sb.TEXT, 300 + i, 40000)
        parts = []
        for j in range(3):
            noise = sb.generate(sb.RANDOM, 1000 * i + j, int(rng.integers(8192, 70000)))
            rep_len = int(rng.integers(1027, 30000))
            parts += [text[:int(rng.integers(2000, 12000))], noise, text[:rep_len], text[:rep_len]]
        data = b"".join(parts)[:3 * 131072]
        f, nseq = sb.compress(data)
        rc, ref, _, tr = oracle.decode_frame(f, cap=len(data) + 64, want_trace=True)
        assert rc == 0 and ref == data
        lls = [s[0] for s in tr["seqs"]]
        mls = [s[1] for s in tr["seqs"]]
        assert max(lls) >= 8192 and max(mls) >= 1027, (max(lls), max(mls))  # the generator really produced escapes
        frames.append(f)
        want.append(data)
    outs, sts = _decode(frames, z.Context(0, seq_variant=seq_variant))
    assert sts == [0] * len(frames)
    assert outs == want


# ---- planning on the device: k_parse instead of the host planner

def _both_plans(frames, ctx):
    outs_h, sts_h = z.decode_frames(frames, ctx)
    outs_d, sts_d = z.decode_frames(frames, ctx, device_plan=True)
    return outs_h, sts_h, outs_d, sts_d


def test_device_planner_corpus_bit_exact(corpus, ctx):
    """The whole golden corpus, headers parsed on the device: same bytes as the corpus says."""
    outs, sts = z.decode_frames([comp for _, comp, *_ in corpus], ctx, device_plan=True)
    assert all(s == 0 for s in sts), [(corpus[i][0], s) for i, s in enumerate(sts) if s]
    for (name, comp, length, sha, exp), got in zip(corpus, outs):
        check_expected(name, got, length, sha, exp)


def test_device_planner_equals_host_planner_layout_and_stats(corpus, ctx):
    """Same slabs, same work: the device-planned batch has the host-planned batch's output layout and
    byte / sequence / stream counts (the roofline accounting does not depend on who planned)."""
    from tools import synth_binding as sb
    blob, off, ln, _, _ = sb.make_batch(4, 5, 64, threads=4)
    frames = [comp for _, comp, *_ in corpus] + [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
    p = z.Plan(device_tables=True)
    for f in frames:
        assert p.add_frame(f)[0] == 0
    b = p.finalize()
    rh = ctx.upload(b)
    lens = np.array([len(f) for f in frames], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    rd = ctx.upload_frames(b"".join(frames), offs, lens)
    try:
        assert rd.out_size == rh.out_size
        lo, lc = rd.frame_layout()
        assert [int(x) for x in lo] == [int(b.frames[i].out_offset) for i in range(b.n_frames)]
        assert [int(x) for x in lc] == [int(b.frames[i].out_capacity) for i in range(b.n_frames)]
        sh, sd = rh.stats(), rd.stats()
        for k in ("compressed_bytes", "scratch_bytes", "out_capacity_bytes", "n_sequences", "n_huf_streams"):
            assert getattr(sh, k) == getattr(sd, k), k
        assert list(sh.n_blocks) == list(sd.n_blocks)
        assert sd.n_huf_built == sh.n_huf_built and sd.n_fse_built >= sh.n_fse_built
        assert sd.parse_ms > 0
        rd.run()
        rh.run()
        od, std, ld = rd.download()
        oh, sth, lh = rh.download()
        assert (std == 0).all() and (sth == 0).all() and (ld == lh).all()
        for i in range(len(frames)):
            o, n = int(lo[i]), int(ld[i])
            assert (od[o:o + n] == oh[o:o + n]).all(), i
    finally:
        rd.free()
        rh.free()
        p.close()


def test_device_planner_synthetic_and_multiblock(ctx, oracle):
    from tools import synth_binding as sb
    frames = []
    for cfg, seed, n in [(1, 3, 24), (2, 4, 24), (3, 5, 16), (4, 6, 48)]:
        blob, off, ln, _, _ = sb.make_batch(cfg, seed, n, threads=4)
        frames += [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
    sb.set_content_checksum(True)
    try:
        blob, off, ln, _, _ = sb.make_batch(4, 9, 16, threads=2)
        frames += [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
    finally:
        sb.set_content_checksum(False)
    outs_h, sts_h, outs_d, sts_d = _both_plans(frames, ctx)
    assert sts_d == sts_h == [0] * len(frames)
    assert outs_d == outs_h
    for f, o in list(zip(frames, outs_d))[::7]:
        rc, want, _, _ = oracle.decode_frame(f, cap=8 << 20)
        assert rc == 0 and o == want


def test_device_planner_checksum_verification(corpus):
    """Checksums captured by the device parser feed k_xxh64 like the host planner's."""
    c = z.Context(0, verify_checksum=True)
    frames = [comp for _, comp, *_ in corpus]
    bad = bytearray(frames[5])
    bad[-1] ^= 0x40  # the last byte of a corpus frame is its checksum
    frames.append(bytes(bad))
    outs_h, sts_h, outs_d, sts_d = _both_plans(frames, c)
    assert sts_d == sts_h
    assert sts_d[-1] == 18 and all(s == 0 for s in sts_d[:-1])
    c.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_device_planner_fuzz_agrees_with_host_planner(corpus, seed):
    """Mutated and truncated frames (headers included): the device parser reports, frame for frame, the
    status the host planner + upload report, and decodes the same bytes when both succeed."""
    rng = np.random.default_rng(977 + seed)
    frames = []
    for _, comp, *_ in corpus:
        for k in range(8):
            b = bytearray(comp)
            if k == 6:
                b = b[:int(rng.integers(0, len(b)))]  # truncated anywhere, header included
            elif k == 7:
                b[int(rng.integers(0, min(len(b), 12)))] ^= int(rng.integers(1, 256))  # header damage
            else:
                for pos in rng.integers(4, len(b), size=1 + k % 3):
                    b[int(pos)] ^= int(rng.integers(1, 256))
            frames.append(bytes(b))
    c = z.Context(0)
    outs_h, sts_h, outs_d, sts_d = _both_plans(frames, c)
    diff = [(i, sts_h[i], sts_d[i]) for i in range(len(frames)) if sts_h[i] != sts_d[i]]
    assert not diff, diff[:20]
    assert outs_h == outs_d
    assert 0 < sum(1 for s in sts_d if s == 0) < len(frames)
    c.close()


def _plan_by_blocks(ctx, on=True):
    """frames of 16 bytes and more are planned block by block (mzd_debug_plan_unit_bytes) / the library's own threshold again"""
    from sparkzstd_amd import _lib
    assert _lib.load().mzd_debug_plan_unit_bytes(ctx._c, 16 if on else 0) == 0


def test_device_planner_block_by_block_corpus_and_fuzz(corpus, oracle):
    """A LARGE frame is planned on the device block by block: one lane walks its 3-byte block headers (k_parse_index), then a
    lane per block parses literal / sequence section headers, Huffman weights and FSE descriptions side by side, and what a
    block inherits -- the table a Treeless literals section or a Repeat_Mode sequence table reuses (literals.go:247-252,
    sequences.go:275-366, framedecompressor.go:283-294), whether an earlier block had sequences -- is resolved between the two
    passes from what every block says it needs and leaves.  With the threshold forced down to 16 bytes the whole corpus and
    its mutations go that way: same bytes as the manifest, and frame for frame the statuses and bytes of the host planner."""
    c = z.Context(0)
    try:
        _plan_by_blocks(c)
        outs, sts = z.decode_frames([comp for _, comp, *_ in corpus], c, device_plan=True)
        assert all(s == 0 for s in sts), [(corpus[i][0], s) for i, s in enumerate(sts) if s]
        for (name, comp, length, sha, exp), got in zip(corpus, outs):
            check_expected(name, got, length, sha, exp)
        rng = np.random.default_rng(4242)
        frames = []
        for _, comp, *_ in corpus:
            for k in range(8):
                b = bytearray(comp)
                if k == 6:
                    b = b[:int(rng.integers(0, len(b)))]  # truncated anywhere, header included
                elif k == 7:
                    b[int(rng.integers(0, min(len(b), 12)))] ^= int(rng.integers(1, 256))  # header damage
                else:
                    for pos in rng.integers(4, len(b), size=1 + k % 3):
                        b[int(pos)] ^= int(rng.integers(1, 256))
                frames.append(bytes(b))
        outs_d, sts_d = z.decode_frames(frames, c, device_plan=True)
        _plan_by_blocks(c, False)
        outs_h, sts_h = z.decode_frames(frames, c)
        diff = [(i, sts_h[i], sts_d[i]) for i in range(len(frames)) if sts_h[i] != sts_d[i]]
        assert not diff, diff[:20]
        assert outs_h == outs_d
        assert 0 < sum(1 for s in sts_d if s == 0) < len(frames)
        # a frame whose blocks end without a last block, cut exactly behind a block: the unit that runs to the frame's end reports it
        name, comp, *_ = max(corpus, key=lambda it: it[2])
        rc, _, _, tr = oracle.decode_frame(comp, cap=2 << 20, want_trace=True)
        assert rc == 0 and len(tr["blocks"]) > 2
        _plan_by_blocks(c)
        for cut in range(len(comp) - 40, len(comp) - 4):
            (_,), (sd,) = z.decode_frames([comp[:cut]], c, device_plan=True)
            (_,), (sh,) = z.decode_frames([comp[:cut]], c)
            assert sd == sh != 0, (cut, sd, sh)
        # a table with a symbol beyond its kind's codes (tests/golden/fuzz_ml_symbol_53.zst, the soak of round 6): MZD_ERR_UNSUPPORTED from both
        # planners, the device's block by block and as one lane, like the host's (tests/test_planner.py)
        ml53 = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_ml_symbol_53.zst"), "rb").read()
        for on in (True, False):
            _plan_by_blocks(c, on)
            assert z.decode_frames([ml53], c, device_plan=True)[1] == [16] and z.decode_frames([ml53], c)[1] == [16]
        _plan_by_blocks(c)
        # ... and cut INSIDE its content checksum (fewer than four bytes behind the last block): the frame's last unit must say
        # whether the four bytes were there -- with verification on, both planners leave the flag clear, the frame decodes, and
        # nobody reports the failure of a checksum that is not there (the block-by-block route used to)
        cv = z.Context(0, verify_checksum=True)
        try:
            for cut in range(len(comp) - 3, len(comp) + 1):
                _plan_by_blocks(cv)
                (od,), (sd,) = z.decode_frames([comp[:cut]], cv, device_plan=True)
                _plan_by_blocks(cv, False)
                (oh,), (sh,) = z.decode_frames([comp[:cut]], cv)
                (ow,), (sw,) = z.decode_frames([comp[:cut]], cv, device_plan=True)  # (the single-lane walk: the library's own threshold)
                assert sd == sh == sw == 0 and od == oh == ow and len(od) == max(it[2] for it in corpus), (cut, sd, sh, sw)
        finally:
            cv.close()
    finally:
        c.close()


def test_device_planner_large_frames_stay_on_the_device(oracle):
    """Frames of 9 and 20 MiB (72 and 160 blocks; the library's own threshold) beside small ones, the compressed blob resident
    on the device: mzd_batch_upload_frames plans them there -- no copy back to the host planner -- and the frames decode to the
    oracle's bytes; the layout is the host planner's."""
    import torch
    from tools import synth_binding as sb
    datas = [sb.generate(sb.TEXT, 71, (9 << 20) + 12345), sb.generate(sb.EXP, 72, 20 << 20), sb.generate(sb.TEXT, 73, 70000), b""]
    frames = [sb.compress(d, sb.MODE_FULL)[0] for d in datas]
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    lens = np.array([len(f) for f in frames], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    c = z.Context(0)
    d_in = torch.zeros(blob.size + 128, dtype=torch.uint8, device="cuda")
    d_in[64:64 + blob.size].copy_(torch.from_numpy(blob))
    rb = c.upload_frames(int(blob.size), offs, lens, device_in_ptr=d_in.data_ptr() + 64)
    try:
        p = z.Plan(device_tables=True)
        for f in frames:
            assert p.add_frame(f)[0] == 0
        b = p.finalize()
        lo, lc = rb.frame_layout()
        assert [int(x) for x in lo] == [int(b.frames[i].out_offset) for i in range(b.n_frames)]
        assert [int(x) for x in lc] == [int(b.frames[i].out_capacity) for i in range(b.n_frames)]
        assert rb.out_size == b.out_size
        rb.run()
        out, st, ln = rb.download()
        assert (st == 0).all() and [int(x) for x in ln] == [len(d) for d in datas]
        for i, d in enumerate(datas):
            assert out[int(lo[i]):int(lo[i]) + len(d)].tobytes() == d, i
        p.close()
    finally:
        rb.free()
        c.close()


def test_trimmed_batch_keeps_its_output_and_nothing_else(ctx):
    """mzd_batch_trim (ABI 7; what a reader calls once the statuses are down): the resident output can still be read -- in pieces
    (mzd_batch_read_out), whole (mzd_batch_download), its layout asked for -- the batch cannot be run again, its scratch is gone
    (mzd_batch_debug_read refuses), and mzd_batch_last_pass still says what the run took."""
    from sparkzstd_amd import _lib
    from sparkzstd_amd.api import MzdError
    from tools import synth_binding as sb
    datas = [sb.generate(sb.TEXT, 40 + i, 50000 + 7000 * i) for i in range(6)]
    frames = [sb.compress(d, sb.MODE_FULL)[0] for d in datas]
    p = z.Plan(device_tables=True)
    for f in frames:
        assert p.add_frame(f)[0] == 0
    b = p.finalize()
    rb = ctx.upload(b)
    try:
        rb.run()
        flags = rb.last_pass()  # (the `ctx` fixture forces each execution kernel in turn)
        _, st, ln = rb.download(want_out=False)
        assert (st == 0).all()
        rb.trim()
        lo, _ = rb.frame_layout()
        for i, d in enumerate(datas):
            buf = np.empty(len(d), dtype=np.uint8)
            rb.read_out(int(lo[i]), buf.ctypes.data, len(d))
            assert buf.tobytes() == d
        out, st2, ln2 = rb.download()
        assert (st2 == 0).all() and (ln2 == ln).all() and out[int(lo[3]):int(lo[3]) + len(datas[3])].tobytes() == datas[3]
        assert rb.last_pass() == flags
        with pytest.raises(MzdError):
            rb.run()
        with pytest.raises(MzdError):
            rb.debug_read(_lib.MZD_DEBUG_RECORDS, np.uint64, 0, 1)
        rb.trim()  # (twice: nothing left to free)
    finally:
        rb.free()
        p.close()


def test_device_planner_edge_batches(ctx):
    """Empty batch, empty frame, frame range outside the blob."""
    rb = ctx.upload_frames(b"", [], [])
    rb.run()
    out, st, ln = rb.download()
    assert len(st) == 0
    rb.free()
    one = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00])  # single segment, size 0, empty raw last block
    rb = ctx.upload_frames(one + one, [0, 9, 4, 1000], [9, 9, 0, 9])
    rb.run()
    out, st, ln = rb.download()
    assert [int(x) for x in st] == [0, 0, 1, 1] and [int(x) for x in ln[:2]] == [0, 0]
    rb.free()


# ---- streaming: batches pipelined through recycled device slots

@pytest.mark.parametrize("depth,pinned", [(2, True), (3, False), (1, True)])
def test_stream_of_batches_equals_one_shot_decode(corpus, ctx, oracle, depth, pinned):
    """Seven batches of different sizes and content (corpus slices, synthetic frames, a batch with damaged
    frames, an empty batch) through a `depth`-slot stream: every frame's status and bytes equal the
    one-shot decode; slots are recycled (batch sizes go up and down)."""
    from tools import synth_binding as sb
    frames_all = [comp for _, comp, *_ in corpus]
    blob4, off4, ln4, _, _ = sb.make_batch(4, 21, 40, threads=4)
    syn = [bytes(blob4[o:o + l]) for o, l in zip(off4, ln4)]
    bad = [bytes(bytearray(f[:len(f) // 2])) for f in frames_all[10:14]] + [b"\x28\xb5\x2f\xfd"]
    batches = [frames_all[:30], syn[:25], frames_all[30:100] + syn[25:], [], bad + frames_all[5:9], syn[:3], frames_all[50:60]]
    want = [z.decode_frames(b, ctx) if b else ([], []) for b in batches]
    # what the stream must deliver is pinned independently of the product: corpus frames by the golden manifest
    # (length, sha256, verbatim bytes), synthetic frames by the oracle
    golden = {comp: (name, length, sha, exp) for name, comp, length, sha, exp in corpus}
    st = z.Stream(ctx, depth=depth)
    bufs, inflight, results = [], [], {}
    for bi, b in enumerate(batches):
        if len(inflight) == depth:
            k, t, out = inflight.pop(0)
            results[k] = (st.wait(t), out)
        ln = np.array([len(f) for f in b], dtype=np.uint64)
        off = np.concatenate([[0], np.cumsum(ln)[:-1]]).astype(np.uint64) if len(b) else np.zeros(0, dtype=np.uint64)
        raw = np.frombuffer(b"".join(b), dtype=np.uint8)
        cap = z.split_frames(raw)[4] if all(sx != 1 and sx != 2 for sx in want[bi][1]) else 64 << 20
        if pinned:
            pin, pout = z.PinnedBuffer(raw.size), z.PinnedBuffer(cap)
            pin.a[:] = raw
            bufs += [pin, pout]
            src, out = pin.a, pout.a
        else:
            src, out = raw.copy(), np.zeros(cap, dtype=np.uint8)
        inflight.append((bi, st.submit(src, off, ln, out), out))
    for k, t, out in inflight:
        results[k] = (st.wait(t), out)
    for bi, b in enumerate(batches):
        (status, out_len, out_off), out = results[bi]
        outs_w, sts_w = want[bi]
        assert [int(x) for x in status] == sts_w, bi
        for i in range(len(b)):
            if sts_w[i] == 0:
                o, n = int(out_off[i]), int(out_len[i])
                got = out[o:o + n].tobytes()
                assert got == outs_w[i], (bi, i)
                if b[i] in golden:
                    name, length, sha, exp = golden[b[i]]
                    check_expected(name, got, length, sha, exp)
                else:
                    rc, ref, _, _ = oracle.decode_frame(b[i], cap=n + 64)
                    assert rc == 0 and got == ref, (bi, i)
            else:
                assert oracle.decode_frame(b[i], cap=4 << 20)[0] != 0, (bi, i)  # the oracle rejects it too
    st.close()


def test_stream_refuses_overcommit_and_small_output(corpus, ctx):
    f = corpus[3][1]
    blob = np.frombuffer(f, dtype=np.uint8).copy()
    off, ln = np.array([0], dtype=np.uint64), np.array([len(f)], dtype=np.uint64)
    st = z.Stream(ctx, depth=1)
    out = np.zeros(z.split_frames(f)[4], dtype=np.uint8)  # exactly what the frame's bound needs
    t = st.submit(blob, off, ln, out)
    with pytest.raises(z.MzdError) as e:
        st.submit(blob, off, ln, out)  # the only slot holds an uncollected ticket
    assert e.value.code == 101
    status, out_len, out_off = st.wait(t)
    assert int(status[0]) == 0
    with pytest.raises(z.MzdError) as e:
        st.submit(blob, off, ln, np.zeros(16, dtype=np.uint8))
    assert e.value.code == 15
    t = st.submit(blob, off, ln, out)  # the slot is usable again
    assert int(st.wait(t)[0][0]) == 0
    st.close()


@pytest.mark.parametrize("window_kib,device_plan", [(64, False), (64, True), (300, True), (1, False)])
def test_input_blob_decoded_window_by_window(corpus, ctx, oracle, window_kib, device_plan):
    """k_seq_q4 addresses bitstreams with 32-bit offsets from a window of the blob; blobs of 4 GiB and more
    are decoded window by window.  With the window shrunk to a few KiB the corpus + synthetic batch takes many
    launches, and a frame wider than a window is decoded with a window per WORKGROUP (round 6: from the workgroup's first chain on;
    it used to fall back to the two-wavefront kernel, which the release library no longer has): same bytes either way."""
    from tools import synth_binding as sb
    blob, off, ln, _, _ = sb.make_batch(4, 31, 24, threads=4)
    frames = [comp for _, comp, *_ in corpus] + [bytes(blob[o:o + l]) for o, l in zip(off, ln)]
    want, sts_w = z.decode_frames(frames, ctx)
    c = z.Context(0, seq_window_kib=window_kib, assume_cus=3)
    outs, sts = z.decode_frames(frames, c, device_plan=device_plan)
    assert sts == sts_w == [0] * len(frames)
    assert outs == want
    for (name, comp, length, sha, exp), got in zip(corpus, outs):  # the golden manifest, not only the product itself
        check_expected(name, got, length, sha, exp)
    for f, got in zip(frames[len(corpus):], outs[len(corpus):]):
        rc, ref, _, _ = oracle.decode_frame(f, cap=len(got) + 64)
        assert rc == 0 and got == ref
    c.close()
