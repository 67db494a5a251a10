"""-m gpu: ONE frame through the device in chunks of whole blocks (mzd_fstream_*, ABI 9) -- the reference's shape for a frame
(framedecompressor.go:198-303 DecodeNextBlock into a ring of the window's size, ringbuffer.go:36-49; framereader.go:51-109 hands
the bytes on as they come) -- against the golden corpus, the generator's content and the whole-frame path, bit for bit."""
import hashlib

import numpy as np
import pytest

import sparkzstd_amd as z
from tests.conftest import check_expected

pytestmark = pytest.mark.gpu

KIB, MIB = 1 << 10, 1 << 20


def stream_decode(comp: bytes, ctx, chunk_bytes: int, piece: int = 1 << 40, dst_bytes: int = 0, on_chunk=None):
    """-> (the frame's bytes, chunks, source bytes consumed, FrameStream's window)"""
    fs = z.FrameStream(ctx, chunk_bytes)
    src = np.frombuffer(comp, dtype=np.uint8)
    dst = np.empty(max(dst_bytes or chunk_bytes, 128 * KIB), dtype=np.uint8)
    out = bytearray()
    pos = chunks = 0
    have = min(len(comp), piece)
    try:
        while not fs.done:
            used, made = fs.next(src[pos:have], dst)
            if used == 0 and made == 0 and not fs.done:  # (done with nothing made: the last chunk of a frame that ends in an empty block)
                assert have < len(comp), "the stream wants bytes behind the frame's end"
                have = min(len(comp), have + piece)
                continue
            pos += used
            chunks += 1 if made or fs.done else 0
            out += dst[:made].tobytes()
            if on_chunk:
                on_chunk(len(out))
            if pos >= have:
                have = min(len(comp), have + piece)
        assert fs.total_out == len(out)
        return bytes(out), chunks, pos, fs.window
    finally:
        fs.close()


def with_window(frame: bytes, window_log: int) -> bytes:
    """A single-segment frame (the synthetic encoder's: the window is the content) re-headed with a Window_Descriptor of
    2^window_log bytes (frame.go:28-36): the same blocks, a bounded window."""
    assert frame[:4] == b"\x28\xb5\x2f\xfd"
    fhd = frame[4]
    assert fhd & 0x20 and (fhd >> 6) >= 1, "single segment with a content size of 2 bytes or more"
    return frame[:4] + bytes([fhd & ~0x20 & 0xFF, (window_log - 10) << 3]) + frame[5:]


@pytest.mark.parametrize("chunk_bytes,piece", [(128 * KIB, 100000), (MIB, 1 << 40)])
def test_corpus_frames_in_chunks(corpus, chunk_bytes, piece):
    """Every frame of the reference's corpus, a block (or eight) per chunk, the source arriving 100 kB at a time: Repeat / Treeless
    tables, repeat offsets and matches across chunk boundaries; Raw / RLE and literal-only blocks as chunks of their own."""
    ctx = z.Context(0)
    many = 0
    for name, comp, length, sha, exp in corpus:
        got, chunks, used, w = stream_decode(comp, ctx, chunk_bytes, piece)
        check_expected(name, got, length, sha, exp)
        assert used == len(comp) - 4, name  # everything but the content checksum (framereader.go:84-94)
        many += chunks > 1
    assert many >= (40 if chunk_bytes == 128 * KIB else 5), many
    ctx.close()


@pytest.mark.parametrize("chunk_mib,window_log", [(4, 23), (16, 23), (4, 26), (1, 23)])
def test_large_frame_in_chunks_with_a_bounded_window(chunk_mib, window_log):
    """A text-like frame of 40 MiB whose matches reach back up to 8 MiB, declared with a window of 8 MiB (or 64 MiB: the frame so far
    stays in the slab whole), in chunks of 1-16 MiB: every chunk but the first starts behind the window (block mode: its first job
    reads those bytes, the jobs behind it are fixed up from them), the history and the tables carry over."""
    from tools import synth_binding as sb
    n = (40 << 20) + 4321
    sb.set_max_offset(1 << 23)
    try:
        d = sb.generate(sb.TEXT, 4242, n)
        comp = with_window(sb.compress(d, sb.MODE_FULL)[0], window_log)
    finally:
        sb.set_max_offset(0)
    ctx = z.Context(0)
    got, chunks, used, w = stream_decode(comp, ctx, chunk_mib * MIB, piece=3 * MIB + 17)
    assert w == 1 << window_log and chunks >= n // (chunk_mib * MIB)
    assert len(got) == n
    if got != d:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(d, np.uint8)
        bad = np.nonzero(a != b)[0]
        raise AssertionError((len(bad), bad[:8].tolist(), chunks))
    # the whole-frame path on the same bytes
    outs, sts = z.decode_frames([comp], ctx)
    assert sts == [0] and outs[0] == d
    ctx.close()


def test_matches_that_reach_back_more_than_8_mib_across_chunks():
    """The frame of test_block_mode_matches_that_reach_back_more_than_8_mib (an incompressible head matched again 10 MiB and 22 MiB
    later) with a window of 32 MiB, in chunks of 4 MiB: the slab holds up to 36 MiB, the passes spell origins with four planes, and
    the bytes the late matches want lie in the window the chunks before left."""
    from tools import synth_binding as sb
    A = sb.generate(sb.RANDOM, 99, 200000)
    T = sb.generate(sb.TEXT, 98, 3 << 20)
    far = A + bytes(10 << 20) + A[:150000] + T + bytes(9 << 20) + A[50000:] + T[:1 << 20]
    comp = with_window(sb.compress(far, sb.MODE_FULL)[0], 25)
    ctx = z.Context(0)
    for chunk in (4 * MIB, 13 * MIB):
        got, chunks, used, w = stream_decode(comp, ctx, chunk)
        assert w == 1 << 25 and chunks >= len(far) // chunk and got == far, chunk
    # declared with a window of 8 MiB the frame is not valid zstd: the late matches reach behind what a decoder has to keep.  The
    # reference's ring wraps and hands out newer bytes (ringbuffer.go:198-225); the chunks report the offset
    bad = with_window(sb.compress(far, sb.MODE_FULL)[0], 23)
    with pytest.raises(z.MzdError) as e:
        stream_decode(bad, ctx, 4 * MIB)
    assert e.value.code == 14  # MZD_ERR_OFFSET
    ctx.close()


@pytest.mark.parametrize("exec_variant", [4, 5])
def test_chunks_with_the_execution_kernel_forced(corpus, exec_variant):
    """k_exec_c on every chunk (5), block mode with jobs of four blocks on every chunk (4) -- the corpus in chunks of 1 MiB and a
    text-like frame of 6 MiB in chunks of 1 MiB -- and the variants that do not know chunks refuse them."""
    from tools import synth_binding as sb
    ctx = z.Context(0, exec_variant=exec_variant)
    for name, comp, length, sha, exp in corpus[::3]:
        got, chunks, used, w = stream_decode(comp, ctx, MIB)
        check_expected(name, got, length, sha, exp)
    d = sb.generate(sb.TEXT, 77, 6 * MIB + 99)
    comp = sb.compress(d, sb.MODE_FULL)[0]
    got, chunks, used, w = stream_decode(comp, ctx, MIB, piece=300000)
    assert got == d and chunks >= 6 and w == len(d)  # (single segment: the window is the content)
    ctx.close()
    ctx = z.Context(0, exec_variant=1, library="release")
    with pytest.raises(z.MzdError) as e:
        stream_decode(comp, ctx, MIB)
    assert e.value.code == 16  # MZD_ERR_UNSUPPORTED
    ctx.close()


def test_chunk_errors_are_the_frames_errors_and_stick(corpus):
    """A frame damaged in its fifth block: the chunks before it come out, the chunk that holds it fails with the status the whole
    frame gets, and the stream stays failed.  A source that ends inside a block: nothing is consumed, nothing produced.  A header
    that declares another content size than the blocks make: MZD_ERR_DST_FULL at the last block, as for a whole frame."""
    from tools import synth_binding as sb
    from tests.frame_splice import frame_blocks, splice_frame
    ctx = z.Context(0)
    d = sb.generate(sb.TEXT, 5, 9 * 131072)
    comp = sb.compress(d, sb.MODE_FULL)[0]
    blocks = frame_blocks(comp)
    assert len(blocks) >= 8
    typ, payload, size = blocks[4]
    hurt = bytearray(payload)
    hurt[-3] ^= 0x5A
    hurt[-2] ^= 0xA5  # (the head of the sequence bitstream: its initial states)
    blocks[4] = (typ, bytes(hurt), size)
    bad = splice_frame(blocks)
    outs, sts = z.decode_frames([bad], ctx)
    assert sts[0] != 0
    fs = z.FrameStream(ctx, 128 * KIB)
    src, dst = np.frombuffer(bad, dtype=np.uint8), np.empty(128 * KIB, dtype=np.uint8)
    pos, out = 0, bytearray()
    with pytest.raises(z.MzdError) as e:
        while not fs.done:
            used, made = fs.next(src[pos:], dst)
            pos += used
            out += dst[:made].tobytes()
    assert e.value.code == sts[0] and bytes(out) == d[:len(out)] and len(out) == 4 * 131072
    with pytest.raises(z.MzdError) as e2:
        fs.next(src[pos:], dst)
    assert e2.value.code == sts[0]
    fs.close()
    # a source that ends inside the last block
    cut = len(comp) - len(blocks[-1][1]) // 2
    fs = z.FrameStream(ctx, 128 * KIB)
    src = np.frombuffer(comp[:cut], dtype=np.uint8)
    pos = n = 0
    while True:
        used, made = fs.next(src[pos:], dst)
        if used == 0 and made == 0:
            break
        pos += used
        n += made
    assert not fs.done and n == (len(blocks) - 1) * 131072 and pos < cut
    fs.close()
    with pytest.raises(z.MzdError) as e3:  # a destination below one block
        z.FrameStream(ctx, 128 * KIB).next(src, np.empty(1000, dtype=np.uint8))
    assert e3.value.code == 101
    # the declared content size (4 bytes here) one byte off
    assert comp[4] >> 6 == 2 and comp[4] & 0x20
    lied = comp[:5] + (int.from_bytes(comp[5:9], "little") + 1).to_bytes(4, "little") + comp[9:]
    assert z.decode_frames([lied], ctx)[1] == [15]
    with pytest.raises(z.MzdError) as e4:
        stream_decode(lied, ctx, 256 * KIB)
    assert e4.value.code == 15  # MZD_ERR_DST_FULL
    ctx.close()


def test_a_frame_larger_than_the_memory_it_is_given():
    """A frame of 256 MiB with a window of 8 MiB in chunks of 4 MiB: the device holds two slabs of window + chunk and the scratch of
    two chunks (the one that runs, the one being copied out) -- the same from the fourth chunk to the last, whatever the frame's
    length, and less than the frame (the whole-frame path holds the output and three planes of it, 1 GiB, and the scratch of all
    of it)."""
    import torch
    from tools import synth_binding as sb
    n = 256 * MIB
    sb.set_max_offset(1 << 23)
    try:
        blob, off, ln, ck, ns = sb.make_batch(4, 91, 1, frame_bytes=n, threads=8)
    finally:
        sb.set_max_offset(0)
    comp = with_window(blob[int(off[0]):int(off[0] + ln[0])].tobytes(), 23)
    ctx = z.Context(0)

    def used():
        free, total = torch.cuda.mem_get_info(0)
        return total - free

    base = used()
    seen = []
    h = hashlib.sha256()
    fs = z.FrameStream(ctx, 4 * MIB)
    src, dst = np.frombuffer(comp, dtype=np.uint8), np.empty(4 * MIB, dtype=np.uint8)
    pos = total = 0
    while not fs.done:
        u, m = fs.next(src[pos:pos + 6 * MIB], dst)
        assert u or m or fs.done
        pos += u
        total += m
        h.update(dst[:m].tobytes())
        seen.append(used() - base)
    fs.close()
    assert total == n and len(seen) >= 48
    print("device memory while the frame goes through: %.0f MiB at most" % (max(seen) / MIB))
    assert max(seen) < n and max(seen) <= max(seen[:6]) + 8 * MIB, (max(seen), seen[:8])
    outs, sts = z.decode_frames([comp], ctx)
    assert sts == [0] and hashlib.sha256(outs[0]).hexdigest() == h.hexdigest() and sb.checksum64(outs[0]) == int(ck[0])
    ctx.close()


class _Dribble:
    """a source that is not seekable and hands out at most `k` bytes per read (a socket, a pipe)"""

    def __init__(self, data, k):
        self._d, self._p, self._k = data, 0, k

    def read(self, n=-1):
        n = self._k if n is None or n < 0 else min(n, self._k)
        d = self._d[self._p:self._p + n]
        self._p += len(d)
        return d


def test_readers_in_chunk_mode(corpus):
    """FrameReader / FrameDecompressor with chunk_bytes (the reference's shape: framereader.go:51-109 over DecodeNextBlock): whole
    reads, small Reads across chunk boundaries, readinto, a source that dribbles; a cut source ends in io.ErrUnexpectedEOF."""
    import io
    from tools import synth_binding as sb
    from sparkzstd_amd.decompression import ZstdError
    ctx = z.Context(0)
    d = sb.generate(sb.TEXT, 321, 5 * MIB + 1234)
    comp = sb.compress(d, sb.MODE_FULL)[0]
    assert z.FrameReader(io.BytesIO(comp), ctx, chunk_bytes=MIB).read() == d
    r = z.FrameReader(_Dribble(comp, 7777), ctx, chunk_bytes=512 * KIB)
    got = bytearray()
    while True:
        part = r.Read(100003)
        if not part:
            break
        assert len(part) <= 100003
        got += part
    assert bytes(got) == d and r.Read(10) == b""
    r.Reset(io.BytesIO(comp))  # (the same reader, the next frame)
    buf = bytearray(300000)
    got = bytearray()
    while True:
        k = r.readinto(buf)
        if not k:
            break
        got += buf[:k]
    assert bytes(got) == d
    r.close()
    for name, comp2, length, sha, exp in corpus[:20]:
        check_expected(name, z.FrameReader(_Dribble(comp2, 4096), ctx, chunk_bytes=128 * KIB).read(), length, sha, exp)
    # FrameDecompressor: DecodeNextBlock = the next chunk
    sink = io.BytesIO()
    fd = z.FrameDecompressor(io.BytesIO(comp), sink, ctx, chunk_bytes=MIB)
    fd.CheckMagicnum()
    n = 0
    while True:
        try:
            fd.DecodeNextBlock()
        except ZstdError as e:
            assert e.code == 17  # ErrOutOfBlocks (framedecompressor.go:196)
            break
        n += 1
    assert sink.getvalue() == d and n == fd.BlockCounter and 5 <= n <= 7
    sink = io.BytesIO()
    z.FrameDecompressor(io.BytesIO(comp), sink, ctx, chunk_bytes=2 * MIB).Decompress()
    assert sink.getvalue() == d
    with pytest.raises(ZstdError) as e:
        z.FrameReader(io.BytesIO(comp[:len(comp) // 2]), ctx, chunk_bytes=MIB).read()
    assert e.value.code == 1
    ctx.close()


def test_repeat_offsets_from_the_chunk_before_reach_back_more_than_8_mib(oracle):
    """Blocks that spell out no offset of their own: each copies 128 KiB from 12 MiB back through the REPEAT offset the block before
    left (hand-made blocks of one sequence under the predefined tables).  A chunk made of such blocks inherits that offset with its
    history -- block mode must take the pass for the position's high bits although none of the chunk's offset codes says so (found
    with a 2.5 GiB frame in round 6: k_blk_scan now counts the history it is handed)."""
    from tests.frame_splice import one_sequence_block, predefined_states, splice_frame
    st = predefined_states(oracle)
    rng = np.random.default_rng(12)
    R = bytes(rng.integers(0, 256, size=12 * MIB, dtype=np.uint8))
    blocks = [(0, R[i:i + 131072], 131072) for i in range(0, len(R), 131072)]
    blocks.append(one_sequence_block(st, b"a", 1, 131071, 12 * MIB + 3))
    for i in range(95):
        blocks.append(one_sequence_block(st, bytes([98 + i % 20]), 1, 131071, 1))
    frame = splice_frame(blocks)
    rc, want, *_ = oracle.decode_frame(frame, cap=32 * MIB)
    assert rc == 0 and len(want) == 24 * MIB
    for ev, chunk in ((0, 4 * MIB), (4, MIB), (5, 2 * MIB)):
        ctx = z.Context(0, exec_variant=ev)
        outs, sts = z.decode_frames([frame], ctx)
        assert sts == [0] and outs[0] == want, ev
        got, chunks, used, w = stream_decode(frame, ctx, chunk)
        assert chunks >= 24 * MIB // chunk and w == 128 * MIB
        if got != want:
            a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
            bad = np.nonzero(a[:min(len(a), len(b))] != b[:min(len(a), len(b))])[0]
            raise AssertionError((ev, chunk, len(got), len(bad), bad[:6].tolist()))
        ctx.close()


def test_zero_sequences_in_the_two_byte_form_on_the_device(oracle):
    """tests/test_planner.py::test_zero_sequences_in_the_two_byte_form on the device: the frame the chunk soak found and the hand-made
    ones, planned on the host and on the device, whole (every execution kernel) and in chunks -- the reference's verdict each time,
    reported in the sequence stage's place (after the block's literal errors, after the blocks before it)."""
    import os
    from tests.test_planner import _zero_sequence_frames
    f = open(os.path.join(os.path.dirname(__file__), "golden", "fuzz_zero_sequences_long_form.zst"), "rb").read()
    cases = [("the soak's frame", f, 11)] + _zero_sequence_frames()
    frames = [c[1] for c in cases]
    want = []
    for what, frame, rc_want in cases:
        rc, ref, *_ = oracle.decode_frame(frame, cap=1 << 20)
        assert rc == rc_want
        want.append(ref if rc == 0 else None)
    for ev in (0, 1, 2, 3, 4, 5):
        ctx = z.Context(0, exec_variant=ev)
        for device_plan in (False, True):
            outs, sts = z.decode_frames(frames, ctx, device_plan=device_plan)
            assert sts == [c[2] for c in cases], (ev, device_plan, sts)
            assert all(o == w for o, w, s in zip(outs, want, sts) if s == 0)
        ctx.close()
    ctx = z.Context(0)
    for (what, frame, rc_want), w in zip(cases, want):
        if rc_want == 0:
            assert stream_decode(frame, ctx, 128 * KIB)[0] == w, what
        else:
            with pytest.raises(z.MzdError) as e:
                stream_decode(frame, ctx, 128 * KIB)
            assert e.value.code == rc_want, what
    ctx.close()
