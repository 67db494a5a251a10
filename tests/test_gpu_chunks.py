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
            if used == 0 and made == 0:
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
