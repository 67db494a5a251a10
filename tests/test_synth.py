"""The synthetic workload generator (tools/synth: own zstd-format encoder) emits frames that the
reference algorithm (oracle) regenerates exactly, that the host planner accepts, and whose
descriptors interpret back to the original.  CPU only."""
import ctypes
import os

import numpy as np
import pytest

import sparkzstd_amd as z
from tests.desc_interp import run_batch
from tools import synth_binding as sb


def _libzstd():
    for p in ("/opt/conda/lib/libzstd.so.1", "libzstd.so.1"):
        try:
            L = ctypes.CDLL(p)
            L.ZSTD_decompress.restype = ctypes.c_size_t
            L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            L.ZSTD_isError.argtypes = [ctypes.c_size_t]
            return L
        except OSError:
            continue
    return None


CASES = [(sb.TEXT, sb.MODE_FULL, 131072), (sb.TEXT, sb.MODE_LITERALS, 20000), (sb.EXP, sb.MODE_LITERALS, 131072),
         (sb.EXP, sb.MODE_FULL, 50000), (sb.RANDOM, sb.MODE_FULL, 4096), (sb.ZERO, sb.MODE_FULL, 131072),
         (sb.TEXT, sb.MODE_FULL, 300000), (sb.TEXT, sb.MODE_FULL, 1 << 20), (sb.EXP, sb.MODE_FULL, 400000),
         (sb.TEXT, sb.MODE_FULL, 63), (sb.TEXT, sb.MODE_FULL, 0),
         (sb.RANDOM, sb.MODE_RAW, 1000), (sb.ZERO, sb.MODE_RLE, 1000)]


@pytest.mark.parametrize("kind,mode,n", CASES)
def test_synth_frames_decode_with_oracle_and_libzstd(oracle, kind, mode, n):
    data = sb.generate(kind, 1234 + n, n)
    frame, nseq = sb.compress(data, mode)
    rc, out, consumed, _ = oracle.decode_frame(frame, cap=n + 64)
    assert rc == 0 and out == data and consumed == len(frame)
    L = _libzstd()
    if L is not None:  # independent cross-check where libzstd exists (not required on the GPU box)
        dst = ctypes.create_string_buffer(n + 64)
        r = L.ZSTD_decompress(dst, n + 64, frame, len(frame))
        assert not L.ZSTD_isError(r) and dst.raw[:r] == data


def test_spliced_frames_decode_alike_with_oracle_and_libzstd(oracle):
    """The frames the GPU suite splices for the executor's small-block path (tests/frame_splice.py: blocks with sequences whose
    matches and offset history reach back over Raw / RLE / literals-only blocks of every size put between them): the oracle and
    libzstd regenerate the same bytes -- the oracle, which the device is checked against there, is pinned on them too."""
    from tests.frame_splice import frame_blocks, splice_frame, literal_block
    L = _libzstd()
    if L is None:
        pytest.skip("no libzstd on this box")
    rng = np.random.default_rng(3)
    sizes = [0, 1, 7, 8, 9, 63, 64, 511, 512, 513, 1024, 2047, 2048, 2049, 5000]
    src = frame_blocks(sb.compress(sb.generate(sb.TEXT, 4242, 3 * 131072 - 777), sb.MODE_FULL)[0])
    huf = [b for b in frame_blocks(sb.compress(sb.generate(sb.TEXT, 4243, 1500), sb.MODE_LITERALS)[0]) if b[0] == 2]
    assert len(src) >= 3 and huf
    blocks = []
    for j, b in enumerate(src):
        blocks.append(b)
        for n in sizes[j::3]:
            data = bytes(rng.integers(0, 256, size=max(n, 1), dtype=np.uint8))[:n]
            # (a compressed block of two bytes -- no literals, no sequences -- is below libzstd's minimum and rejected there; the reference
            # takes it, and so do the oracle and the device: the GPU suite has it, this comparison cannot)
            blocks += [(0, data, n), (1, b"\x7e", max(n, 1)), literal_block(data or b"x"), literal_block(data or b"x", rle=True), huf[0]]
    frame = splice_frame(blocks)
    rc, out, consumed, _ = oracle.decode_frame(frame, cap=8 << 20)
    assert rc == 0 and consumed == len(frame)
    dst = ctypes.create_string_buffer(8 << 20)
    r = L.ZSTD_decompress(dst, 8 << 20, frame, len(frame))
    assert not L.ZSTD_isError(r) and r == len(out) and dst.raw[:r] == out


def test_config_batches_have_the_surveyed_shape(oracle):
    """config 4 ~ 12.5k sequences / ~43 KB per frame (zstd -3 shape), config 3 has 0
    sequences and MaxBits 11, config 2 alternates raw / rle."""
    blob, off, ln, ck, ns = sb.make_batch(4, 0, 8, threads=2)
    assert 11000 < ns.mean() < 15000 and 40000 < ln.mean() < 50000
    for i in range(8):
        fr = blob[int(off[i]):int(off[i] + ln[i])].tobytes()
        rc, out, _, tr = oracle.decode_frame(fr, cap=131072 + 64, want_trace=True)
        assert rc == 0 and len(out) == 131072 and sb.checksum64(out) == int(ck[i])
        b = tr["blocks"][0]
        assert (b["lit_type"], b["lit_streams"], b["ll_mode"], b["of_mode"], b["ml_mode"]) == (2, 4, 2, 2, 2)
    blob, off, ln, ck, ns = sb.make_batch(3, 0, 4, threads=2)
    for i in range(4):
        fr = blob[int(off[i]):int(off[i] + ln[i])].tobytes()
        rc, out, _, tr = oracle.decode_frame(fr, cap=131072 + 64, want_trace=True)
        b = tr["blocks"][0]
        assert rc == 0 and b["n_seq"] == 0 and b["huf_max_bits"] == 11 and b["lit_regen"] == 131072
        assert sb.checksum64(out) == int(ck[i])
    blob, off, ln, ck, ns = sb.make_batch(2, 0, 4, threads=1)
    assert [int(x) for x in ln] == [131072 + 12, 13, 131072 + 12, 13]


def test_planner_and_descriptors_on_synth_frames():
    frames, datas = [], []
    for kind, mode, n in [(sb.TEXT, sb.MODE_FULL, 6000), (sb.EXP, sb.MODE_LITERALS, 3000), (sb.ZERO, sb.MODE_FULL, 5000),
                          (sb.TEXT, sb.MODE_FULL, 140000)]:
        d = sb.generate(kind, n, n)
        frames.append(sb.compress(d, mode)[0])
        datas.append(d)
    p = z.Plan()
    for f in frames:
        assert p.add_frame(f) == (0, len(f))
    b = p.finalize()
    blob = bytes((ctypes.c_uint8 * b.in_size).from_address(b.in_))
    assert run_batch(b, blob) == datas
    p.close()


def test_add_frames_adopts_blob_and_is_thread_safe():
    blob, off, ln, ck, ns = sb.make_batch(4, 100, 96, frame_bytes=8192, threads=4)
    p1, p2 = z.Plan(), z.Plan()
    assert p1.add_frames(blob, off, ln, threads=1) == 0
    assert p2.add_frames(blob, off, ln, threads=8) == 0
    b1, b2 = p1.finalize(), p2.finalize()
    assert (b1.n_frames, b1.n_blocks, b1.n_fse_entries, b1.n_huf_entries) == (b2.n_frames, b2.n_blocks, b2.n_fse_entries, b2.n_huf_entries)
    assert b1.in_ == blob.ctypes.data  # adopted, not copied
    for i in range(b1.n_blocks):
        assert bytes(b1.blocks[i]) == bytes(b2.blocks[i])
    p1.close(); p2.close()


def test_synth_content_checksum_is_xxh64_of_the_content(oracle):
    """Optional content checksum of the synthetic frames == oracle XXH64 (low 32 bits, little endian)."""
    from tools import synth_binding as sb
    data = sb.generate(sb.TEXT, 5, 70000)
    try:
        sb.set_content_checksum(True)
        frame, _ = sb.compress(data)
    finally:
        sb.set_content_checksum(False)
    plain, _ = sb.compress(data)
    assert frame[4] == 0xA4 and plain[4] == 0xA0 and len(frame) == len(plain) + 4 and frame[5:-4] == plain[5:]
    assert int.from_bytes(frame[-4:], "little") == oracle.xxh64(data) & 0xFFFFFFFF
    rc, out, consumed, _ = oracle.decode_frame(frame, cap=len(data) + 64)
    assert rc == 0 and out == data and consumed == len(frame) - 4
