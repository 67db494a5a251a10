"""Pins the CPU oracle against every golden vector the reference holds for the hot path
(SURVEY.md 8c): decodecorpus pairs, the predefined LL FSE table, reverse-bitstream reads and
the ring buffer KATs + property tests.  CPU only."""
import ctypes
import random

from tests.conftest import check_expected
from tests.oracle_binding import FseTable, Rbs, Ring


def test_decodecorpus_bit_exact(oracle, corpus):
    """reference acceptance test: cmd/sparkzstd/main.go:66-108 (FrameReader output == original)."""
    assert len(corpus) == 100
    for name, comp, length, sha, exp in corpus:
        rc, out, consumed, _ = oracle.decode_frame(comp, cap=length + 64)
        assert rc == 0, f"{name}: {oracle.strerror(rc)}"
        check_expected(name, out, length, sha, exp)
        # every corpus frame sets the checksum flag; the reference never reads those 4 bytes
        assert consumed == len(comp) - 4, name


def test_reverse_bitstream_edges(oracle, kat):
    """bitstream/reversebitstream_test.go:172-229"""
    k = kat["rbs_edges"]
    data = bytes(k["data"])
    r = Rbs()
    oracle.lib.orc_rbs_init(ctypes.byref(r), data, len(data))
    got = [oracle.lib.orc_rbs_read(ctypes.byref(r), n) for n in k["reads"]]
    assert got == k["expect"]


def test_reverse_bitstream_ramp(oracle):
    """bitstream/reversebitstream_test.go:7-170: 256-byte ramp read in several bit patterns."""
    data = bytes(range(256))
    ref_bits = []  # bit list, highest bit of last byte first
    for b in reversed(data):
        ref_bits += [(b >> i) & 1 for i in range(7, -1, -1)]
    for pattern in ([8], [4], [5] * 8 + [8], [3] * 8 + [8], [6, 3, 3, 3, 3, 6, 8], [7, 7, 7, 3, 8]):
        r = Rbs()
        oracle.lib.orc_rbs_init(ctypes.byref(r), data, len(data))
        pos = 0
        i = 0
        while pos + pattern[i % len(pattern)] <= len(ref_bits):
            n = pattern[i % len(pattern)]
            want = 0
            for b in ref_bits[pos:pos + n]:
                want = (want << 1) | b
            assert oracle.lib.orc_rbs_read(ctypes.byref(r), n) == want
            pos += n
            i += 1


def test_reverse_bitstream_overread(oracle):
    """reversebitstream.go:23-27,67-75: past the start reads zeros, cursor keeps decrementing."""
    data = bytes([0xFF])
    r = Rbs()
    oracle.lib.orc_rbs_init(ctypes.byref(r), data, 1)
    assert oracle.lib.orc_rbs_read(ctypes.byref(r), 5) == 31
    assert oracle.lib.orc_rbs_read(ctypes.byref(r), 6) == 0b111000  # 3 real bits then zeros
    assert r.offset == -4
    assert oracle.lib.orc_rbs_read(ctypes.byref(r), 7) == 0
    assert r.offset == -11


def test_predefined_ll_table(oracle, kat):
    """fse/fse_test.go:8-41 {Baseline, AddBits, NbBits, BaseValue} for the predefined LL table."""
    t = FseTable()
    assert oracle.lib.orc_fse_build_predefined(ctypes.byref(t), 0) == 0
    assert t.acc_log == 6
    for i, (baseline, addbits, nbits, base) in enumerate(kat["ll_predefined_table"]):
        e = t.table[i]
        assert (e.baseline, e.additional_bits, e.nbits, e.symbol) == (baseline, addbits, nbits, base), i
    oracle.lib.orc_fse_free(ctypes.byref(t))


def _ring_run(oracle, spec):
    rb = Ring()
    oracle.lib.orc_ring_init(ctypes.byref(rb), spec["len"])
    buf = ctypes.create_string_buffer(spec["len"])
    for op, arg, window, dumped in spec["steps"]:
        before = rb.dump_len
        if op == "push":
            oracle.lib.orc_ring_push(ctypes.byref(rb), arg.encode(), len(arg))
        else:
            oracle.lib.orc_ring_repeat(ctypes.byref(rb), arg[0], arg[1])
        n = oracle.lib.orc_ring_string(ctypes.byref(rb), buf)
        assert buf.raw[:n].decode() == window
        got = bytes(bytearray(rb.dump[i] for i in range(before, rb.dump_len))).decode()
        assert got == dumped
    oracle.lib.orc_ring_free(ctypes.byref(rb))


def test_ring_push_kat(oracle, kat):
    """decompression/ringbuffer_test.go:9-83"""
    _ring_run(oracle, kat["ring_push"])


def test_ring_repeat_kat(oracle, kat):
    """decompression/ringbuffer_test.go:85-154"""
    _ring_run(oracle, kat["ring_repeat"])


def test_ring_random_property(oracle):
    """decompression/ringbuffer_test.go:156-317 restated: output == concatenation of all pushes
    and repeats (incl. overlapping), window == its last Len bytes."""
    rng = random.Random(1234)
    L = 100
    rb = Ring()
    oracle.lib.orc_ring_init(ctypes.byref(rb), L)
    model = bytearray()
    first = bytes(33 + rng.randrange(94) for _ in range(50))
    oracle.lib.orc_ring_push(ctypes.byref(rb), first, len(first))
    model += first
    buf = ctypes.create_string_buffer(L)
    for it in range(3000):
        if rng.random() < 0.5:
            d = bytes(33 + rng.randrange(94) for _ in range(rng.randrange(L)))
            oracle.lib.orc_ring_push(ctypes.byref(rb), d, len(d))
            model += d
        else:
            n = rng.randrange(L)
            oldest = 1 + rng.randrange(min(len(model), L) - 1)
            assert oracle.lib.orc_ring_repeat_before_index(ctypes.byref(rb), n, oldest) == 0
            for _ in range(n):
                model.append(model[-oldest])
        k = oracle.lib.orc_ring_string(ctypes.byref(rb), buf)
        assert buf.raw[:k] == bytes(model[-k:])
        assert rb.dump_len == len(model) - k
    oracle.lib.orc_ring_flush(ctypes.byref(rb))
    assert bytes(bytearray(rb.dump[i] for i in range(rb.dump_len))) == bytes(model)
    oracle.lib.orc_ring_free(ctypes.byref(rb))


def test_trace_consistency(oracle, corpus):
    """The per-block trace (literals, sequences) re-executed in Python reproduces the output."""
    name, comp, length, sha, exp = corpus[6]
    rc, out, _, tr = oracle.decode_frame(comp, cap=length + 64, want_trace=True)
    assert rc == 0
    assert sum(b["n_seq"] for b in tr["blocks"]) == len(tr["seqs"])
    assert tr["blocks"][-1]["out_end"] == length


def test_xxh64_restatement_matches_the_corpus_checksums(oracle, corpus):
    """Every decodecorpus frame carries a content checksum = low 32 bits of XXH64(original, 0), little
    endian after the last block (the reference never reads it): 100 golden vectors for the oracle's
    XXH64, plus the published test values for the short-input branches."""
    import hashlib  # noqa: F401  (only to make clear no third-party xxhash is involved)
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert oracle.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1
    n = 0
    for name, comp, length, sha, exp in corpus:
        assert (comp[4] >> 2) & 1, name  # Content_Checksum_flag
        rc, out, consumed, _ = oracle.decode_frame(comp, cap=length + 64)
        assert rc == 0 and consumed == len(comp) - 4
        stored = int.from_bytes(comp[consumed:consumed + 4], "little")
        assert oracle.xxh64(out) & 0xFFFFFFFF == stored, name
        n += 1
    assert n == 100
