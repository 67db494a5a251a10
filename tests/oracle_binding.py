"""ctypes binding of oracle/libsparkzstd_oracle.so -- the CPU checker (test infrastructure only)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
u8p = ctypes.POINTER(ctypes.c_uint8)


class Rbs(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_int64), ("offset", ctypes.c_int64)]


class FseEntry(ctypes.Structure):
    _fields_ = [("baseline", ctypes.c_uint16), ("additional_bits", ctypes.c_uint8),
                ("nbits", ctypes.c_uint8), ("symbol", ctypes.c_int32), ("raw_symbol", ctypes.c_uint8)]


class FseTable(ctypes.Structure):
    _fields_ = [("acc_log", ctypes.c_int), ("n_values", ctypes.c_int),
                ("values", ctypes.c_int32 * 256), ("table", ctypes.POINTER(FseEntry)),
                ("is_rle", ctypes.c_int), ("rle_value", ctypes.c_int32),
                ("rle_additional_bits", ctypes.c_int), ("state", ctypes.c_int64)]


class HufTable(ctypes.Structure):
    _fields_ = [("max_bits", ctypes.c_int), ("n_entries", ctypes.c_int), ("symbols", ctypes.c_uint8 * 4096), ("nbits", ctypes.c_uint8 * 4096)]


class Ring(ctypes.Structure):
    _fields_ = [("data", u8p), ("len", ctypes.c_int), ("offset", ctypes.c_int),
                ("all_dirty", ctypes.c_int), ("dump", u8p), ("dump_len", ctypes.c_size_t),
                ("dump_cap", ctypes.c_size_t)]


class Sequence(ctypes.Structure):
    _fields_ = [("match_length", ctypes.c_int32), ("literal_length", ctypes.c_int32),
                ("offset", ctypes.c_uint32)]


class BlockInfo(ctypes.Structure):
    _fields_ = [("block_type", ctypes.c_int), ("block_size", ctypes.c_uint32),
                ("lit_type", ctypes.c_int), ("lit_regen", ctypes.c_uint32),
                ("lit_compressed", ctypes.c_uint32), ("lit_streams", ctypes.c_int),
                ("huf_max_bits", ctypes.c_int), ("n_seq", ctypes.c_int), ("ll_mode", ctypes.c_int),
                ("of_mode", ctypes.c_int), ("ml_mode", ctypes.c_int),
                ("out_begin", ctypes.c_uint64), ("out_end", ctypes.c_uint64)]


class Trace(ctypes.Structure):
    _fields_ = [("blocks", ctypes.POINTER(BlockInfo)), ("n_blocks", ctypes.c_int),
                ("cap_blocks", ctypes.c_int), ("literals", u8p), ("n_literals", ctypes.c_size_t),
                ("cap_literals", ctypes.c_size_t), ("seqs", ctypes.POINTER(Sequence)),
                ("n_seqs", ctypes.c_size_t), ("cap_seqs", ctypes.c_size_t),
                ("resolved_offsets", ctypes.POINTER(ctypes.c_int64))]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_decode_frame.restype = ctypes.c_int
        L.orc_decode_frame.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
        L.orc_rbs_read.restype = ctypes.c_uint64
        L.orc_rbs_read.argtypes = [ctypes.POINTER(Rbs), ctypes.c_int]
        L.orc_rbs_init.argtypes = [ctypes.POINTER(Rbs), ctypes.c_char_p, ctypes.c_int64]
        L.orc_fse_build_predefined.argtypes = [ctypes.POINTER(FseTable), ctypes.c_int]
        L.orc_fse_free.argtypes = [ctypes.POINTER(FseTable)]
        L.orc_fse_build.argtypes = [ctypes.POINTER(FseTable), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.orc_huf_build.argtypes = [ctypes.POINTER(HufTable), ctypes.c_void_p, ctypes.c_int]
        L.orc_ring_init.argtypes = [ctypes.POINTER(Ring), ctypes.c_int]
        L.orc_ring_free.argtypes = [ctypes.POINTER(Ring)]
        L.orc_ring_push.argtypes = [ctypes.POINTER(Ring), ctypes.c_char_p, ctypes.c_int]
        L.orc_ring_repeat.argtypes = [ctypes.POINTER(Ring), ctypes.c_int, ctypes.c_int]
        L.orc_ring_repeat_before_index.argtypes = [ctypes.POINTER(Ring), ctypes.c_int, ctypes.c_int]
        L.orc_ring_flush.argtypes = [ctypes.POINTER(Ring)]
        L.orc_ring_string.argtypes = [ctypes.POINTER(Ring), ctypes.c_char_p]
        L.orc_trace_free.argtypes = [ctypes.POINTER(Trace)]
        L.orc_strerror.restype = ctypes.c_char_p
        L.orc_xxh64.restype = ctypes.c_uint64
        L.orc_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
        L.orc_decode_frames.restype = ctypes.c_int
        L.orc_decode_frames.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 5
        L.orc_decode_frames_wsum.restype = ctypes.c_int
        L.orc_decode_frames_wsum.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 6

    def decode_frame(self, src: bytes, cap: int = None, want_trace=False):
        """-> (rc, output bytes, consumed, trace-or-None)"""
        if cap is None:
            cap = max(1 << 16, 64 * len(src))
        dst = (ctypes.c_uint8 * cap)()
        ol, cons = ctypes.c_size_t(), ctypes.c_size_t()
        tr = Trace() if want_trace else None
        rc = self.lib.orc_decode_frame(src, len(src), dst, cap, ctypes.byref(ol), ctypes.byref(cons),
                                       ctypes.byref(tr) if tr is not None else None)
        out = bytes(memoryview(dst)[:ol.value])
        trace = None
        if tr is not None:
            trace = {
                "blocks": [{f[0]: getattr(tr.blocks[i], f[0]) for f in BlockInfo._fields_}
                           for i in range(tr.n_blocks)],
                "literals": bytes(memoryview((ctypes.c_uint8 * tr.n_literals).from_address(
                    ctypes.addressof(tr.literals.contents)))) if tr.n_literals else b"",
                "seqs": [(tr.seqs[i].literal_length, tr.seqs[i].match_length, tr.seqs[i].offset,
                          tr.resolved_offsets[i]) for i in range(tr.n_seqs)],
            }
            self.lib.orc_trace_free(ctypes.byref(tr))
        return rc, out, cons.value, trace

    def decode_frames_wsum(self, blob, off, ln, cap, threads=1):
        """Every frame of a batch (numpy: blob uint8, off / ln uint64) through the oracle on `threads` host threads, each frame of a
        thread into the same `cap`-byte buffer -> (status int32[], out_len uint64[], wsum uint64[]): the position-weighted word sum
        of the bytes the ORACLE regenerated, the figure tools/synth takes from the original content (sb.checksum64)."""
        import threading
        import numpy as np
        n = len(off)
        st = np.zeros(n, dtype=np.int32)
        ol = np.zeros(n, dtype=np.uint64)
        ws = np.zeros(n, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ln = np.ascontiguousarray(ln, dtype=np.uint64)
        threads = max(1, min(threads, n))
        per = (n + threads - 1) // threads
        keep, ths = [], []
        for t in range(threads):
            a, b = t * per, min(n, (t + 1) * per)
            if a >= b:
                continue
            dst = np.zeros(cap + 64, dtype=np.uint8)
            doff = np.zeros(b - a, dtype=np.uint64)
            dcap = np.full(b - a, cap, dtype=np.uint64)
            keep.append((dst, doff, dcap))
            ths.append(threading.Thread(target=self.lib.orc_decode_frames_wsum, args=(
                blob.ctypes.data, off[a:b].ctypes.data, ln[a:b].ctypes.data, b - a, dst.ctypes.data, doff.ctypes.data, dcap.ctypes.data,
                ol[a:b].ctypes.data, st[a:b].ctypes.data, ws[a:b].ctypes.data)))
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return st, ol, ws

    def xxh64(self, data: bytes, seed: int = 0) -> int:
        return int(self.lib.orc_xxh64(data, len(data), seed))

    def strerror(self, rc):
        return self.lib.orc_strerror(rc).decode()


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libsparkzstd_oracle.so"])


def load_oracle() -> Oracle:
    so = os.path.join(ORACLE_DIR, "libsparkzstd_oracle.so")
    src = os.path.join(ORACLE_DIR, "sparkzstd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        build_oracle()
    return Oracle(ctypes.CDLL(so))
