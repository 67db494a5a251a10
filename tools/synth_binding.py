"""ctypes binding of tools/synth/libmzd_synth.so (workload generator; bench/test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "synth", "libmzd_synth.so")
TEXT, EXP, RANDOM, ZERO = 0, 1, 2, 3
MODE_FULL, MODE_LITERALS, MODE_RAW, MODE_RLE = 0, 1, 2, 3

_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "synth", "synth.cpp")
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "synth")])
        L = ctypes.CDLL(SO)
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
        L.synth_generate.argtypes = [i32, u64, vp, u64]
        L.synth_generate.restype = None
        L.synth_checksum64.argtypes = [vp, u64]
        L.synth_checksum64.restype = u64
        L.synth_compress.argtypes = [vp, u64, i32, vp, u64, ctypes.POINTER(u32)]
        L.synth_compress.restype = u64
        L.synth_make_batch.argtypes = [i32, u64, u32, u32, vp, u64, vp, vp, vp, vp, u32]
        L.synth_make_batch.restype = u64
        L.synth_set_content_checksum.argtypes = [i32]
        L.synth_set_content_checksum.restype = None
        L.synth_set_max_offset.argtypes = [u64]
        L.synth_set_max_offset.restype = None
        _lib = L
    return _lib


def set_content_checksum(on: bool):
    """Frames produced from now on carry the zstd content checksum (low half of XXH64(content, 0))."""
    lib().synth_set_content_checksum(1 if on else 0)


def set_max_offset(n: int):
    """Matches of the frames produced from now on reach back at most n bytes (zstd's windowLog: 2^23 at levels up to 19);
    0 = the default, 2^27."""
    lib().synth_set_max_offset(n)


def generate(kind: int, seed: int, n: int) -> bytes:
    buf = np.empty(n, dtype=np.uint8)
    lib().synth_generate(kind, seed, buf.ctypes.data, n)
    return buf.tobytes()


def compress(data: bytes, mode: int = MODE_FULL):
    """-> (frame bytes, n_seq)"""
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    cap = len(data) + len(data) // 2 + 1024
    dst = np.empty(cap, dtype=np.uint8)
    ns = ctypes.c_uint32()
    n = lib().synth_compress(src.ctypes.data, len(data), mode, dst.ctypes.data, cap, ctypes.byref(ns))
    assert n > 0
    return dst[:n].tobytes(), ns.value


def checksum64(data: bytes) -> int:
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    return lib().synth_checksum64(src.ctypes.data, len(data))


def make_batch(config: int, first: int, count: int, frame_bytes: int = 131072, threads: int = 0):
    """-> (blob uint8[], off uint64[], len uint64[], checksum uint64[], n_seq uint32[])"""
    cap = count * (frame_bytes + 64) + 4096
    blob = np.empty(cap, dtype=np.uint8)
    off = np.empty(count, dtype=np.uint64)
    ln = np.empty(count, dtype=np.uint64)
    ck = np.empty(count, dtype=np.uint64)
    ns = np.empty(count, dtype=np.uint32)
    total = lib().synth_make_batch(config, first, count, frame_bytes, blob.ctypes.data, cap, off.ctypes.data,
                                   ln.ctypes.data, ck.ctypes.data, ns.ctypes.data, threads)
    assert total > 0
    return blob[:total].copy() if total < cap // 2 else blob[:total], off, ln, ck, ns
