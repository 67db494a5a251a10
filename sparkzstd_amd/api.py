"""Batch-level host API over the C-ABI: Plan (host planner), Context (one per GPU),
ResidentBatch (a batch in HBM), decode_frames()."""
import ctypes
import functools
import threading

import numpy as np

from . import _lib
from ._lib import Batch, BatchStats, Options


# a bytes object that the copy engine fills before anybody sees it (the C API's way: PyBytes_FromStringAndSize(NULL, n))
_PyBytes_New = ctypes.pythonapi.PyBytes_FromStringAndSize
_PyBytes_New.restype = ctypes.py_object
_PyBytes_New.argtypes = [ctypes.c_char_p, ctypes.c_ssize_t]
_PyBytes_AsString = ctypes.pythonapi.PyBytes_AsString
_PyBytes_AsString.restype = ctypes.c_void_p
_PyBytes_AsString.argtypes = [ctypes.py_object]


def _ctx_locked(fn):
    """A method of Context / ResidentBatch / Stream that enters the library with the context: one thread at a time per
    mzd_ctx (its events, its second stream and its run counter are not thread-safe; ctypes drops the GIL around the call).
    The Go shim holds x.mu around the same calls (shim/go/gpu/mzd.go)."""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        ctx = self if isinstance(self, Context) else self.ctx
        with ctx._mu:
            return fn(self, *a, **kw)
    return wrapper


class MzdError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = _lib.load().mzd_strerror(code).decode()
        super().__init__(f"{where}: {msg} (code {code})" if where else f"{msg} (code {code})")


def strerror(code: int) -> str:
    return _lib.load().mzd_strerror(code).decode()


class Plan:
    """Host planner: parses frames, builds FSE/Huffman tables, emits the descriptors of mzd.h.
    Mirrors what sparkzstd's Go host code keeps doing (frame.go, block.go, literals.go:67-289,
    sequences.go:228-450, huffman.go:40-190, fse.go:28-230)."""

    def __init__(self, device_tables: bool = False):
        """device_tables: emit FSE tables as normalised counts and Huffman tables as weights; the library
        builds the decode tables on the device at upload (mzd_plan_set_device_tables)."""
        self._L = _lib.load()
        self._p = self._L.mzd_plan_create()
        if device_tables:
            self._L.mzd_plan_set_device_tables(self._p, 1)
        self._keep = []
        self._batch = None

    def close(self):
        if self._p:
            self._L.mzd_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        self.close()

    def add_frame(self, frame: bytes):
        """-> (status, consumed)"""
        consumed = ctypes.c_uint64()
        buf = (ctypes.c_uint8 * len(frame)).from_buffer_copy(frame) if len(frame) else (ctypes.c_uint8 * 1)()
        rc = self._L.mzd_plan_add_frame(self._p, ctypes.addressof(buf), len(frame), ctypes.byref(consumed))
        self._batch = None
        return rc, consumed.value

    def add_frames(self, blob: np.ndarray, offs: np.ndarray, lens: np.ndarray, threads: int = 0):
        """Adopts `blob` (uint8 array, kept alive) and parses frames [offs[i], offs[i]+lens[i])."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        self._keep += [blob, offs, lens]
        self._batch = None
        return self._L.mzd_plan_add_frames(self._p, blob.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                           len(offs), threads)

    def finalize(self) -> Batch:
        bp = self._L.mzd_plan_finalize(self._p)
        self._batch = bp.contents
        return self._batch

    def frame_status(self, i: int) -> int:
        return self._L.mzd_plan_frame_status(self._p, i)


class ResidentBatch:
    def __init__(self, ctx, handle, batch: Batch = None, n_frames: int = 0):
        self.ctx = ctx
        self._h = handle
        if batch is not None:
            self.n_frames = batch.n_frames
            self.out_size = batch.out_size
            self.frame_out_offset = np.array([batch.frames[i].out_offset for i in range(batch.n_frames)], dtype=np.uint64) \
                if batch.n_frames <= 4096 else None
        else:  # planned on the device (Context.upload_frames): the library knows the layout
            self.n_frames = n_frames
            self.out_size = int(ctx._L.mzd_batch_out_size(handle))
            self.frame_out_offset = self.frame_layout()[0]
        self._batch = batch

    @_ctx_locked
    def frame_layout(self):
        """-> (out_offset uint64[n], out_capacity uint64[n]) of every frame's slab in the output blob"""
        off = np.zeros(self.n_frames, dtype=np.uint64)
        cap = np.zeros(self.n_frames, dtype=np.uint64)
        rc = self.ctx._L.mzd_batch_frame_layout(self._h, off.ctypes.data if self.n_frames else None,
                                                cap.ctypes.data if self.n_frames else None)
        if rc:
            raise MzdError(rc, "mzd_batch_frame_layout")
        return off, cap

    @_ctx_locked
    def run(self, stream=None):
        rc = self.ctx._L.mzd_batch_run(self.ctx._c, self._h, stream)
        if rc:
            raise MzdError(rc, "mzd_batch_run: " + self.ctx.last_error())

    @_ctx_locked
    def download(self, want_out=True):
        """-> (out blob as np.uint8 array or None, status int32[n], out_len uint64[n])"""
        out = np.empty(self.out_size, dtype=np.uint8) if want_out else None
        status = np.empty(self.n_frames, dtype=np.int32)
        out_len = np.empty(self.n_frames, dtype=np.uint64)
        rc = self.ctx._L.mzd_batch_download(self.ctx._c, self._h, out.ctypes.data if want_out and self.out_size else None,
                                            status.ctypes.data if self.n_frames else None,
                                            out_len.ctypes.data if self.n_frames else None)
        if rc:
            raise MzdError(rc, "mzd_batch_download: " + self.ctx.last_error())
        return out, status, out_len

    @_ctx_locked
    def read_out(self, offset: int, dst_addr: int, nbytes: int):
        """bytes [offset, offset + nbytes) of the output blob straight to host address `dst_addr` (mzd_batch_read_out)"""
        rc = self.ctx._L.mzd_batch_read_out(self.ctx._c, self._h, int(offset), dst_addr, int(nbytes))
        if rc:
            raise MzdError(rc, "mzd_batch_read_out: " + self.ctx.last_error())

    @_ctx_locked
    def read_fse_table(self, table: int) -> np.ndarray:
        """Device decoding table `table` as uint32 cells (baseline | nbits << 16 | symbol << 24)."""
        out = np.empty(512, dtype=np.uint32)
        n = self.ctx._L.mzd_batch_read_fse_table(self.ctx._c, self._h, table, out.ctypes.data, 512)
        if n < 0:
            raise MzdError(-n, "mzd_batch_read_fse_table: " + self.ctx.last_error())
        return out[:n].copy()

    @_ctx_locked
    def read_huf_table(self, table: int) -> np.ndarray:
        """Device Huffman decode table `table` as uint16 cells (symbol | nbits << 8)."""
        out = np.empty(2048, dtype=np.uint16)
        n = self.ctx._L.mzd_batch_read_huf_table(self.ctx._c, self._h, table, out.ctypes.data, 2048)
        if n < 0:
            raise MzdError(-n, "mzd_batch_read_huf_table: " + self.ctx.last_error())
        return out[:n].copy()

    @_ctx_locked
    def debug_read(self, what: int, dtype, offset_bytes: int, count: int) -> np.ndarray:
        """Scratch of the batch after run() (mzd_batch_debug_read): what = _lib.MZD_DEBUG_*; `count` items of
        `dtype` from byte offset `offset_bytes` (the library checks the range against the array's extent)."""
        out = np.empty(int(count), dtype=dtype)
        rc = self.ctx._L.mzd_batch_debug_read(self.ctx._c, self._h, what, offset_bytes,
                                              out.ctypes.data if out.size else None, out.nbytes)
        if rc:
            raise MzdError(rc, "mzd_batch_debug_read: " + self.ctx.last_error())
        return out

    @_ctx_locked
    def debug_blocks(self, n_blocks: int):
        """The device view of every block (mzd_debug_block): where its literals / records / tiles are."""
        arr = (_lib.DebugBlock * n_blocks)()
        rc = self.ctx._L.mzd_batch_debug_read(self.ctx._c, self._h, _lib.MZD_DEBUG_BLOCKS, 0, ctypes.addressof(arr) if n_blocks else None,
                                              ctypes.sizeof(arr))
        if rc:
            raise MzdError(rc, "mzd_batch_debug_read: " + self.ctx.last_error())
        return arr

    @_ctx_locked
    def trim(self):
        """keeps the output, the statuses and the layout; frees every other device allocation of the batch (mzd_batch_trim)"""
        rc = self.ctx._L.mzd_batch_trim(self.ctx._c, self._h)
        if rc:
            raise MzdError(rc, "mzd_batch_trim: " + self.ctx.last_error())

    def last_pass(self) -> int:
        """MZD_PASS_* flags of the kernels the last run() took (_lib.MZD_PASS_EXEC_C ...)"""
        return int(self.ctx._L.mzd_batch_last_pass(self._h))

    def device_out_ptr(self):
        return self.ctx._L.mzd_batch_device_out(self._h)

    def device_status_ptr(self):
        return self.ctx._L.mzd_batch_device_status(self._h)

    def device_out_len_ptr(self):
        return self.ctx._L.mzd_batch_device_out_len(self._h)

    def stats(self) -> BatchStats:
        st = BatchStats()
        self.ctx._L.mzd_batch_get_stats(self._h, ctypes.byref(st))
        return st

    @_ctx_locked
    def free(self):
        if self._h:
            self.ctx._L.mzd_batch_free(self.ctx._c, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per GPU (mzd_ctx). Raises if no HIP device: there is no CPU path."""

    def __init__(self, device: int = 0, seq_variant: int = 0, exec_threads: int = 0, exec_chunk: int = 0,
                 huf_min_lds: int = 0, no_split: bool = False, assume_cus: int = 0, verify_checksum: bool = False,
                 seq_window_kib: int = 0, huf_variant: int = 0, exec_variant: int = 0, library: str = None):
        # (a kernel variant that is a second implementation for the parity tests lives in libmzd_test.so: round 6; `library` =
        # "release" / "test" overrides the choice -- the tests of the split itself)
        want_test = library == "test" or (library is None and _lib.needs_test_kernels(seq_variant, huf_variant, exec_variant))
        self._L = _lib.load_test() if want_test else _lib.load()
        self._mu = threading.RLock()  # one thread at a time inside the library per context (see _ctx_locked)
        opt = Options()
        opt.seq_variant = seq_variant
        opt.exec_threads = exec_threads
        opt.exec_chunk = exec_chunk
        opt.huf_min_lds = huf_min_lds
        opt.no_split = 1 if no_split else 0
        opt.assume_cus = assume_cus
        opt.verify_checksum = 1 if verify_checksum else 0  # extension: the reference never checks it
        opt.huf_variant = huf_variant  # 0 auto, 1 k_huf beside the sequence stage, 2 k_huf_seg, 3 k_huf first with its transposed bulk phase
        opt.exec_variant = exec_variant  # 0 auto, 1 k_exec (workgroup per frame, lane per sequence), 2 k_exec_b (wavefront per frame, lane per byte), 3 k_exec_b, blocks side by side, 4 the same in jobs of four blocks
        opt.seq_window_kib = seq_window_kib  # testing: size of the blob window one k_seq_pipe launch covers
        err = ctypes.c_int()
        self._c = self._L.mzd_create(device, ctypes.byref(opt), ctypes.byref(err))
        if not self._c:
            raise MzdError(err.value, "mzd_create")

    def last_error(self) -> str:
        return self._L.mzd_last_error(self._c).decode()

    @_ctx_locked
    def upload(self, batch: Batch, device_in_ptr=None, device_out_ptr=None) -> ResidentBatch:
        """device_in_ptr / device_out_ptr: raw device addresses (e.g. torch tensor .data_ptr())
        that replace batch.in / batch.out; the input must carry MZD_IN_PAD bytes of slack."""
        b = Batch()
        ctypes.memmove(ctypes.byref(b), ctypes.byref(batch), ctypes.sizeof(Batch))
        if device_in_ptr is not None:
            b.in_ = device_in_ptr
            b.flags |= _lib.MZD_BATCH_IN_ON_DEVICE
        if device_out_ptr is not None:
            b.out = device_out_ptr
            b.flags |= _lib.MZD_BATCH_OUT_ON_DEVICE
        h = ctypes.c_void_p()
        rc = self._L.mzd_batch_upload(self._c, ctypes.byref(b), ctypes.byref(h))
        if rc:
            raise MzdError(rc, "mzd_batch_upload: " + self.last_error())
        return ResidentBatch(self, h, batch)

    @_ctx_locked
    def upload_frames(self, blob, frame_off, frame_len, device_in_ptr=None, device_out_ptr=None, device_out_size=0) -> ResidentBatch:
        """Planning on the device (mzd_batch_upload_frames): whole zstd frames in, no host
        planner.  blob: bytes-like / np.uint8 array (host), or pass device_in_ptr (with MZD_IN_PAD slack) and
        blob = its size in bytes.  device_out_ptr / device_out_size: caller-owned device output blob."""
        off = np.ascontiguousarray(frame_off, dtype=np.uint64)
        ln = np.ascontiguousarray(frame_len, dtype=np.uint64)
        assert off.shape == ln.shape
        h = ctypes.c_void_p()
        if device_in_ptr is not None:
            ptr, size, flags = device_in_ptr, int(blob), _lib.MZD_BATCH_IN_ON_DEVICE
        else:
            arr = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob, dtype=np.uint8)
            self._keep = arr
            ptr, size, flags = (arr.ctypes.data if arr.size else None), arr.size, 0
        if device_out_ptr is not None:
            flags |= _lib.MZD_BATCH_OUT_ON_DEVICE
        rc = self._L.mzd_batch_upload_frames(self._c, ptr, size, flags, off.ctypes.data if off.size else None,
                                             ln.ctypes.data if ln.size else None, off.size, device_out_ptr, device_out_size,
                                             ctypes.byref(h))
        if rc:
            raise MzdError(rc, "mzd_batch_upload_frames: " + self.last_error())
        return ResidentBatch(self, h, None, n_frames=off.size)

    @_ctx_locked
    def measure_copy(self, read_bytes: int, write_bytes: int, iters: int = 10) -> float:
        """ms per launch of the plain streaming kernel that reads read_bytes and writes write_bytes (the copy
        ceiling roofline fractions are quoted against, mzd_measure_copy)."""
        ms = ctypes.c_float()
        rc = self._L.mzd_measure_copy(self._c, int(read_bytes), int(write_bytes), iters, ctypes.byref(ms))
        if rc:
            raise MzdError(rc, "mzd_measure_copy: " + self.last_error())
        return float(ms.value)

    @_ctx_locked
    def backbits(self, stream: bytes, reads):
        """The device's backward bit reader on a raw stream (mzd_debug_backbits): -> (values, bits_still_in_stream)"""
        nb = np.asarray(reads, dtype=np.uint8)
        buf = np.frombuffer(bytes(stream), dtype=np.uint8) if len(stream) else np.zeros(1, dtype=np.uint8)
        vals = np.zeros(nb.size, dtype=np.uint64)
        left = np.zeros(nb.size, dtype=np.int64)
        rc = self._L.mzd_debug_backbits(self._c, buf.ctypes.data, len(stream), nb.ctypes.data, nb.size, vals.ctypes.data,
                                        left.ctypes.data)
        if rc:
            raise MzdError(rc, "mzd_debug_backbits: " + self.last_error())
        return [int(v) for v in vals], [int(v) for v in left]

    @_ctx_locked
    def sync(self):
        rc = self._L.mzd_sync(self._c)
        if rc:
            raise MzdError(rc, "mzd_sync: " + self.last_error())

    @_ctx_locked
    def timing_reset(self, enable=True):
        self._L.mzd_timing_reset(self._c, 1 if enable else 0)

    @_ctx_locked
    def kernel_ms(self):
        """average ms per kernel over the runs since timing_reset() (call after sync())"""
        names = (ctypes.c_char_p * 8)()
        ms = (ctypes.c_float * 8)()
        n = self._L.mzd_last_run_kernel_ms(self._c, names, ms, 8)
        return {names[i].decode(): ms[i] for i in range(n)}

    @_ctx_locked
    def close(self):
        if self._c:
            self._L.mzd_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_pool_mu = threading.Lock()  # guards _default_ctx and _device_pool (readers that batch create contexts from background threads)


def default_context(device: int = 0) -> Context:
    with _pool_mu:
        if device not in _default_ctx:
            _default_ctx[device] = Context(device)
        return _default_ctx[device]


def split_frames(blob):
    """Frame boundaries of a buffer of concatenated (and skippable) frames, by header walk on the host
    (mzd_split_frames).  -> (rc, frame_off uint64[n], frame_len uint64[n], out_bound uint64[n], out_total)"""
    L = _lib.load()
    arr = blob if isinstance(blob, np.ndarray) else np.frombuffer(blob, dtype=np.uint8)
    n, total = ctypes.c_uint32(), ctypes.c_uint64()
    cap = 1024
    while True:
        off, ln, ob = (np.zeros(cap, dtype=np.uint64) for _ in range(3))
        rc = L.mzd_split_frames(arr.ctypes.data if arr.size else None, arr.size, off.ctypes.data, ln.ctypes.data, ob.ctypes.data,
                                cap, ctypes.byref(n), ctypes.byref(total))
        if n.value <= cap:
            return rc, off[:n.value].copy(), ln[:n.value].copy(), ob[:n.value].copy(), int(total.value)
        cap = n.value


class PinnedBuffer:
    """Pinned host memory (mzd_host_alloc) as a numpy uint8 array `.a`; free() or garbage collection releases it."""

    def __init__(self, nbytes: int):
        self._L = _lib.load()
        self._p = self._L.mzd_host_alloc(max(int(nbytes), 1))
        if not self._p:
            raise MemoryError("mzd_host_alloc failed")
        self.a = np.ctypeslib.as_array((ctypes.c_uint8 * max(int(nbytes), 1)).from_address(self._p))[:int(nbytes)]

    def free(self):
        if self._p:
            self.a = None
            self._L.mzd_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    """Batches of frames pipelined through `depth` device slots (mzd_stream_*): copy-in +
    device planning of batch k+1 and copy-out of batch k-1 overlap the decode of batch k."""

    def __init__(self, ctx: Context, depth: int = 2):
        self.ctx = ctx
        err = ctypes.c_int()
        self._s = ctx._L.mzd_stream_create(ctx._c, depth, ctypes.byref(err))
        if not self._s:
            raise MzdError(err.value, "mzd_stream_create")
        self._keep = {}

    @_ctx_locked
    def submit(self, blob: np.ndarray, frame_off, frame_len, out: np.ndarray) -> int:
        """blob / out: uint8 arrays (pinned for real overlap: PinnedBuffer(...).a).  -> ticket"""
        off = np.ascontiguousarray(frame_off, dtype=np.uint64)
        ln = np.ascontiguousarray(frame_len, dtype=np.uint64)
        t = ctypes.c_uint64()
        rc = self.ctx._L.mzd_stream_submit(self._s, blob.ctypes.data if blob.size else None, blob.size,
                                           off.ctypes.data if off.size else None, ln.ctypes.data if ln.size else None, off.size,
                                           out.ctypes.data if out.size else None, out.size, ctypes.byref(t))
        if rc:
            raise MzdError(rc, "mzd_stream_submit: " + self.ctx.last_error())
        self._keep[t.value] = (blob, off, ln, out)
        return t.value

    @_ctx_locked
    def wait(self, ticket: int):
        """-> (status int32[n], out_len uint64[n], out_offset uint64[n]) of the batch; its bytes are in `out`"""
        blob, off, ln, out = self._keep[ticket]
        n = off.size
        st = np.zeros(n, dtype=np.int32)
        ol = np.zeros(n, dtype=np.uint64)
        oo = np.zeros(n, dtype=np.uint64)
        rc = self.ctx._L.mzd_stream_wait(self._s, ticket, st.ctypes.data if n else None, ol.ctypes.data if n else None,
                                         oo.ctypes.data if n else None)
        if rc:
            raise MzdError(rc, "mzd_stream_wait: " + self.ctx.last_error())
        del self._keep[ticket]
        return st, ol, oo

    @_ctx_locked
    def close(self):
        if self._s:
            self.ctx._L.mzd_stream_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cursor:
    """The host half of a frame in chunks (mzd_cursor_*, ABI 9): walks the frame's blocks as their bytes arrive and describes every
    chunk of whole blocks as a batch of one frame (framedecompressor.go:198-303 walks them one by one)."""

    def __init__(self, threads: int = 0):
        """threads: host threads a chunk's blocks are parsed on (0: up to eight; 1: the serial walk)"""
        self._L = _lib.load()
        self._c = self._L.mzd_cursor_create()
        if threads:
            self._L.mzd_cursor_set_threads(self._c, threads)
        self._src = None

    def next(self, src, max_out: int, start: int = 0, hist=None):
        """src: the frame's bytes not yet consumed (a uint8 array; kept alive while the chunk is in use).
        -> (status, consumed, batch or None, last)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self._src = src
        consumed = ctypes.c_uint64()
        bp = ctypes.POINTER(Batch)()
        last = ctypes.c_int()
        h = (ctypes.c_int32 * 3)(*hist) if hist is not None else None
        rc = self._L.mzd_cursor_next(self._c, src.ctypes.data if src.size else None, src.size, max_out, start, h, ctypes.byref(consumed),
                                     ctypes.byref(bp), ctypes.byref(last))
        return rc, consumed.value, (bp.contents if bp else None), bool(last.value)

    @property
    def window(self) -> int:
        return self._L.mzd_cursor_window(self._c)

    @property
    def content_size(self) -> int:
        return self._L.mzd_cursor_content_size(self._c)

    def close(self):
        if self._c:
            self._L.mzd_cursor_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()


class FrameStream:
    """One frame through the device in chunks of whole blocks (mzd_fstream_*, ABI 9): the device keeps the frame's window and its
    offset history between two chunks and nothing else, so the frame may be larger than the device's memory, its source may arrive
    piecewise and its first bytes are out before its last ones are in -- FrameDecompressor.DecodeNextBlock + Ringbuffer
    (framedecompressor.go:198-303, ringbuffer.go:36-49) for one frame."""

    def __init__(self, ctx: "Context" = None, chunk_bytes: int = 0, threads: int = 0):
        """threads: host threads the cursor parses a chunk's blocks on (0: up to eight; 1: the serial walk)"""
        self.ctx = ctx or default_context()
        h = ctypes.c_void_p()
        rc = self.ctx._L.mzd_fstream_open(self.ctx._c, chunk_bytes, ctypes.byref(h))
        if rc:
            raise MzdError(rc, "mzd_fstream_open")
        self._h = h
        if threads:
            self.ctx._L.mzd_fstream_set_threads(h, threads)
        self.done = False

    @_ctx_locked
    def next(self, src: np.ndarray, dst: np.ndarray):
        """src: the frame's bytes not yet consumed; dst: where a chunk's bytes go (uint8 arrays; dst at least 128 KiB, the same size
        every call).  A step of a two-stage pipeline: the chunk taken from src goes to the device, the bytes that come back are those
        of the chunk the call before took.  -> (consumed, produced); both 0: src holds no whole block yet and nothing is on the
        device.  After the frame's last block has been consumed one more call (src may be empty) hands out the last chunk and sets
        self.done."""
        consumed, produced, done = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        rc = self.ctx._L.mzd_fstream_next(self._h, src.ctypes.data if src.size else None, src.size, dst.ctypes.data, dst.size,
                                          ctypes.byref(consumed), ctypes.byref(produced), ctypes.byref(done))
        if rc:
            raise MzdError(rc, "mzd_fstream_next: " + self.ctx.last_error())
        self.done = bool(done.value)
        return consumed.value, produced.value

    @property
    def total_out(self) -> int:
        return self.ctx._L.mzd_fstream_total_out(self._h)

    def timing(self) -> dict:
        """host milliseconds spent so far per stage of a call (mzd_fstream_timing)"""
        ms = (ctypes.c_double * 4)()
        n = self.ctx._L.mzd_fstream_timing(self._h, ms, 4)
        return dict(zip(("plan", "upload_and_launch", "wait_for_the_chunk_before", "copy_out")[:n], (round(ms[i], 3) for i in range(n))))

    @property
    def window(self) -> int:
        return self.ctx._L.mzd_cursor_window(self.ctx._L.mzd_fstream_cursor(self._h))

    @_ctx_locked
    def close(self):
        if self._h:
            self.ctx._L.mzd_fstream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def declared_frame_cost(frame) -> int:
    """Compressed + declared decompressed bytes of one frame (the C + D a device pass moves for it), from the frame
    header alone (frame.go:23-61: descriptor byte, window descriptor, dictionary id, Frame_Content_Size).  A frame that
    declares no content size counts with its window (or 128 KiB); anything that is not a frame with its length."""
    n = len(frame)
    if n < 6 or bytes(frame[:4]) != b"\x28\xb5\x2f\xfd":
        return n
    fhd = frame[4]
    fcs_flag, single, did = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5
    window = 128 * 1024
    if not single:
        wd = frame[pos]
        base = 1 << (10 + (wd >> 3))
        window = base + (base >> 3) * (wd & 7)
        pos += 1
    pos += (0, 1, 2, 4)[did]
    fcs_bytes = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if fcs_bytes == 0 or pos + fcs_bytes > n:
        return n + window
    d = int.from_bytes(bytes(frame[pos:pos + fcs_bytes]), "little") + (256 if fcs_bytes == 2 else 0)
    return n + d


def shard_frames(frames, world: int):
    """The contiguous frame range of every one of `world` devices: equal counts when the frames cost the same
    (sharding.frame_range), equal C + D otherwise (sharding.balanced_ranges).  -> list of (lo, hi)."""
    from .sharding import frame_range, balanced_ranges
    costs = [declared_frame_cost(f) for f in frames]
    if not costs or min(costs) == max(costs):
        return [frame_range(len(frames), r, world) for r in range(world)]
    return balanced_ranges(costs, world)


_device_pool = {}


def device_contexts(devices):
    """One Context per entry of `devices` (device ids, or Context objects that are taken as they are); a device id that
    appears twice gets two contexts (two independent streams on one GPU)."""
    seen, out = {}, []
    for d in devices:
        if isinstance(d, Context):
            out.append(d)
            continue
        k = (int(d), seen.get(int(d), 0))
        seen[int(d)] = k[1] + 1
        with _pool_mu:
            if k not in _device_pool:
                _device_pool[k] = Context(int(d))
            out.append(_device_pool[k])
    return out


def decode_frames_multi(frames, devices, device_tables: bool = True, device_plan: bool = False):
    """One batch of independent frames over several GPUs of one node: frames share nothing (tables, offset history and
    window are per frame: framedecompressor.go:42-52), so every device takes a contiguous range of them, one host
    thread and one context per device, and the results are stitched in frame order.  No collective, no RCCL."""
    import threading
    ctxs = device_contexts(devices)
    ranges = shard_frames(frames, len(ctxs))
    outs, sts = [None] * len(frames), [0] * len(frames)
    errors = []

    def work(ctx, lo, hi):
        try:
            o, s = decode_frames(frames[lo:hi], ctx, device_tables=device_tables, device_plan=device_plan)
            outs[lo:hi] = o
            sts[lo:hi] = s
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(c, lo, hi)) for c, (lo, hi) in zip(ctxs, ranges) if hi > lo]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return outs, sts


def decode_frames(frames, ctx: Context = None, device_tables: bool = True, device_plan: bool = False, devices=None):
    """Decodes independent zstd frames (list of bytes) in ONE device batch.
    -> (outputs: list of bytes-or-None, statuses: list of int).  The batched analogue of calling
    sparkzstd's FrameDecompressor.Decompress() (framedecompressor.go:153) once per frame.
    device_tables: FSE tables are shipped as normalised counts and built on the device.
    device_plan: no host planner at all -- headers are parsed on the device too (Context.upload_frames).
    devices: a list of device ids (or Contexts): the batch is split over them (decode_frames_multi)."""
    if devices is not None and len(devices) > 0:
        if len(devices) == 1:
            ctx = device_contexts(devices)[0]
        else:
            return decode_frames_multi(frames, devices, device_tables=device_tables, device_plan=device_plan)
    ctx = ctx or default_context()
    # few large frames: every frame's bytes go from HBM straight into the bytes object that is returned (mzd_batch_read_out) --
    # a host blob first and a copy per frame out of it was most of the host side (one 256 MiB frame: 90.8 ms against 37.3 for
    # the blob alone).  Many frames: one download and a slice each (a copy-engine call per 128 KiB frame would cost more).
    total = sum(len(f) for f in frames)
    if 0 < len(frames) <= 64 and total >= (4 << 20) * len(frames) // 4:
        rb, lay, out_len, sts = decode_frames_resident(frames, ctx, device_tables=device_tables, device_plan=device_plan)
        try:
            outs = []
            for i in range(len(sts)):
                if sts[i] != 0:
                    outs.append(None)
                    continue
                n = int(out_len[i])
                if n == 0:
                    outs.append(b"")
                    continue
                obj = _PyBytes_New(None, n)
                rb.read_out(int(lay[i]), _PyBytes_AsString(obj), n)
                outs.append(obj)
        finally:
            rb.free()
        return outs, sts
    out, lay, out_len, sts = decode_frames_blob(frames, ctx, device_tables=device_tables, device_plan=device_plan)
    outs = []
    for i in range(len(sts)):
        o = int(lay[i])
        outs.append(out[o:o + int(out_len[i])].tobytes() if sts[i] == 0 else None)
    return outs, sts


def decode_frames_resident(frames, ctx: Context = None, device_tables: bool = True, device_plan: bool = False, trim: bool = True):
    """Decodes `frames` and LEAVES the output in HBM: -> (ResidentBatch, slab offset of every frame, out_len of every frame,
    statuses).  The caller reads what it wants with ResidentBatch.read_out and frees the batch -- what a reader does whose
    consumer takes the frame piece by piece (decompression.FrameReader: every Read moves its own bytes over PCIe, once)."""
    ctx = ctx or default_context()
    n = len(frames)
    if n == 1:
        blob = np.frombuffer(frames[0], dtype=np.uint8)
    else:
        blob = np.frombuffer(b"".join(bytes(f) if not isinstance(f, (bytes, bytearray)) else f for f in frames), dtype=np.uint8)
    ln = np.array([len(f) for f in frames], dtype=np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        off[1:] = np.cumsum(ln)[:-1]
    if device_plan:
        rb = ctx.upload_frames(blob, off, ln)
        try:
            rb.run()
            last_pass = rb.last_pass()
            _, status, out_len = rb.download(want_out=False)
            if trim:
                rb.trim()
        except Exception:
            rb.free()
            raise
        rb.pass_flags = last_pass
        return rb, np.asarray(rb.frame_out_offset, dtype=np.uint64), out_len, [int(x) for x in status]
    plan = Plan(device_tables=device_tables)
    try:
        if blob.size == 0:
            blob = np.zeros(1, dtype=np.uint8)
        plan.add_frames(blob, off, ln, threads=0)
        batch = plan.finalize()
        plan_status = [plan.frame_status(i) for i in range(n)]
        lay = np.array([int(batch.frames[i].out_offset) for i in range(n)], dtype=np.uint64)
        rb = ctx.upload(batch)
        try:
            rb.run()
            last_pass = rb.last_pass()
            _, status, out_len = rb.download(want_out=False)
            if trim:
                rb.trim()  # (a slow consumer then pins the frame's bytes in HBM and nothing else: ADVICE r4)
        except Exception:
            rb.free()
            raise
        rb.pass_flags = last_pass
        rb._batch = None  # (the planner's arrays go with the plan; the resident batch no longer needs them)
        return rb, lay, out_len, [plan_status[i] or int(status[i]) for i in range(n)]
    finally:
        plan.close()


def decode_frames_blob(frames, ctx: Context = None, device_tables: bool = True, device_plan: bool = False):
    """decode_frames without the per-frame copies: -> (output blob np.uint8[], slab offset of every frame, out_len of every
    frame, statuses).  Frame i is blob[offset[i] : offset[i] + out_len[i]] when its status is 0.  The input is laid out
    ONCE (a single frame is not copied at all) and the host planner adopts it in place (mzd_plan_add_frames): for one
    large frame -- the reference's own usage, one frame per reader -- the copies used to be most of the host side."""
    ctx = ctx or default_context()
    n = len(frames)
    if n == 1:
        blob = np.frombuffer(frames[0], dtype=np.uint8)
    else:
        blob = np.frombuffer(b"".join(bytes(f) if not isinstance(f, (bytes, bytearray)) else f for f in frames), dtype=np.uint8)
    ln = np.array([len(f) for f in frames], dtype=np.uint64)
    off = np.zeros(n, dtype=np.uint64)
    if n > 1:
        off[1:] = np.cumsum(ln)[:-1]
    if device_plan:
        rb = ctx.upload_frames(blob, off, ln)
        try:
            rb.run()
            out, status, out_len = rb.download()
            lay = rb.frame_out_offset
        finally:
            rb.free()
        return out, np.asarray(lay, dtype=np.uint64), out_len, [int(x) for x in status]
    plan = Plan(device_tables=device_tables)
    try:
        if blob.size == 0:
            blob = np.zeros(1, dtype=np.uint8)
        plan.add_frames(blob, off, ln, threads=0)
        batch = plan.finalize()
        plan_status = [plan.frame_status(i) for i in range(n)]
        lay = np.array([int(batch.frames[i].out_offset) for i in range(n)], dtype=np.uint64)
        rb = ctx.upload(batch)
        try:
            rb.run()
            out, status, out_len = rb.download()
        finally:
            rb.free()
        return out, lay, out_len, [plan_status[i] or int(status[i]) for i in range(n)]
    finally:
        plan.close()
