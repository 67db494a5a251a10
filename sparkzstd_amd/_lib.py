"""ctypes loader for libmzd.so (the C-ABI of include/mzd.h).  Fails loudly: there is no
Python or CPU fallback for the hot path."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MZD_LIB") or os.path.join(HERE, "libmzd.so")

u8p = ctypes.POINTER(ctypes.c_uint8)

MZD_ABI_VERSION = 9
MZD_UNKNOWN_SIZE = 0xFFFFFFFFFFFFFFFF
MZD_IN_PAD = 64
MZD_BATCH_IN_ON_DEVICE = 1
MZD_BATCH_OUT_ON_DEVICE = 2

# every symbol include/mzd.h declares (tests/test_abi.py checks the export list against the header)
EXPORTS = [
    "mzd_abi_version", "mzd_build_id", "mzd_backend", "mzd_strerror", "mzd_device_count", "mzd_create", "mzd_destroy",
    "mzd_last_error", "mzd_batch_upload", "mzd_batch_run", "mzd_sync", "mzd_batch_download", "mzd_batch_read_out",
    "mzd_batch_device_out", "mzd_batch_device_status", "mzd_batch_device_out_len", "mzd_batch_free",
    "mzd_decode_batch", "mzd_last_run_kernel_ms", "mzd_timing_reset", "mzd_batch_get_stats", "mzd_plan_create",
    "mzd_plan_destroy", "mzd_plan_reset", "mzd_plan_add_frame", "mzd_plan_add_frames",
    "mzd_plan_finalize", "mzd_plan_frame_status", "mzd_plan_set_device_tables", "mzd_batch_read_fse_table", "mzd_batch_read_huf_table",
    "mzd_batch_upload_frames", "mzd_batch_out_size", "mzd_batch_frame_layout",
    "mzd_stream_create", "mzd_stream_destroy", "mzd_stream_submit", "mzd_stream_wait", "mzd_host_alloc", "mzd_host_free", "mzd_split_frames",
    "mzd_measure_copy", "mzd_batch_debug_read", "mzd_debug_backbits", "mzd_debug_force_fixup_bail",
    "mzd_batch_last_pass", "mzd_batch_trim", "mzd_debug_plan_unit_bytes",
    "mzd_cursor_create", "mzd_cursor_destroy", "mzd_cursor_next", "mzd_cursor_set_threads", "mzd_cursor_window", "mzd_cursor_content_size", "mzd_cursor_checksum",
    "mzd_fstream_open", "mzd_fstream_next", "mzd_fstream_set_threads", "mzd_fstream_total_out", "mzd_fstream_cursor", "mzd_fstream_close", "mzd_fstream_timing",
]
MZD_PASS_BLOCK_MODE, MZD_PASS_EXEC_C, MZD_PASS_EXEC_B, MZD_PASS_SPLIT, MZD_PASS_TWO_GROUPS = 2, 4, 8, 16, 32

MZD_DEBUG_LITERALS, MZD_DEBUG_RECORDS, MZD_DEBUG_TILES, MZD_DEBUG_BLOCKS = 0, 1, 2, 3


class DebugBlock(ctypes.Structure):
    _fields_ = [("src_off", ctypes.c_uint64), ("lit_src", ctypes.c_uint64), ("rec_off", ctypes.c_uint64),
                ("size", ctypes.c_uint32), ("lit_regen", ctypes.c_uint32), ("n_seq", ctypes.c_uint32),
                ("tile_off", ctypes.c_uint32), ("type", ctypes.c_uint8), ("lit_type", ctypes.c_uint8),
                ("lit_in_place", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 5)]



class FrameDesc(ctypes.Structure):
    _fields_ = [("first_block", ctypes.c_uint32), ("n_blocks", ctypes.c_uint32),
                ("out_offset", ctypes.c_uint64), ("out_capacity", ctypes.c_uint64),
                ("content_size", ctypes.c_uint64), ("window_size", ctypes.c_uint64),
                ("checksum", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("start", ctypes.c_uint64), ("hist", ctypes.c_int32 * 3), ("reserved", ctypes.c_uint32)]  # ABI 9: a chunk of a frame


MZD_FRAME_HAS_CHECKSUM = 1
MZD_FRAME_CONTINUES = 2
MZD_ERR_CHECKSUM = 18


class BlockDesc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint8), ("lit_type", ctypes.c_uint8), ("lit_streams", ctypes.c_uint8),
                ("seq_status", ctypes.c_uint8), ("size", ctypes.c_uint32), ("src_off", ctypes.c_uint64),
                ("lit_off", ctypes.c_uint64), ("lit_regen", ctypes.c_uint32),
                ("lit_stream_size", ctypes.c_uint32 * 4), ("huf_table", ctypes.c_uint32),
                ("n_seq", ctypes.c_uint32), ("seq_size", ctypes.c_uint32), ("seq_off", ctypes.c_uint64),
                ("ll_table", ctypes.c_uint32), ("of_table", ctypes.c_uint32), ("ml_table", ctypes.c_uint32),
                ("reserved1", ctypes.c_uint32)]


class FseEntry(ctypes.Structure):
    _fields_ = [("baseline", ctypes.c_uint16), ("nbits", ctypes.c_uint8), ("symbol", ctypes.c_uint8)]


class FseTableDesc(ctypes.Structure):
    _fields_ = [("entries_off", ctypes.c_uint32), ("acc_log", ctypes.c_uint8), ("kind", ctypes.c_uint8),
                ("build", ctypes.c_uint16)]


MZD_FSE_FROM_COUNTS = 0x8000
MZD_HUF_FROM_WEIGHTS = 0x80000000


class HufEntry(ctypes.Structure):
    _fields_ = [("symbol", ctypes.c_uint8), ("nbits", ctypes.c_uint8)]


class HufTableDesc(ctypes.Structure):
    _fields_ = [("entries_off", ctypes.c_uint32), ("max_bits", ctypes.c_uint32)]


class Batch(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("in_", ctypes.c_void_p), ("in_size", ctypes.c_uint64),
                ("out", ctypes.c_void_p), ("out_size", ctypes.c_uint64),
                ("frames", ctypes.POINTER(FrameDesc)), ("n_frames", ctypes.c_uint32),
                ("blocks", ctypes.POINTER(BlockDesc)), ("n_blocks", ctypes.c_uint32),
                ("fse_tables", ctypes.POINTER(FseTableDesc)), ("n_fse_tables", ctypes.c_uint32),
                ("fse_entries", ctypes.POINTER(FseEntry)), ("n_fse_entries", ctypes.c_uint32),
                ("huf_tables", ctypes.POINTER(HufTableDesc)), ("n_huf_tables", ctypes.c_uint32),
                ("huf_entries", ctypes.POINTER(HufEntry)), ("n_huf_entries", ctypes.c_uint32)]


class Options(ctypes.Structure):
    _fields_ = [("seq_variant", ctypes.c_uint32), ("exec_threads", ctypes.c_uint32),
                ("exec_chunk", ctypes.c_uint32), ("huf_min_lds", ctypes.c_uint32), ("no_split", ctypes.c_uint32),
                ("assume_cus", ctypes.c_uint32), ("verify_checksum", ctypes.c_uint32), ("seq_window_kib", ctypes.c_uint32),
                ("huf_variant", ctypes.c_uint32), ("exec_variant", ctypes.c_uint32)]


class BatchStats(ctypes.Structure):
    _fields_ = [("compressed_bytes", ctypes.c_uint64), ("table_bytes", ctypes.c_uint64),
                ("scratch_bytes", ctypes.c_uint64), ("out_capacity_bytes", ctypes.c_uint64),
                ("n_sequences", ctypes.c_uint64), ("n_huf_streams", ctypes.c_uint64),
                ("n_blocks", ctypes.c_uint64 * 3), ("n_fse_built", ctypes.c_uint64), ("n_huf_built", ctypes.c_uint64),
                ("fse_build_ms", ctypes.c_double), ("parse_ms", ctypes.c_double)]


_lib = None


TEST_LIB_PATH = os.environ.get("MZD_TEST_LIB") or os.path.join(HERE, "libmzd_test.so")
_test_lib = None


def needs_test_kernels(seq_variant=0, huf_variant=0, exec_variant=0) -> bool:
    """The second implementations the parity tests force (k_seq 1, k_seq_pipe 3; k_huf_seg 2; k_exec_b 2, 3) are compiled into
    libmzd_test.so only (round 6: the release library has one kernel per stage)."""
    return seq_variant in (1, 3) or huf_variant == 2 or exec_variant in (2, 3)


def load_test():
    """libmzd_test.so: the release library + the parity tests' second implementations (-DMZD_TEST_KERNELS).  Test scaffolding:
    a Context that asks for one of those variants loads it (api.Context); nothing else does."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise ImportError(f"{TEST_LIB_PATH} not built (make -C sparkzstd_amd/csrc): it holds the kernel variants of the parity tests")
        _test_lib = _open(TEST_LIB_PATH)
    return _test_lib


def load():
    """Returns the loaded library; raises if libmzd.so is missing (build it: python -c
    'import __graft_entry__ as g; g.build()' or make -C sparkzstd_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: the sparkzstd_amd hot path is HIP only, "
                          "there is no fallback. Run `make -C sparkzstd_amd/csrc`.")
    _lib = _open(LIB_PATH)
    return _lib


def _open(path):
    L = ctypes.CDLL(path)
    vp, i32, u32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
    sig = {
        "mzd_abi_version": (i32, []),
        "mzd_backend": (ctypes.c_char_p, []),
        "mzd_build_id": (ctypes.c_char_p, []),
        "mzd_strerror": (ctypes.c_char_p, [i32]),
        "mzd_device_count": (i32, []),
        "mzd_create": (vp, [i32, ctypes.POINTER(Options), ctypes.POINTER(i32)]),
        "mzd_destroy": (None, [vp]),
        "mzd_last_error": (ctypes.c_char_p, [vp]),
        "mzd_batch_upload": (i32, [vp, ctypes.POINTER(Batch), ctypes.POINTER(vp)]),
        "mzd_batch_run": (i32, [vp, vp, vp]),
        "mzd_sync": (i32, [vp]),
        "mzd_batch_download": (i32, [vp, vp, vp, vp, vp]),
        "mzd_batch_read_out": (i32, [vp, vp, u64, vp, u64]),
        "mzd_batch_device_out": (vp, [vp]),
        "mzd_batch_device_status": (vp, [vp]),
        "mzd_batch_device_out_len": (vp, [vp]),
        "mzd_batch_free": (None, [vp, vp]),
        "mzd_decode_batch": (i32, [vp, ctypes.POINTER(Batch), vp, vp]),
        "mzd_last_run_kernel_ms": (i32, [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), i32]),
        "mzd_timing_reset": (None, [vp, i32]),
        "mzd_batch_get_stats": (i32, [vp, ctypes.POINTER(BatchStats)]),
        "mzd_plan_create": (vp, []),
        "mzd_plan_set_device_tables": (None, [vp, i32]),
        "mzd_batch_read_fse_table": (i32, [vp, vp, u32, vp, u32]),
        "mzd_batch_read_huf_table": (i32, [vp, vp, u32, vp, u32]),
        "mzd_batch_upload_frames": (i32, [vp, vp, u64, u32, vp, vp, u32, vp, u64, ctypes.POINTER(vp)]),
        "mzd_stream_create": (vp, [vp, u32, ctypes.POINTER(i32)]),
        "mzd_stream_destroy": (None, [vp]),
        "mzd_stream_submit": (i32, [vp, vp, u64, vp, vp, u32, vp, u64, ctypes.POINTER(u64)]),
        "mzd_stream_wait": (i32, [vp, u64, vp, vp, vp]),
        "mzd_host_alloc": (vp, [u64]),
        "mzd_host_free": (None, [vp]),
        "mzd_split_frames": (i32, [vp, u64, vp, vp, vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u64)]),
        "mzd_batch_out_size": (u64, [vp]),
        "mzd_batch_frame_layout": (i32, [vp, vp, vp]),
        "mzd_plan_destroy": (None, [vp]),
        "mzd_plan_reset": (None, [vp]),
        "mzd_plan_add_frame": (i32, [vp, vp, u64, ctypes.POINTER(u64)]),
        "mzd_plan_add_frames": (i32, [vp, vp, vp, vp, u32, u32]),
        "mzd_plan_finalize": (ctypes.POINTER(Batch), [vp]),
        "mzd_plan_frame_status": (i32, [vp, u32]),
        "mzd_measure_copy": (i32, [vp, u64, u64, i32, ctypes.POINTER(ctypes.c_float)]),
        "mzd_batch_debug_read": (i32, [vp, vp, i32, u64, vp, u64]),
        "mzd_debug_backbits": (i32, [vp, vp, u32, vp, u32, vp, vp]),
        "mzd_debug_force_fixup_bail": (i32, [vp, u32]),
        "mzd_batch_last_pass": (u32, [vp]),
        "mzd_batch_trim": (i32, [vp, vp]),
        "mzd_debug_plan_unit_bytes": (i32, [vp, u64]),
        "mzd_cursor_create": (vp, []),
        "mzd_cursor_destroy": (None, [vp]),
        "mzd_cursor_set_threads": (None, [vp, u32]),
        "mzd_cursor_next": (i32, [vp, vp, u64, u64, u64, vp, ctypes.POINTER(u64), ctypes.POINTER(ctypes.POINTER(Batch)), ctypes.POINTER(i32)]),
        "mzd_cursor_window": (u64, [vp]),
        "mzd_cursor_content_size": (u64, [vp]),
        "mzd_cursor_checksum": (i32, [vp, ctypes.POINTER(u32)]),
        "mzd_fstream_open": (i32, [vp, u64, ctypes.POINTER(vp)]),
        "mzd_fstream_next": (i32, [vp, vp, u64, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(i32)]),
        "mzd_fstream_total_out": (u64, [vp]),
        "mzd_fstream_cursor": (vp, [vp]),
        "mzd_fstream_close": (None, [vp]),
        "mzd_fstream_set_threads": (None, [vp, u32]),
        "mzd_fstream_timing": (i32, [vp, ctypes.POINTER(ctypes.c_double), i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.mzd_abi_version() != MZD_ABI_VERSION:
        raise ImportError(f"{path}: ABI version mismatch")
    return L
