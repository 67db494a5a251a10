"""sparkzstd_amd -- MI355X-native zstd block-decode hot path behind sparkzstd's FrameReader /
FrameDecompressor API.  The compute path is hand-written HIP (sparkzstd_amd/csrc) behind the
C-ABI of include/mzd.h; this package is the Python host mirror of the reference's interface."""
from . import _lib  # noqa: F401
from .api import Context, MzdError, PinnedBuffer, Plan, ResidentBatch, Stream, decode_frames, default_context, split_frames, strerror  # noqa: F401
from .decompression import (DecodeFrames, FrameDecompressor, FrameReader, NewFrameDecompressor,  # noqa: F401
                            NewFrameReader, ZstdError)

__all__ = ["Context", "Plan", "ResidentBatch", "Stream", "PinnedBuffer", "split_frames", "decode_frames", "FrameReader", "FrameDecompressor",
           "NewFrameReader", "NewFrameDecompressor", "DecodeFrames", "MzdError", "ZstdError"]
