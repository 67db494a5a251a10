"""sparkzstd_amd -- MI355X-native zstd block-decode hot path behind sparkzstd's FrameReader /
FrameDecompressor API.  The compute path is hand-written HIP (sparkzstd_amd/csrc) behind the
C-ABI of include/mzd.h; this package is the Python host mirror of the reference's interface."""
from . import _lib  # noqa: F401
from .api import Context, Cursor, FrameStream, MzdError, PinnedBuffer, Plan, ResidentBatch, Stream, decode_frames, decode_frames_multi, default_context, device_contexts, shard_frames, split_frames, strerror  # noqa: F401
from .decompression import (BatchFrameReader, DecodeFrames, FrameDecompressor, FrameReader, NewBatchFrameReader,  # noqa: F401
                            NewFrameDecompressor, NewFrameReader, ZstdError)

__all__ = ["Context", "Cursor", "FrameStream", "Plan", "ResidentBatch", "Stream", "PinnedBuffer", "split_frames", "decode_frames", "decode_frames_multi", "shard_frames", "device_contexts", "FrameReader", "FrameDecompressor",
           "NewFrameReader", "NewFrameDecompressor", "BatchFrameReader", "NewBatchFrameReader", "DecodeFrames", "MzdError", "ZstdError"]
