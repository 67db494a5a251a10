/*
 * mzd.h -- C ABI of the MI355X zstd block-decode hot path ("mzd").
 *
 * This is the drop-in boundary underneath KillingSpark/sparkzstd's
 * FrameReader / FrameDecompressor.  The reference has no FFI of its own (pure
 * Go); the seam this ABI replaces is the inside of
 *   (*FrameDecompressor).DecodeNextBlock      decompression/framedecompressor.go:198-244
 * i.e. the three hot loops
 *   (*HuffmanDecodingTable).DecodeStream       structure/huffman.go:221-264
 *   (*SequencesSection).DecodeSequences        structure/sequences.go:126-206
 *   (*FrameDecompressor).ExecuteSequences      decompression/sequence_execution.go:14-63
 *     + Ringbuffer.Push / RepeatBeforeIndex    decompression/ringbuffer.go:102,242
 * and the Raw / RLE block arms                  framedecompressor.go:211-215,229-241.
 *
 * Division of labour (BASELINE.json north_star): the HOST keeps frame / block /
 * section header parsing and FSE / Huffman table construction and describes a
 * whole batch of independent frames with the flat, pointer-free arrays below;
 * the DEVICE (hand-written HIP for gfx950) does Huffman literal decode, FSE
 * sequence decode and sequence execution for every block of every frame.
 *
 * Two host-side producers of these descriptors exist:
 *   - the library's own C++ planner (mzd_plan_*), which mirrors the reference's
 *     Go host code and is what the tests / bench / Python mirror drive;
 *   - the Go cgo shim sketched in INTEGRATION.md, which fills the same structs
 *     from sparkzstd's own parsed Block / FSETable / HuffmanDecodingTable values.
 *
 * All structs are plain C, little-endian, naturally aligned, no pointers inside
 * arrays (cgo-safe).  No torch types anywhere.  Thread-safety: one mzd_ctx per
 * device, used by one host thread at a time; contexts are independent.
 */
#ifndef MZD_H
#define MZD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZD_ABI_VERSION 9

/* ------------------------------------------------------------------ status codes
 * Per-frame status mirrors the reference's sentinel errors (file:line of the
 * Go sentinel each replaces). 0 == success. */
enum {
    MZD_OK = 0,
    MZD_ERR_TRUNCATED = 1,        /* io.ErrUnexpectedEOF from the reference's readers */
    MZD_ERR_MAGIC = 2,            /* framedecompressor.go:128 ErrWrongMagicnumber */
    MZD_ERR_BLOCK_TYPE = 3,       /* block.go:29 ErrIllegalBlockType */
    MZD_ERR_BLOCK_SIZE = 4,       /* block.go:30 ErrIllegalBlockSize */
    MZD_ERR_FSE_TABLE = 5,        /* fse.go:133 ErrDidntReadAllProbabilities (+ build panics :168,:188) */
    MZD_ERR_HUF_WEIGHTS = 6,      /* huffman.go:109-110 ErrWrongSumOfWeights / ErrCorruptedHuffTree */
    MZD_ERR_NO_PREV_TABLE = 7,    /* literals.go:206, sequences.go:271-273 */
    MZD_ERR_BAD_PADDING = 8,      /* huffman.go:218 / fse.go:303 ErrBadPadding */
    MZD_ERR_HUF_BITS = 9,         /* huffman.go:219 ErrDidntUseAllBitsToDecodeHuffman */
    MZD_ERR_HUF_LENGTH = 10,      /* literals.go:207 ErrStreamDidntDecodeToRightLength */
    MZD_ERR_SEQ_BITS = 11,        /* sequences.go:208 ErrNotAllBitsUsed */
    MZD_ERR_CORRUPT_SIZES = 12,   /* framedecompressor.go:90 ErrCorruptSizes, literals.go:43-44 */
    MZD_ERR_LITERALS = 13,        /* sequence_execution.go:11 ErrDidntCopyAllLiteralBytes */
    MZD_ERR_OFFSET = 14,          /* ringbuffer.go:189 ErrCantRepeatBytes (offset beyond produced data / zero) */
    MZD_ERR_DST_FULL = 15,        /* frame output exceeds out_capacity / content size mismatch */
    MZD_ERR_UNSUPPORTED = 16,     /* outside the device path's documented limits */
    MZD_ERR_OUT_OF_BLOCKS = 17,   /* framedecompressor.go:196 ErrOutOfBlocks (host mirror only) */
    MZD_ERR_CHECKSUM = 18,        /* content checksum mismatch; only with mzd_options.verify_checksum -- the
                                     reference never reads the checksum (framereader.go:84-94, Readme.md:62) */
    MZD_ERR_DEVICE = 100,         /* HIP runtime error; see mzd_last_error() */
    MZD_ERR_INVALID_ARG = 101,
    MZD_ERR_NO_DEVICE = 102       /* no HIP device: the product has NO CPU fallback */
};

/* ------------------------------------------------------------------ descriptors */

enum { MZD_BLOCK_RAW = 0, MZD_BLOCK_RLE = 1, MZD_BLOCK_COMPRESSED = 2 }; /* block.go:15-20 */
/* literals.go:22-27; Compressed and Treeless both arrive as MZD_LIT_HUF with the
 * table already resolved by the host (Treeless = same huf_table index as before) */
enum { MZD_LIT_RAW = 0, MZD_LIT_RLE = 1, MZD_LIT_HUF = 2 };

#define MZD_NO_TABLE 0xFFFFFFFFu
#define MZD_UNKNOWN_SIZE 0xFFFFFFFFFFFFFFFFull

/* One frame = one independent unit (tables, offset history {1,4,8} and window
 * are per frame: framedecompressor.go:42-52, ringbuffer.go:36-49). */
typedef struct mzd_frame_desc {
    uint32_t first_block;   /* index into blocks[] */
    uint32_t n_blocks;
    uint64_t out_offset;    /* byte offset of this frame's output slab in `out`; multiple of 16 */
    uint64_t out_capacity;  /* bytes available at out_offset */
    uint64_t content_size;  /* Frame_Content_Size, or MZD_UNKNOWN_SIZE (frame.go:49-61) */
    uint64_t window_size;   /* frame.go:28-36 / framedecompressor.go:358-360; informational + limit check */
    uint32_t checksum;      /* the 4 bytes after the last block (low half of XXH64(content, 0)) when flags say so */
    uint32_t flags;         /* MZD_FRAME_* */
    /* ABI 9 -- a CHUNK of a frame (MZD_FRAME_CONTINUES): the blocks listed are not the frame's first.  The slab then BEGINS with
     * `start` bytes the frame regenerated before them -- at least its last window_size bytes: ringbuffer.go:36-49 keeps exactly
     * that much -- and the chunk's output follows at slab byte `start`; out_capacity counts from the slab's beginning, the reported
     * length includes `start`; hist is the offset history behind the blocks before (framedecompressor.go:23), {1, 4, 8} if the
     * frame had no sequences yet.  Both are ignored without the flag. */
    uint64_t start;
    int32_t hist[3];
    uint32_t reserved;
} mzd_frame_desc;
#define MZD_FRAME_HAS_CHECKSUM 1u /* Content_Checksum_flag set (frame.go:106-108) and the 4 bytes were there */
#define MZD_FRAME_CONTINUES 2u    /* the description is a chunk of a frame: `start`, `hist` (ABI 9); a batch that holds one keeps the
                                     offset history behind every frame's last block for the chunk after it (mzd_fstream_next) */
#define MZD_FRAME_PLAN_STATUS_SHIFT 8 /* bits 8..15: the MZD_ERR_* with which the planner gave the frame up (it then has no
                                        blocks); 0 = planned.  The device reports it as the frame's status. */

/* One block (block.go:22-26 BlockHeader + the slices the reference's section
 * parsers produce: literals.go:30-41,283-361, sequences.go:371-433). Offsets are
 * byte offsets into the input blob `in`. */
typedef struct mzd_block_desc {
    uint8_t type;            /* MZD_BLOCK_* */
    uint8_t lit_type;        /* MZD_LIT_* (compressed blocks) */
    uint8_t lit_streams;     /* 1 or 4 (MZD_LIT_HUF) */
    uint8_t seq_status;      /* n_seq == 0 only: what the reference's DecodeSequences makes of a block whose Number_of_Sequences is zero in
                                its TWO-byte form (0x80 0x00: modes, tables and a bitstream follow all the same; it reads the padding and
                                the three initial states and wants the stream used up, sequences.go:126-208) -- MZD_OK, MZD_ERR_BAD_PADDING
                                or MZD_ERR_SEQ_BITS; reported where the sequence stage's status would be.  0 for every other block */
    uint32_t size;           /* Raw/RLE: regenerated size (Block_Size). Compressed: Block_Size (informational) */
    uint64_t src_off;        /* Raw: payload. RLE: the byte to repeat. */
    uint64_t lit_off;        /* Raw literals: the bytes; RLE literals: the byte; HUF: first stream */
    uint32_t lit_regen;      /* Regenerated_Size of the literals section */
    uint32_t lit_stream_size[4]; /* HUF: compressed size of each stream (jump table + computed 4th) */
    uint32_t huf_table;      /* index into huf_tables[] or MZD_NO_TABLE */
    uint32_t n_seq;          /* Number_of_Sequences (0: block output == literals) */
    uint32_t seq_size;       /* bytes of the sequence bitstream */
    uint64_t seq_off;        /* sequence bitstream */
    uint32_t ll_table, of_table, ml_table; /* indices into fse_tables[] */
    uint32_t reserved1;
} mzd_block_desc;

/* FSE decode table cell == fse.go:10-15 FSETableEntry with the symbol kept
 * UNtranslated; the device applies predefined.go:5-20,36-50 itself.  An RLE-mode
 * table (sequences.go:27-62) is a 1-cell table with acc_log 0. */
typedef struct mzd_fse_entry {
    uint16_t baseline;  /* fse.go:11 Baseline */
    uint8_t nbits;      /* fse.go:13 NumberOfBits */
    uint8_t symbol;     /* literal-length / match-length / offset CODE */
} mzd_fse_entry;

enum { MZD_FSE_LL = 0, MZD_FSE_OF = 1, MZD_FSE_ML = 2 };

/* A table arrives either BUILT (build == 0: fse_entries[entries_off .. +1<<acc_log) are the decoding-table
 * cells, what fse.go:136-230 BuildDecodingTable produces) or as its NORMALISED COUNTS
 * (build == MZD_FSE_FROM_COUNTS | n_symbols): then it occupies only (n_symbols + 1) / 2 cells of
 * fse_entries[], the counts of fse.go:28-130, two int16 per cell in symbol order (-1 == "less
 * than one"), and the library builds the decoding table on the device at upload (SURVEY 8f #1:
 * table construction off the host). */
#define MZD_FSE_FROM_COUNTS 0x8000u
typedef struct mzd_fse_table_desc {
    uint32_t entries_off;  /* first cell in fse_entries[] */
    uint8_t acc_log;       /* table has 1<<acc_log cells; 0 == RLE mode */
    uint8_t kind;          /* MZD_FSE_* */
    uint16_t build;        /* 0, or MZD_FSE_FROM_COUNTS | number of symbols (<= 53) */
} mzd_fse_table_desc;

/* Huffman decode table cell == huffman.go:30-37 (Symbols[j], NumberOfBits[j]),
 * flat table of 1<<max_bits cells in the reference's fill order (huffman.go:163-187). */
typedef struct mzd_huf_entry {
    uint8_t symbol;
    uint8_t nbits;
} mzd_huf_entry;

/* Like the FSE tables, a Huffman table arrives either BUILT (1 << max_bits cells) or as its WEIGHTS
 * (max_bits field = MZD_HUF_FROM_WEIGHTS | n_weights << 8 | max_bits): then it occupies
 * (n_weights + 1) / 2 cells of huf_entries[], cell i = {symbol: weight[2i], nbits: weight[2i+1]}
 * (huffman.go:40-107 decoded weights, the last symbol's weight NOT included: huffman.go:126-131
 * infers it), and the library fills the decode table on the device (huffman.go:112-190). */
#define MZD_HUF_FROM_WEIGHTS 0x80000000u
typedef struct mzd_huf_table_desc {
    uint32_t entries_off;  /* first cell in huf_entries[]; even */
    uint32_t max_bits;     /* 1..11 [| MZD_HUF_FROM_WEIGHTS | n_weights << 8] */
} mzd_huf_table_desc;

#define MZD_IN_PAD 64 /* readable slack required before and after `in` when it is a device pointer */

enum {
    MZD_BATCH_IN_ON_DEVICE = 1u << 0,  /* `in` is a device pointer (with MZD_IN_PAD slack both sides) */
    MZD_BATCH_OUT_ON_DEVICE = 1u << 1  /* `out` is a device pointer */
};

/* A batch of independent frames. Descriptor arrays are always HOST memory. */
typedef struct mzd_batch {
    uint32_t abi_version;  /* MZD_ABI_VERSION */
    uint32_t flags;        /* MZD_BATCH_* */
    const uint8_t *in;     /* concatenated compressed payloads (whole frames or just the sections) */
    uint64_t in_size;
    uint8_t *out;          /* output blob; may be NULL for upload (library allocates on device) */
    uint64_t out_size;     /* must leave >= 64 readable bytes after the last frame's slab (the planner adds 256) */
    const mzd_frame_desc *frames;
    uint32_t n_frames;
    const mzd_block_desc *blocks;
    uint32_t n_blocks;
    const mzd_fse_table_desc *fse_tables;
    uint32_t n_fse_tables;
    const mzd_fse_entry *fse_entries;
    uint32_t n_fse_entries;
    const mzd_huf_table_desc *huf_tables;
    uint32_t n_huf_tables;
    const mzd_huf_entry *huf_entries;
    uint32_t n_huf_entries;
} mzd_batch;

/* ------------------------------------------------------------------ library */

typedef struct mzd_ctx mzd_ctx;       /* one per device */
typedef struct mzd_dbatch mzd_dbatch; /* a batch resident in HBM */
typedef struct mzd_plan mzd_plan;     /* host planner state */

int mzd_abi_version(void);
/* Identity of the binary (ABI 8): the first 16 hex digits of the sha256 over the sources the library was built from
 * (sparkzstd_amd/csrc/Makefile puts it in at build time; "unstamped" for a build that went around the Makefile).  bench.py stamps
 * its line and the rocprof counter files with it: what RAN, not what lies beside it in the tree. */
const char *mzd_build_id(void);
/* "hip-gfx950". There is no CPU backend. */
const char *mzd_backend(void);
const char *mzd_strerror(int code);
/* number of HIP devices, 0 if none / runtime unavailable */
int mzd_device_count(void);

/* Tuning knobs (0 == default). */
typedef struct mzd_options {
    uint32_t seq_variant;     /* sequence decoder: 0 default (= 2); 1: k_seq (two wavefronts per 16 chains); 2: k_seq_q4
                                 (four lanes per chain, five-stage pipeline); 3: k_seq_pipe (one lane per chain, three
                                 stages); see DESIGN.md */
    uint32_t exec_threads;    /* threads per frame in the execution kernel (multiple of 64) */
    uint32_t exec_chunk;      /* k_exec: LDS window chunk in bytes (multiple of 1024, clamped to 4 KiB..128 KiB; 0 = 8 KiB).
                                 k_exec_b: EXTRA LDS per frame on top of its own (a residency cap; clamped to what 64 KiB leaves) */
    uint32_t huf_min_lds;     /* minimum LDS bytes requested per Huffman workgroup (residency cap; 0 = none, the default) */
    uint32_t no_split;        /* 1: never overlap k_seq(tail) with k_exec(head) on a second stream */
    uint32_t assume_cus;      /* testing: pretend the device has this many CUs when choosing the split */
    uint32_t verify_checksum; /* 1: frames that carry a content checksum are verified on the device after the
                                 pass (k_xxh64); a mismatch gives MZD_ERR_CHECKSUM.  Default 0:
                                 the reference never checks it */
    uint32_t seq_window_kib;  /* testing: k_seq_pipe launches cover at most this many KiB of the input blob (default:
                                 just under 4 GiB -- the kernel addresses bitstreams with 32-bit offsets from the
                                 window; larger blobs are decoded window by window) */
    uint32_t huf_variant;     /* Huffman literal kernel: 0 = by the batch (k_huf_seg when streams are long and few, else
                                 k_huf; k_huf FIRST, with its transposed bulk phase, when the batch also has sequences, many
                                 streams and small tables); 1 = k_huf (one lane per stream) beside the sequence stage; 2 =
                                 k_huf_seg (one wavefront per stream, segments decoded in parallel: Huffman codes
                                 self-synchronise); 3 = k_huf first with the transposed bulk phase; see DESIGN.md */
    uint32_t exec_variant;    /* execution kernel: 0 = by the batch; 1 = k_exec (a workgroup per frame, a lane per sequence,
                                 dataflow on an 8 KiB LDS chunk); 2 = k_exec_b (a wavefront per frame, a lane per output
                                 byte, strictly in order; 7.7 KiB of LDS per frame); 3 = BLOCK MODE (a wavefront per
                                 block: the blocks of a frame side by side, 3 or 4 passes + an in-order fix-up walk; frames
                                 below 2 GiB) with k_exec_b's passes, jobs of one block; 4 = block mode with k_exec_c's
                                 passes, jobs of four consecutive blocks (fewer fix-up steps) -- what 0 picks for batches of
                                 few large frames, with the job size chosen by the batch; 5 = k_exec_c (k_exec_b's method,
                                 two bytes per lane and pass, fixed-point passes) */
} mzd_options;

mzd_ctx *mzd_create(int device, const mzd_options *opt, int *err);
void mzd_destroy(mzd_ctx *ctx);
const char *mzd_last_error(mzd_ctx *ctx);

/* Make a batch resident: copies (or adopts, per flags) the input blob, uploads
 * descriptors and tables, derives the per-kernel work lists, allocates scratch
 * (literal buffer, sequence records) and, if batch->out is NULL or host memory, the
 * device output blob.  Replaces nothing in the reference (which has no device). */
int mzd_batch_upload(mzd_ctx *ctx, const mzd_batch *batch, mzd_dbatch **out);
/* Planning ON THE DEVICE: the same residency as mzd_batch_upload, but from the frames
 * themselves -- no host planner, no descriptor arrays.  `in` holds whole zstd frames (magic number first) at
 * frame_off[i] .. +frame_len[i]; with flags = MZD_BATCH_IN_ON_DEVICE it is a device pointer (MZD_IN_PAD slack
 * both sides), else host memory that is copied up.  One lane per frame walks frame / block / section headers
 * (frame.go, block.go, literals.go:67-289, sequences.go:228-433), reads the FSE table descriptions
 * (fse.go:28-130) and Huffman weights (huffman.go:40-131) and writes the work lists; k_fse_build / k_huf_build
 * then build every table.  The host only turns the per-frame counts into offsets.  The output blob is
 * library-owned unless flags has MZD_BATCH_OUT_ON_DEVICE: then `out_dev` (out_dev_size bytes of device
 * memory) is used, and MZD_ERR_DST_FULL is returned if the frames need more (sum of the frames' bounds, each
 * rounded up to 256, + 256).  mzd_batch_out_size / mzd_batch_frame_layout tell where every frame's slab is.
 * A frame that fails to parse gets the same status the host planner gives it (mzd_plan_frame_status),
 * reported through mzd_batch_download like a decode error. */
int mzd_batch_upload_frames(mzd_ctx *ctx, const uint8_t *in, uint64_t in_size, uint32_t flags,
                            const uint64_t *frame_off, const uint64_t *frame_len, uint32_t n_frames,
                            uint8_t *out_dev, uint64_t out_dev_size, mzd_dbatch **out);
/* size of the resident batch's output blob, and the slab (offset, capacity) of every frame in it */
uint64_t mzd_batch_out_size(mzd_dbatch *db);
int mzd_batch_frame_layout(mzd_dbatch *db, uint64_t *out_offset, uint64_t *out_capacity);
/* ------------------------------------------------------------------ streaming
 * Batches of frames pipelined through `depth` recycled device slots on three HIP streams: while batch k
 * decodes, batch k+1 is copied in and planned (k_parse) and batch k-1 is copied out.  The slots' device
 * buffers only grow, so a steady stream allocates nothing.  This is what a streaming consumer of
 * sparkzstd's io.Reader API (framereader.go:51-109) sits on when it has many frames to read: the Go shim's
 * DecodeFrames becomes submit ... wait.  Host buffers should be pinned (mzd_host_alloc) -- with pageable
 * memory the copies are staged by the runtime and do not overlap.  PCIe, not HBM, bounds this path. */
typedef struct mzd_stream mzd_stream;
mzd_stream *mzd_stream_create(mzd_ctx *ctx, uint32_t depth /* 1..8; 2 = double buffering */, int *err);
void mzd_stream_destroy(mzd_stream *s);
/* Enqueues one batch (arguments as mzd_batch_upload_frames, host memory).  The regenerated frames are written
 * to out_host (out_cap bytes; each frame's slab is its bound rounded up to 256 -- MZD_ERR_DST_FULL if the
 * batch needs more).  `in`, the offset arrays and out_host must stay valid until the ticket is collected.
 * Returns MZD_ERR_INVALID_ARG when all `depth` slots hold uncollected tickets. */
int mzd_stream_submit(mzd_stream *s, const uint8_t *in, uint64_t in_size, const uint64_t *frame_off,
                      const uint64_t *frame_len, uint32_t n_frames, uint8_t *out_host, uint64_t out_cap,
                      uint64_t *ticket);
/* Blocks until batch `ticket` is in out_host; per frame: status, regenerated length, offset in out_host
 * (any of the three may be NULL).  Tickets may be collected in any order. */
int mzd_stream_wait(mzd_stream *s, uint64_t ticket, int32_t *status, uint64_t *out_len, uint64_t *out_offset);
/* pinned host memory for the stream's buffers */
void *mzd_host_alloc(uint64_t bytes);
void mzd_host_free(void *p);

/* Frame boundaries of a buffer of concatenated frames (multi-frame files, skippable frames: the reference
 * reads one frame per reader and stops, framereader.go:84-94).  Host code, header walk only (magic, frame
 * header, block headers).  Fills frame_off / frame_len / out_bound (upper bound of the regenerated size: the
 * declared content size capped by what the blocks can produce) for up to `cap` frames; *n_frames = frames
 * found (may exceed cap: call again); *out_total = bytes of output blob the frames need (bounds rounded up
 * to 256) -- what mzd_stream_submit / a caller-owned output of mzd_batch_upload_frames must provide.
 * Returns MZD_OK, or the defect that ended the walk (frames before it are reported). */
int mzd_split_frames(const uint8_t *blob, uint64_t size, uint64_t *frame_off, uint64_t *frame_len,
                     uint64_t *out_bound, uint32_t cap, uint32_t *n_frames, uint64_t *out_total);

/* The hot path: launches the kernels for every frame of the batch on `stream`
 * (a hipStream_t, or NULL for the context's stream). Asynchronous.
 * Replaces huffman.go:221, sequences.go:126, sequence_execution.go:14 and the
 * Raw/RLE arms of framedecompressor.go:198-244 for all frames at once. */
int mzd_batch_run(mzd_ctx *ctx, mzd_dbatch *db, void *stream);
/* Blocks until the context's work is done (the cgo call returns after this). */
int mzd_sync(mzd_ctx *ctx);
/* Copies results back. Any pointer may be NULL. `out_host` receives the whole
 * output blob (out_size bytes). status/out_len have n_frames entries. */
int mzd_batch_download(mzd_ctx *ctx, mzd_dbatch *db, uint8_t *out_host, int32_t *status,
                       uint64_t *out_len);
/* Copies bytes [offset, offset + nbytes) of the resident batch's output blob to `dst` (host memory), after the context's work is
 * done.  What a reader's Read(p) needs (framereader.go:51-109 copies the decoded bytes into the caller's buffer): the frame stays
 * in HBM and every Read moves exactly the bytes it hands out, once, without a host copy of the whole output in between.
 * MZD_ERR_INVALID_ARG when the range leaves the blob. */
int mzd_batch_read_out(mzd_ctx *ctx, mzd_dbatch *db, uint64_t offset, uint8_t *dst, uint64_t nbytes);

/* A reader that keeps a decoded frame in HBM while its consumer drains it (mzd_batch_read_out) needs the OUTPUT only: this frees
 * everything else the batch holds on the device -- the compressed input copy, descriptors, tables, sequence records, literal
 * scratch, block mode's planes (three times the output for a large frame) -- after waiting for the batch's last pass.  The
 * batch can then be read (mzd_batch_read_out, mzd_batch_download, mzd_batch_frame_layout) and freed, not run again
 * (MZD_ERR_INVALID_ARG).  ABI 7. */
int mzd_batch_trim(mzd_ctx *ctx, mzd_dbatch *db);
/* Device pointers of the resident batch (for callers that keep results in HBM). */
void *mzd_batch_device_out(mzd_dbatch *db);
void *mzd_batch_device_status(mzd_dbatch *db);
void *mzd_batch_device_out_len(mzd_dbatch *db);
void mzd_batch_free(mzd_ctx *ctx, mzd_dbatch *db);

/* upload + run + sync + download + free: the single synchronous call a
 * FrameDecompressor.DecodeNextBlock-level shim makes. Returns 0 or the first
 * non-zero frame status. */
int mzd_decode_batch(mzd_ctx *ctx, const mzd_batch *batch, int32_t *status, uint64_t *out_len);

/* Per-kernel durations, averaged over every mzd_batch_run since the last
 * mzd_timing_reset (HIP events recorded on the launch stream around each kernel).
 * Call after mzd_sync. names/ms arrays of capacity `cap`; returns the number of
 * kernels. */
int mzd_last_run_kernel_ms(mzd_ctx *ctx, const char **names, float *ms, int cap);
/* Forget accumulated timings; enable != 0 keeps recording events on later runs. */
void mzd_timing_reset(mzd_ctx *ctx, int enable);
/* Byte counts of the resident batch for roofline accounting. */
typedef struct mzd_batch_stats {
    uint64_t compressed_bytes;    /* sum over blocks of the bytes the kernels must read (C) */
    uint64_t table_bytes;         /* FSE + Huffman table cells shipped to the device */
    uint64_t scratch_bytes;       /* literal buffer + sequence records (implementation traffic) */
    uint64_t out_capacity_bytes;
    uint64_t n_sequences;
    uint64_t n_huf_streams;
    uint64_t n_blocks[3];         /* raw, rle, compressed */
    uint64_t n_fse_built;         /* FSE tables built on the device from their counts (MZD_FSE_FROM_COUNTS) */
    uint64_t n_huf_built;         /* Huffman tables filled on the device from their weights (MZD_HUF_FROM_WEIGHTS) */
    double fse_build_ms;          /* duration of that build (k_fse_build, once per upload) */
    double parse_ms;              /* mzd_batch_upload_frames only: the two planning passes on the device (k_parse) */
} mzd_batch_stats;
int mzd_batch_get_stats(mzd_dbatch *db, mzd_batch_stats *st);
/* Copies the DEVICE decoding table number `table` (1 << acc_log cells, after the device-side build)
 * back to the host; returns the number of cells or -MZD_ERR_*.  Lets tests compare device-built
 * tables with the host planner's, cell by cell. */
int mzd_batch_read_fse_table(mzd_ctx *ctx, mzd_dbatch *db, uint32_t table, mzd_fse_entry *out, uint32_t cap);
/* the same for Huffman decode table `table` (1 << max_bits cells) */
int mzd_batch_read_huf_table(mzd_ctx *ctx, mzd_dbatch *db, uint32_t table, mzd_huf_entry *out, uint32_t cap);

/* ------------------------------------------------------------------ measurement and test hooks
 * Not part of the seam: they exist so that bench.py can quote its roofline fraction against a MEASURED copy
 * ceiling and so that the parity tests can look at the stage boundaries the reference exposes as
 * Go values (literals.go:283-361 LiteralSection.Data, sequences.go:11-15 Sequence) and drive the device's bit
 * reader like bitstream/reversebitstream_test.go drives Reversebitstream. */

/* Plain streaming kernel (16 bytes per lane, grid-stride): reads read_bytes once and writes write_bytes once
 * (device buffers allocated and freed inside the call), `iters` timed launches after one warm-up; *ms = average
 * duration of one launch from HIP events on the context's stream.  (read_bytes + write_bytes) / *ms is the
 * achievable ceiling for a pass whose algorithmic bytes are C = read_bytes in and D = write_bytes out. */
int mzd_measure_copy(mzd_ctx *ctx, uint64_t read_bytes, uint64_t write_bytes, int iters, float *ms);

/* Scratch of a resident batch after mzd_batch_run + mzd_sync, copied to the host.  `offset` and `bytes` are
 * in bytes of the array; reading past its end gives MZD_ERR_INVALID_ARG. */
enum {
    MZD_DEBUG_LITERALS = 0,  /* regenerated Huffman literals; a block's start: mzd_debug_block.lit_src */
    MZD_DEBUG_RECORDS = 1,   /* 8 bytes per sequence: LL:17 | ML:18 | offset:29 (bit 28 of the offset field set:
                                symbolic "history slot (u & 3) at block start minus (u >> 2)", u = field & 0x0FFFFFFF) */
    MZD_DEBUG_TILES = 2,     /* 8 bytes per 64 sequences: running (literal position, output position) of the block */
    MZD_DEBUG_BLOCKS = 3     /* mzd_debug_block per block of the batch */
};
typedef struct mzd_debug_block {
    uint64_t src_off, lit_src, rec_off; /* lit_src: offset in MZD_DEBUG_LITERALS (Huffman), in the input blob (Raw / RLE
                                           literals), or -- lit_in_place -- in the OUTPUT blob */
    uint32_t size, lit_regen, n_seq, tile_off;
    uint8_t type, lit_type;
    uint8_t lit_in_place; /* a block without sequences whose literals the Huffman stage wrote straight to its output */
    uint8_t pad[5];
} mzd_debug_block;
int mzd_batch_debug_read(mzd_ctx *ctx, mzd_dbatch *db, int what, uint64_t offset, void *dst, uint64_t bytes);

/* The device's backward bit reader (row B0: bitstream/reversebitstream.go) on a raw stream: n_reads calls of
 * Read(nbits[i]), nbits[i] <= 32; values[i] = what Read returned, bits_still[i] = BitsStillInStream() after it
 * (reversebitstream.go:13-15: -1 == exactly empty, below: over-read, which reads zeros :23-27,67-75). */
int mzd_debug_backbits(mzd_ctx *ctx, const uint8_t *stream, uint32_t len, const uint8_t *nbits, uint32_t n_reads,
                       uint64_t *values, int64_t *bits_still);

/* Test hook of block mode's fix-up walk (mzd_exec_blk.hip): with step > 0, workgroup 1 of every frame gives up waiting at that
 * step of the walk, as it would if its siblings were not resident; the rescue launch then has to finish the frame.  0 (the
 * default) = off.  For the parity tests only: a decoded frame never depends on it. */
int mzd_debug_force_fixup_bail(mzd_ctx *ctx, uint32_t step);

/* Test hook of the device planner (mzd_parse.hip): mzd_batch_upload_frames parses a frame of `frame_bytes` compressed bytes and
 * more block by block -- a lane per block after a serial walk of the block headers -- instead of in one lane; 0 (the default) =
 * the library's own threshold (1 MiB).  For the parity tests: small frames through the large-frame path.  ABI 7. */
int mzd_debug_plan_unit_bytes(mzd_ctx *ctx, uint64_t frame_bytes);

/* Which kernels the batch's LAST mzd_batch_run took (the library chooses by the batch's shape; the parity tests assert that the
 * path they mean to cover is the one that ran).  ABI 7. */
enum {
    MZD_PASS_BLOCK_MODE = 2, /* the blocks of a frame side by side (mzd_exec_blk.hip) */
    MZD_PASS_EXEC_C = 4,     /* k_exec_c executed the sequences */
    MZD_PASS_EXEC_B = 8,     /* k_exec_b */
    MZD_PASS_SPLIT = 16,     /* the last round of the sequence stage ran beside the execution of the frames before it */
    MZD_PASS_TWO_GROUPS = 32 /* a heterogeneous batch: the frames that hold its longest chains were decoded and executed on a stream of
                              * their own, beside the others (mzd_batch_upload groups them) */
};
uint32_t mzd_batch_last_pass(const mzd_dbatch *db);

/* ------------------------------------------------------------------ host planner
 * C++ restatement of the reference's host side, exposed in C so that tests, the
 * bench and the Python mirror can drive the device without Go:
 *   frame header        structure/frame.go:23-127, framedecompressor.go:130-150,306-374
 *   block header        structure/block.go:33-55
 *   literals header     structure/literals.go:67-289 (+ jump table :46-62)
 *   huffman tree        structure/huffman.go:40-190 (+ fse.go:307-390 for the weights)
 *   sequences header    structure/sequences.go:228-450
 *   FSE tables          fse/fse.go:28-230, fse/predefined.go
 *   table carry-over    framedecompressor.go:283-294 (Repeat / Treeless) */
mzd_plan *mzd_plan_create(void);
/* on != 0: FSE tables are emitted as normalised counts and Huffman tables as weights, and the
 * decode tables are built on the device at upload (MZD_FSE_FROM_COUNTS, MZD_HUF_FROM_WEIGHTS);
 * default off: the planner builds the cells on the host.  Survives mzd_plan_reset. */
void mzd_plan_set_device_tables(mzd_plan *p, int on);
void mzd_plan_destroy(mzd_plan *p);
void mzd_plan_reset(mzd_plan *p);
/* Parses one frame starting at `frame` (magic number first) and appends it to the
 * plan.  The frame's bytes are copied into the plan's input blob.  *consumed =
 * bytes used up to and including the last block (the 4-byte content checksum is
 * never read: SURVEY quirk 4).  Returns MZD_OK or the parse error; a failed frame
 * is still appended (zero blocks) so that indices stay aligned, with the error as
 * its planning status. */
int mzd_plan_add_frame(mzd_plan *p, const uint8_t *frame, uint64_t len, uint64_t *consumed);
/* Many frames, parsed on `n_threads` host threads (0 = hardware concurrency). */
int mzd_plan_add_frames(mzd_plan *p, const uint8_t *blob, const uint64_t *frame_off,
                        const uint64_t *frame_len, uint32_t n_frames, uint32_t n_threads);
/* Lays out the output slabs (content size when known, else n_blocks * 128 KiB),
 * finalises the batch view.  The returned pointer stays valid until the plan is
 * modified or destroyed. */
const mzd_batch *mzd_plan_finalize(mzd_plan *p);
/* planning status of frame i (MZD_OK if it parsed) */
int mzd_plan_frame_status(const mzd_plan *p, uint32_t i);


/* ------------------------------------------------------------------ one frame in chunks (ABI 9)
 * The reference decodes a frame block by block into a ring of the frame's WINDOW size and its reader hands the bytes on as they
 * come (framedecompressor.go:198-303 DecodeNextBlock, ringbuffer.go:36-49, framereader.go:51-109): it never holds a frame
 * whole, in or out.  A batch of whole frames does (every frame's output has a slab).  These entry points are the reference's
 * shape for ONE frame: the frame goes through the device as a sequence of CHUNKS of whole blocks; the device keeps the window,
 * the offset history and nothing else between two chunks.  So a frame may be larger than the device's memory, its source may
 * arrive piecewise, and its first bytes are out before its last ones are in.
 *
 * Host half -- the cursor: walks the frame's blocks as their bytes arrive and describes each chunk as a batch of one frame
 * (MZD_FRAME_CONTINUES, `start`, `hist`), with the tables in force at the chunk's start (framedecompressor.go:283-294) in it. */
typedef struct mzd_cursor mzd_cursor;
mzd_cursor *mzd_cursor_create(void);
void mzd_cursor_destroy(mzd_cursor *c);
/* `src` / `len`: the frame's bytes that no earlier call consumed (magic number first on the first call).  Takes the whole blocks
 * that are there, as long as what they can regenerate stays within max_out (one block at least, whatever max_out), and describes
 * them: *chunk is a batch whose `in` is `src` itself (valid until the next call on the cursor; `src` must stay where it is while
 * the batch is used), with out_capacity = start + the blocks' bound.  `start` and `hist` go into the frame description as they
 * are (hist NULL: {1, 4, 8}).  *consumed: bytes of src that are done with.  *chunk == NULL with MZD_OK: no whole block in src yet
 * -- come back with more bytes (the unconsumed ones first).  *last: the chunk holds the frame's last block; the 4 bytes of a
 * content checksum are not consumed, like the reference's reader leaves them (framereader.go:84-94).  After the last block:
 * MZD_ERR_OUT_OF_BLOCKS (framedecompressor.go:196).  A parse error is returned and sticks. */
int mzd_cursor_next(mzd_cursor *c, const uint8_t *src, uint64_t len, uint64_t max_out, uint64_t start, const int32_t hist[3],
                    uint64_t *consumed, const mzd_batch **chunk, int *last);
/* host threads a chunk's blocks are parsed on (ranges of blocks side by side, stitched in order: the same descriptions as one
 * thread makes); 0, the default: up to eight; 1: the serial walk */
void mzd_cursor_set_threads(mzd_cursor *c, uint32_t n_threads);
/* from the frame header, once a call has consumed it: Window_Size (frame.go:28-36; the content size of a single-segment frame),
 * Frame_Content_Size or MZD_UNKNOWN_SIZE; the content checksum if the frame has one and its bytes were in sight: returns 1 */
uint64_t mzd_cursor_window(const mzd_cursor *c);
uint64_t mzd_cursor_content_size(const mzd_cursor *c);
int mzd_cursor_checksum(const mzd_cursor *c, uint32_t *checksum);

/* Device half -- mzd_fstream: the cursor + two slabs of (window + chunk) bytes on the device.  A chunk is decoded behind the
 * window bytes of the chunks before it, by the same kernels as a batch (block mode for a large chunk); its bytes go to the
 * caller, the last window_size bytes and the history stay.  Replaces FrameDecompressor.DecodeNextBlock + Ringbuffer for one
 * frame (framedecompressor.go:198-303, ringbuffer.go) and is what FrameReader.Read sits on (framereader.go:51-109). */
typedef struct mzd_fstream mzd_fstream;
/* chunk_out: bytes a chunk may regenerate (0: 64 MiB; at least one block).  The device memory the frame needs is
 * 2 x (window + chunk_out) + the chunk's scratch, whatever its length. */
int mzd_fstream_open(mzd_ctx *ctx, uint64_t chunk_out, mzd_fstream **fs);
/* One step: takes the next chunk of whole blocks from src (src / len as for mzd_cursor_next) and sends it to the device; hands
 * out the bytes of the chunk the call BEFORE sent (a two-stage pipeline: the host describes chunk i while the device decodes
 * chunk i - 1, and chunk i - 1 is copied out while chunk i runs).  dst: host memory for a chunk's bytes, dst_cap >= 128 KiB and
 * the same from call to call (a chunk's bound is min(chunk_out, dst_cap)); pinned memory (mzd_host_alloc) copies at the link's
 * rate.  *consumed: bytes of src used; *produced: bytes written to dst -- of an earlier part of the frame than the bytes just
 * consumed.  Both zero with MZD_OK: src holds no whole block yet and nothing is on the device.  After the call that consumed the
 * frame's last block one more call (src may be empty) hands out the last chunk: *done is set with it (then, if the header
 * declared a content size and the frame regenerated another: MZD_ERR_DST_FULL).  A chunk's error is reported by the call that
 * would hand out its bytes.  A frame whose window does not fit beside a chunk in 2 GiB: MZD_ERR_UNSUPPORTED.  Errors stick. */
int mzd_fstream_next(mzd_fstream *fs, const uint8_t *src, uint64_t len, uint8_t *dst, uint64_t dst_cap, uint64_t *consumed,
                     uint64_t *produced, int *done);
/* mzd_cursor_set_threads of the stream's cursor */
void mzd_fstream_set_threads(mzd_fstream *fs, uint32_t n_threads);
/* bytes regenerated so far; the cursor (header fields) */
uint64_t mzd_fstream_total_out(const mzd_fstream *fs);
const mzd_cursor *mzd_fstream_cursor(const mzd_fstream *fs);
void mzd_fstream_close(mzd_fstream *fs);
/* Measurement hook: host milliseconds the stream's calls have spent so far in {the cursor (header and section parsing, table
 * construction), the chunk's upload and launch (allocations, copy-in, table build), waiting for the chunk before it (what of its
 * pass the cursor's work did not cover) and freeing the scratch of the one before that, the copy-out of a chunk's bytes}; returns
 * the number of entries written (4 at most). */
int mzd_fstream_timing(const mzd_fstream *fs, double *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif
