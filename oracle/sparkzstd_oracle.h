/*
 * sparkzstd_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the zstd decode algorithm exactly as
 * KillingSpark/sparkzstd implements it.  It exists to CHECK the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in
 * the product library (sparkzstd_amd/csrc) includes, links or calls this.
 *
 * Parity pin: the reference itself is Go and cannot be built in this image (no
 * Go toolchain), so the oracle is pinned against the reference's own golden
 * vectors instead -- the 100 decodecorpus pairs, the predefined-LL FSE table of
 * fse/fse_test.go:8-41, the reverse-bitstream vectors of
 * bitstream/reversebitstream_test.go and the ring buffer string KATs of
 * decompression/ringbuffer_test.go:9-154 (see tests/test_oracle_*.py).
 *
 * Each function cites the reference file:line whose behaviour it restates.
 */
#ifndef SPARKZSTD_ORACLE_H
#define SPARKZSTD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes (mirror the reference's sentinel errors, grouped) */
enum {
    ORC_OK = 0,
    ORC_ERR_SRC_TRUNCATED = 1,   /* io.ErrUnexpectedEOF family */
    ORC_ERR_MAGIC = 2,           /* framedecompressor.go:128 ErrWrongMagicnumber */
    ORC_ERR_BLOCK_TYPE = 3,      /* block.go:29 ErrIllegalBlockType */
    ORC_ERR_BLOCK_SIZE = 4,      /* block.go:30 ErrIllegalBlockSize */
    ORC_ERR_FSE_TABLE = 5,       /* fse.go:130 ErrDidntReadAllProbabilities + build panics */
    ORC_ERR_HUF_WEIGHTS = 6,     /* huffman.go:109-110 */
    ORC_ERR_NO_PREV_TABLE = 7,   /* literals.go:206, sequences.go:271-273 */
    ORC_ERR_BAD_PADDING = 8,     /* huffman.go:218, fse.go:303 */
    ORC_ERR_HUF_BITS = 9,        /* huffman.go:219 ErrDidntUseAllBitsToDecodeHuffman */
    ORC_ERR_HUF_LENGTH = 10,     /* literals.go:207 ErrStreamDidntDecodeToRightLength */
    ORC_ERR_SEQ_BITS = 11,       /* sequences.go:208 ErrNotAllBitsUsed */
    ORC_ERR_CORRUPT_SIZES = 12,  /* framedecompressor.go:90 ErrCorruptSizes, literals.go:43-44 */
    ORC_ERR_LITERALS = 13,       /* sequence_execution.go:11 ErrDidntCopyAllLiteralBytes */
    ORC_ERR_OFFSET = 14,         /* ringbuffer.go:189 ErrCantRepeatBytes */
    ORC_ERR_DST_FULL = 15,       /* output capacity exceeded (oracle-only) */
    ORC_ERR_UNSUPPORTED = 16
};

/* ---------------------------------------------------------------- L0 bit I/O */

/* bitstream/reversebitstream.go:3-88 */
typedef struct {
    const uint8_t *data;
    int64_t len;
    int64_t offset; /* index of the next bit to read; -1 == exactly empty */
} orc_rbs;

void orc_rbs_init(orc_rbs *r, const uint8_t *data, int64_t len); /* :9-11 */
uint64_t orc_rbs_read(orc_rbs *r, int n);                       /* :17-88 */
int64_t orc_rbs_bits_still_in_stream(const orc_rbs *r);         /* :13-15 */

/* bitstream/bitstream.go:9-90 (forward, LSB first) over a byte slice */
typedef struct {
    const uint8_t *data;
    int64_t len;
    int64_t bitpos; /* absolute bit index of next bit */
    int err;
} orc_fbs;
void orc_fbs_init(orc_fbs *b, const uint8_t *data, int64_t len);
uint64_t orc_fbs_read(orc_fbs *b, int n);

/* ---------------------------------------------------------------- L1 FSE */

/* fse/fse.go:10-15 FSETableEntry */
typedef struct {
    uint16_t baseline;
    uint8_t additional_bits;
    uint8_t nbits;
    int32_t symbol; /* LL/ML: translated base value; OF / huffman weights: raw symbol */
    uint8_t raw_symbol; /* untranslated FSE symbol (oracle convenience) */
} orc_fse_entry;

#define ORC_FSE_MAX_ACCLOG 20
#define ORC_FSE_MAX_SYMBOLS 256

typedef struct {
    int acc_log;
    int n_values;
    int32_t values[ORC_FSE_MAX_SYMBOLS]; /* probability + 1 (fse.go:19) */
    orc_fse_entry *table;                /* 1<<acc_log entries, malloc'd */
    int is_rle;                          /* sequences.go:27-62 RepeatingDecodingTable */
    int32_t rle_value;
    int rle_additional_bits;
    int64_t state;
} orc_fse_table;

/* fse.go:28-130; returns bytes consumed (>0) or -ORC_ERR_* */
int orc_fse_read_description(orc_fse_table *t, const uint8_t *src, int64_t len);
/* fse.go:136-230; translation / extra may be NULL */
int orc_fse_build(orc_fse_table *t, const int32_t *translation, int n_translation,
                  const uint8_t *extra_bits, int n_extra);
void orc_fse_free(orc_fse_table *t);
/* predefined.go:22,52,70 */
int orc_fse_build_predefined(orc_fse_table *t, int which /*0 LL,1 OF,2 ML*/);
uint32_t orc_highbit32(uint32_t v); /* fse.go:235-249 */

/* ---------------------------------------------------------------- L2 Huffman */

typedef struct {
    int max_bits;
    int n_entries;
    uint8_t symbols[1 << 12];
    uint8_t nbits[1 << 12];
} orc_huf_table;

/* huffman.go:40-107 ; returns bytes consumed or -err ; weights out */
int orc_huf_read_weights(const uint8_t *src, int64_t len, uint8_t *weights, int *n_weights);
/* huffman.go:112-190 */
int orc_huf_build(orc_huf_table *t, const uint8_t *weights, int n_weights);
/* huffman.go:221-264 ; returns symbols decoded (>=0) or -err */
int64_t orc_huf_decode_stream(const orc_huf_table *t, const uint8_t *data, int64_t len,
                              uint8_t *out, int64_t out_cap);

/* ---------------------------------------------------------------- sequences */

typedef struct {
    int32_t match_length, literal_length;
    uint32_t offset; /* raw offset value, repeat codes unresolved (sequences.go:11-15) */
} orc_sequence;

/* sequences.go:126-206 ; returns 0 or err */
int orc_decode_sequences(orc_fse_table *ll, orc_fse_table *of, orc_fse_table *ml,
                         const uint8_t *data, int64_t len, int n_seq, orc_sequence *out);

/* sequence_execution.go:65-114 */
int64_t orc_next_offset(int64_t hist[3], uint32_t offset_value, int32_t literal_length);

/* ---------------------------------------------------------------- ring buffer */

/* decompression/ringbuffer.go, semantics only: a window of `len` bytes that
 * streams evicted bytes, in order, to a dump sink. */
typedef struct {
    uint8_t *data;
    int len;
    int offset;
    int all_dirty;
    uint8_t *dump;      /* growing dump sink */
    size_t dump_len, dump_cap;
} orc_ring;
int orc_ring_init(orc_ring *rb, int len);
void orc_ring_free(orc_ring *rb);
int orc_ring_push(orc_ring *rb, const uint8_t *d, int n);        /* ringbuffer.go:102-178 */
int orc_ring_repeat(orc_ring *rb, int n, int after);             /* :197-233 */
int orc_ring_repeat_before_index(orc_ring *rb, int n, int oldest); /* :242-277 */
void orc_ring_flush(orc_ring *rb);                               /* :326 */
/* content stitched in order (ringbuffer.go:331-337 String) ; returns length */
int orc_ring_string(const orc_ring *rb, uint8_t *out);

/* ---------------------------------------------------------------- frames */

typedef struct {
    uint64_t window_size, content_size;
    int single_segment, checksum_flag, dict_id_bytes, fcs_bytes, header_bytes;
} orc_frame_header;
/* frame.go:23-127 + framedecompressor.go:130-150,306-374 ; src starts at magic */
int orc_parse_frame_header(const uint8_t *src, size_t n, orc_frame_header *h);

/* Content checksum (zstd frame format: low 32 bits of XXH64(content, seed 0), little endian, after
 * the last block).  The REFERENCE never reads it (framereader.go:84-94, Readme.md:62: "checksum
 * is not checked"); xxHash is third-party to it and absent from /root/reference, so this is a
 * restatement of the published XXH64 specification (xxHash 0.8 doc/xxhash_spec.md), pinned on the
 * 100 checksums that the decodecorpus frames carry.  Used to check the device-side verification
 * (SURVEY 8f #3). */
uint64_t orc_xxh64(const uint8_t *p, size_t n, uint64_t seed);

/* optional per-block trace used by the tests to compare intermediates */
typedef struct {
    int block_type;        /* 0 raw 1 rle 2 compressed */
    uint32_t block_size;
    int lit_type;          /* 0 raw 1 rle 2 compressed 3 treeless */
    uint32_t lit_regen, lit_compressed;
    int lit_streams, huf_max_bits;
    int n_seq;
    int ll_mode, of_mode, ml_mode;
    uint64_t out_begin, out_end;
} orc_block_info;

typedef struct {
    orc_block_info *blocks;
    int n_blocks, cap_blocks;
    /* concatenated intermediates of all compressed blocks */
    uint8_t *literals; size_t n_literals, cap_literals;
    orc_sequence *seqs; size_t n_seqs, cap_seqs;
    int64_t *resolved_offsets; /* parallel to seqs: after nextOffset */
} orc_trace;
void orc_trace_free(orc_trace *t);

/* Whole frame (framedecompressor.go:153-170 Decompress).  `consumed` = bytes of
 * src used up to and including the last block (the reference never reads the
 * checksum).  trace may be NULL. */
int orc_decode_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                     size_t *out_len, size_t *consumed, orc_trace *trace);

/* Many independent frames back to back in one blob (bench cpu_baseline leg). */
int orc_decode_frames(const uint8_t *blob, const uint64_t *frame_off, const uint64_t *frame_len,
                      int n_frames, uint8_t *dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                      uint64_t *out_len, int32_t *status);

/* The same, and a word sum of every frame's regenerated bytes: sum_j w_j * (2j + 1) mod 2^64 over its little-endian 64-bit words
 * (tail zero padded) -- the detector bench.py and the full-size tests compare a synthetic batch with (a position-weighted sum,
 * not a hash; tools/synth takes it from the ORIGINAL content).  Not a reference function: test plumbing that lets the frames of
 * a thread share one destination buffer and still be checked byte for byte against the generator, at the headline's full size. */
int orc_decode_frames_wsum(const uint8_t *blob, const uint64_t *frame_off, const uint64_t *frame_len,
                           int n_frames, uint8_t *dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                           uint64_t *out_len, int32_t *status, uint64_t *wsum);

const char *orc_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif
