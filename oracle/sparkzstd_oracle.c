/*
 * sparkzstd_oracle.c -- TEST INFRASTRUCTURE ONLY (see sparkzstd_oracle.h).
 *
 * Plain-C restatement of KillingSpark/sparkzstd's decode algorithm.  Written
 * from the reference's behaviour (file:line cited per function); structure is
 * buffer-to-buffer instead of io.Reader streaming.  Never linked into the
 * product library.
 */
#include "sparkzstd_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ======================================================================= L0 */

void orc_rbs_init(orc_rbs *r, const uint8_t *data, int64_t len)
{
    r->data = data;
    r->len = len;
    r->offset = len * 8 - 1; /* reversebitstream.go:10 */
}

int64_t orc_rbs_bits_still_in_stream(const orc_rbs *r) { return r->offset; }

/* reversebitstream.go:17-88.  Observable behaviour: returns the n bits ending
 * at the cursor (cursor bit is the MSB of the result), bits below bit 0 read as
 * zero, cursor always moves down by n (also when already negative, :23-27). */
uint64_t orc_rbs_read(orc_rbs *r, int n)
{
    if (n == 0) return 0; /* :18-20 */
    uint64_t v = 0;
    if (r->offset <= -1) { /* :23-27 */
        r->offset -= n;
        return 0;
    }
    int64_t lo = r->offset - n + 1;
    if (lo >= 0 && n <= 56) {
        /* fast path, same result: gather the bytes covering bits [lo, offset] */
        int64_t b0 = lo >> 3;
        int64_t nb = (r->offset >> 3) - b0 + 1; /* <= 8 */
        uint64_t w = 0;
        for (int64_t i = 0; i < nb; i++) w |= (uint64_t)r->data[b0 + i] << (8 * i);
        v = (w >> (lo & 7)) & (((uint64_t)1 << n) - 1);
        r->offset -= n;
        return v;
    }
    for (int i = 0; i < n; i++) {
        int64_t bit = r->offset - i;
        uint64_t b = 0;
        if (bit >= 0) b = (r->data[bit >> 3] >> (bit & 7)) & 1u;
        v = (v << 1) | b;
    }
    r->offset -= n;
    return v;
}

void orc_fbs_init(orc_fbs *b, const uint8_t *data, int64_t len)
{
    b->data = data;
    b->len = len;
    b->bitpos = 0;
    b->err = 0;
}

/* bitstream.go:39-90: LSB-first, little-endian accumulation. */
uint64_t orc_fbs_read(orc_fbs *b, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
        int64_t p = b->bitpos + i;
        if ((p >> 3) >= b->len) {
            b->err = 1;
            return 0;
        }
        v |= (uint64_t)((b->data[p >> 3] >> (p & 7)) & 1u) << i;
    }
    b->bitpos += n;
    return v;
}

/* ======================================================================= L1 */

/* fse.go:235-249 (index of highest set bit; 0 for v==0 like the De Bruijn table) */
uint32_t orc_highbit32(uint32_t v)
{
    uint32_t r = 0;
    while (v >>= 1) r++;
    return r;
}

/* predefined.go:5-20,36-50,64-68 */
static const int32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                    20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400,
                                    0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
static const uint8_t LL_EXTRA[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                     1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                   2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                    21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                    39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                    4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_EXTRA[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                     0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                     2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

void orc_fse_free(orc_fse_table *t)
{
    free(t->table);
    t->table = NULL;
}

/* fse.go:28-130 */
int orc_fse_read_description(orc_fse_table *t, const uint8_t *src, int64_t len)
{
    orc_fbs bs;
    orc_fbs_init(&bs, src, len);
    memset(t, 0, sizeof(*t));
    t->acc_log = (int)orc_fbs_read(&bs, 4) + 5; /* :31-36 */
    if (bs.err) return -ORC_ERR_SRC_TRUNCATED;
    int64_t remaining = (int64_t)1 << t->acc_log; /* :38-39 */
    int cur = 0;
    while (remaining > 0) { /* :44 */
        uint32_t bits_needed = orc_highbit32((uint32_t)(remaining + 1)) + 1; /* :45 */
        uint16_t value = (uint16_t)orc_fbs_read(&bs, (int)bits_needed);
        if (bs.err) return -ORC_ERR_SRC_TRUNCATED;
        uint16_t lowermask = (uint16_t)(((uint16_t)1 << (bits_needed - 1)) - 1); /* :62 */
        uint16_t thresh = (uint16_t)(((uint16_t)1 << bits_needed) - 1 - (uint16_t)(remaining + 1)); /* :63 */
        if ((uint16_t)(value & lowermask) < thresh) { /* :65-77 small number: give one bit back */
            bs.bitpos--;
            value = value & lowermask;
        } else if (value > lowermask) { /* :78-81 */
            value = (uint16_t)(value - thresh);
        }
        if (cur >= ORC_FSE_MAX_SYMBOLS) return -ORC_ERR_FSE_TABLE;
        t->values[cur++] = value; /* :84 */
        int prob = (int)value - 1;
        if (prob == -1) remaining--; /* :89-93 */
        else remaining -= prob;
        if (prob == 0) { /* :96-117 zero-run flags */
            uint64_t skip = 3;
            while (skip == 3) {
                skip = orc_fbs_read(&bs, 2);
                if (bs.err) return -ORC_ERR_SRC_TRUNCATED;
                for (uint64_t i = 0; i < skip; i++) {
                    if (cur >= ORC_FSE_MAX_SYMBOLS) return -ORC_ERR_FSE_TABLE;
                    t->values[cur++] = 1;
                }
            }
        }
    }
    t->n_values = cur;
    if (remaining != 0) return -ORC_ERR_FSE_TABLE; /* :126-128 */
    return (int)((bs.bitpos + 7) / 8);             /* :121-124 */
}

/* fse.go:136-230 */
int orc_fse_build(orc_fse_table *t, const int32_t *translation, int n_translation,
                  const uint8_t *extra_bits, int n_extra)
{
    if (t->acc_log > ORC_FSE_MAX_ACCLOG) return ORC_ERR_FSE_TABLE;
    int tablesize = 1 << t->acc_log;
    int high = tablesize - 1;
    int symbol_next[ORC_FSE_MAX_SYMBOLS];
    t->table = (orc_fse_entry *)calloc((size_t)tablesize, sizeof(orc_fse_entry));
    uint8_t *used = (uint8_t *)calloc((size_t)tablesize, 1);
    if (!t->table || !used) { free(used); return ORC_ERR_UNSUPPORTED; }
    for (int s = 0; s < t->n_values; s++) { /* :146-155 */
        int prob = t->values[s] - 1;
        if (prob == -1) {
            if (high < 0) { free(used); return ORC_ERR_FSE_TABLE; }
            t->table[high].raw_symbol = (uint8_t)s;
            used[high] = 1;
            high--;
            symbol_next[s] = 1;
        } else {
            symbol_next[s] = prob;
        }
    }
    int pos = 0;
    int step = (tablesize >> 1) + (tablesize >> 3) + 3; /* :176 */
    for (int s = 0; s < t->n_values; s++) {             /* :160-184 */
        int prob = t->values[s] - 1;
        for (int i = 0; i < prob; i++) {
            if (used[pos]) { free(used); return ORC_ERR_FSE_TABLE; } /* :166-169 panic */
            t->table[pos].raw_symbol = (uint8_t)s;
            used[pos] = 1;
            pos = (pos + step) & (tablesize - 1);
            int guard = 0;
            while (pos > high) { /* :180-183 */
                pos = (pos + step) & (tablesize - 1);
                if (++guard > tablesize) { free(used); return ORC_ERR_FSE_TABLE; }
            }
        }
    }
    free(used);
    if (pos != 0) return ORC_ERR_FSE_TABLE; /* :186-189 panic */
    for (int i = 0; i < tablesize; i++) {   /* :192-228 */
        orc_fse_entry *e = &t->table[i];
        int s = e->raw_symbol;
        uint32_t next = (uint32_t)symbol_next[s]++;
        e->nbits = (uint8_t)((uint32_t)t->acc_log - orc_highbit32(next));
        e->baseline = (uint16_t)((next << e->nbits) - (uint32_t)tablesize);
        e->symbol = s;
        if (n_translation > s) e->symbol = translation[s]; /* :216-218 */
        if (n_extra > s) e->additional_bits = extra_bits[s]; /* :219-221 */
    }
    t->is_rle = 0;
    return ORC_OK;
}

int orc_fse_build_predefined(orc_fse_table *t, int which)
{
    memset(t, 0, sizeof(*t));
    if (which == 0) { /* predefined.go:22-30 */
        t->acc_log = 6;
        t->n_values = 36;
        for (int i = 0; i < 36; i++) t->values[i] = LL_DEFAULT[i] + 1;
        return orc_fse_build(t, LL_BASE, 36, LL_EXTRA, 36);
    } else if (which == 1) { /* :70-78 */
        t->acc_log = 5;
        t->n_values = 29;
        for (int i = 0; i < 29; i++) t->values[i] = OF_DEFAULT[i] + 1;
        return orc_fse_build(t, NULL, 0, NULL, 0);
    } else { /* :52-60 */
        t->acc_log = 6;
        t->n_values = 53;
        for (int i = 0; i < 53; i++) t->values[i] = ML_DEFAULT[i] + 1;
        return orc_fse_build(t, ML_BASE, 53, ML_EXTRA, 53);
    }
}

/* sequences.go:32-62 DecodingTable interface, both implementations */
static void dt_init_state(orc_fse_table *t, orc_rbs *src)
{
    if (t->is_rle) return;                                   /* sequences.go:57-59 */
    t->state = (int64_t)orc_rbs_read(src, t->acc_log);       /* fse.go:253-257 */
}
static int32_t dt_peek(const orc_fse_table *t)
{
    if (t->is_rle) return t->rle_value;                      /* sequences.go:48-50 */
    return t->table[t->state].symbol;                        /* fse.go:272-278 */
}
static int dt_additional_bits(const orc_fse_table *t)
{
    if (t->is_rle) return t->rle_additional_bits;            /* sequences.go:51-53 */
    return t->table[t->state].additional_bits;               /* fse.go:261-263 */
}
static void dt_next_state(orc_fse_table *t, orc_rbs *src)
{
    if (t->is_rle) return;                                   /* sequences.go:45-47 */
    const orc_fse_entry *e = &t->table[t->state];            /* fse.go:282-290 */
    uint64_t add = orc_rbs_read(src, e->nbits);
    t->state = (int64_t)e->baseline + (int64_t)add;
}

/* fse.go:307-390 with two tables sharing one decoding table (huffman.go:62-78).
 * Returns number of symbols written or -err. */
static int fse_decode_interleaved2(orc_fse_table *a, const uint8_t *src, int64_t len, uint8_t *out,
                                   int out_cap)
{
    orc_fse_table t[2];
    t[0] = *a;
    t[1] = *a; /* shallow copy: separate state, shared table (huffman.go:65) */
    orc_rbs bs;
    orc_rbs_init(&bs, src, len);
    int bits = 0;
    uint64_t x = 0;
    while (x == 0) { /* :314-321 */
        x = orc_rbs_read(&bs, 1);
        bits++;
        if (bits > 8) return -ORC_ERR_BAD_PADDING; /* :323-325 (bounded here) */
    }
    dt_init_state(&t[0], &bs); /* :330-336 */
    dt_init_state(&t[1], &bs);
    int n = 0;
    for (;;) { /* :342-386 */
        for (int idx = 0; idx < 2; idx++) {
            int32_t sym = dt_peek(&t[idx]);
            dt_next_state(&t[idx], &bs);
            if (n >= out_cap) return -ORC_ERR_HUF_WEIGHTS;
            out[n++] = (uint8_t)sym;
            if (orc_rbs_bits_still_in_stream(&bs) < -1) { /* :363-383 */
                int other = (idx + 1) % 2;
                if (n >= out_cap) return -ORC_ERR_HUF_WEIGHTS;
                out[n++] = (uint8_t)dt_peek(&t[other]);
                return n;
            }
        }
    }
}

/* ======================================================================= L2 Huffman */

/* huffman.go:40-107 */
int orc_huf_read_weights(const uint8_t *src, int64_t len, uint8_t *weights, int *n_weights)
{
    if (len < 1) return -ORC_ERR_SRC_TRUNCATED;
    uint8_t header = src[0];
    int used = 1;
    if (header < 128) { /* :48-88 FSE compressed */
        int length_in_byte = header;
        if (1 + (int64_t)length_in_byte > len) return -ORC_ERR_SRC_TRUNCATED;
        orc_fse_table fset;
        int bs = orc_fse_read_description(&fset, src + 1, length_in_byte);
        if (bs < 0) return bs;
        int rc = orc_fse_build(&fset, NULL, 0, NULL, 0);
        if (rc) { orc_fse_free(&fset); return -rc; }
        int stream_len = length_in_byte - bs; /* :67 */
        if (stream_len < 0) { orc_fse_free(&fset); return -ORC_ERR_SRC_TRUNCATED; }
        int n = fse_decode_interleaved2(&fset, src + 1 + bs, stream_len, weights, 255);
        orc_fse_free(&fset);
        if (n < 0) return n;
        *n_weights = n;
        used += length_in_byte;
    } else { /* :89-104 direct */
        int nw = header - 127;
        int nbytes = (nw + 1) / 2;
        if (1 + (int64_t)nbytes > len) return -ORC_ERR_SRC_TRUNCATED;
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            weights[i] = (i % 2 == 0) ? (uint8_t)(b >> 4) : (uint8_t)(b & 0xF);
        }
        *n_weights = nw;
        used += nbytes;
    }
    return used;
}

/* huffman.go:112-190 */
int orc_huf_build(orc_huf_table *t, const uint8_t *weights, int n_weights)
{
    uint64_t sum = 0;
    for (int i = 0; i < n_weights; i++) {
        if (weights[i] > 12) return ORC_ERR_HUF_WEIGHTS;
        if (weights[i] > 0) sum += (uint64_t)1 << (weights[i] - 1); /* :113-120 */
    }
    if (sum == 0) return ORC_ERR_HUF_WEIGHTS;
    uint32_t log = orc_highbit32((uint32_t)sum) + 1; /* :125 */
    uint64_t actual = (uint64_t)1 << log;
    uint64_t left = actual - sum;
    if (left & (left - 1)) return ORC_ERR_HUF_WEIGHTS; /* :128-130 */
    uint32_t last_weight = orc_highbit32((uint32_t)left) + 1; /* :131 */
    int max_bits = (int)log;
    if (max_bits > 12 || n_weights > 255) return ORC_ERR_HUF_WEIGHTS;
    int numbits[257];
    int rank_count[16] = {0};
    for (int i = 0; i < n_weights; i++) { /* :138-145 */
        int nob = 0;
        if (weights[i] > 0) nob = max_bits + 1 - weights[i];
        if (nob < 0) return ORC_ERR_HUF_WEIGHTS;
        numbits[i] = nob;
        rank_count[nob]++;
    }
    int last_nob = max_bits + 1 - (int)last_weight; /* :147-152 */
    numbits[n_weights] = last_nob;
    rank_count[last_nob]++;
    int nsym = n_weights + 1;

    t->max_bits = max_bits;
    t->n_entries = 1 << max_bits;
    int rank_idx[16] = {0};
    for (int i = max_bits; i >= 1; i--) { /* :163-171 longest codes first from index 0 */
        rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1 << (max_bits - i));
        int base = rank_idx[i];
        for (int j = 0; j < rank_idx[i - 1] - rank_idx[i]; j++) {
            if (base + j >= t->n_entries) return ORC_ERR_HUF_WEIGHTS;
            t->nbits[base + j] = (uint8_t)i;
        }
    }
    if (rank_idx[0] != t->n_entries) return ORC_ERR_HUF_WEIGHTS; /* :173-175 */
    for (int i = 0; i < nsym; i++) { /* :177-187 */
        if (numbits[i] != 0) {
            int code = rank_idx[numbits[i]];
            int l = 1 << (max_bits - numbits[i]);
            for (int j = 0; j < l; j++) t->symbols[code + j] = (uint8_t)i;
            rank_idx[numbits[i]] += l;
        }
    }
    return ORC_OK;
}

/* huffman.go:221-264 (+ InitState :192-196, DecodeSymbol :199-216) */
int64_t orc_huf_decode_stream(const orc_huf_table *t, const uint8_t *data, int64_t len, uint8_t *out,
                              int64_t out_cap)
{
    orc_rbs bs;
    orc_rbs_init(&bs, data, len);
    int bitsum = 0;
    uint64_t x = 0;
    while (x == 0 && bitsum <= 8) { /* :227-233 */
        x = orc_rbs_read(&bs, 1);
        bitsum++;
    }
    if (bitsum > 8) return -ORC_ERR_BAD_PADDING; /* :235-237 */
    int state = (int)orc_rbs_read(&bs, t->max_bits); /* :239 */
    int mask = (1 << t->max_bits) - 1;
    int64_t total = 0;
    while (orc_rbs_bits_still_in_stream(&bs) + 1 > -(int64_t)t->max_bits) { /* :248 */
        int sym = t->symbols[state];
        int b = t->nbits[state];
        uint64_t rest = orc_rbs_read(&bs, b);
        state = (int)(((uint32_t)(state << b) + (uint32_t)rest) & (uint32_t)mask); /* :214 */
        if (total >= out_cap) return -ORC_ERR_HUF_LENGTH; /* Go: index-out-of-range panic :254 */
        out[total++] = (uint8_t)sym;
    }
    if (orc_rbs_bits_still_in_stream(&bs) + 1 != -(int64_t)t->max_bits) /* :257-261 */
        return -ORC_ERR_HUF_BITS;
    return total;
}

/* ======================================================================= sequences */

/* sequences.go:126-206 + DecodeSequence :64-123 */
int orc_decode_sequences(orc_fse_table *ll, orc_fse_table *of, orc_fse_table *ml,
                         const uint8_t *data, int64_t len, int n_seq, orc_sequence *out)
{
    orc_rbs bs;
    orc_rbs_init(&bs, data, len);
    int bits = 0;
    uint64_t x = 0;
    while (x == 0) { /* :133-139 */
        x = orc_rbs_read(&bs, 1);
        bits++;
        if (bits > 8) return ORC_ERR_BAD_PADDING; /* :141-143 */
    }
    dt_init_state(ll, &bs); /* :145 order LL, OF, ML */
    dt_init_state(of, &bs);
    dt_init_state(ml, &bs);
    for (int i = 0; i < n_seq; i++) {
        int32_t ofcode = dt_peek(of); /* :67-78 */
        int32_t llcode = dt_peek(ll);
        int32_t mlcode = dt_peek(ml);
        if (ofcode > 31) return ORC_ERR_UNSUPPORTED;
        uint64_t offx = orc_rbs_read(&bs, ofcode);          /* :99 */
        out[i].offset = (uint32_t)(((uint64_t)1 << ofcode) + offx); /* :104 */
        uint64_t mlx = orc_rbs_read(&bs, dt_additional_bits(ml)); /* :106-112 */
        out[i].match_length = mlcode + (int32_t)mlx;
        uint64_t llx = orc_rbs_read(&bs, dt_additional_bits(ll)); /* :114-120 */
        out[i].literal_length = llcode + (int32_t)llx;
        if (i < n_seq - 1) { /* :178-194 order LL, ML, OF */
            dt_next_state(ll, &bs);
            dt_next_state(ml, &bs);
            dt_next_state(of, &bs);
        }
    }
    if (orc_rbs_bits_still_in_stream(&bs) != -1) return ORC_ERR_SEQ_BITS; /* :197-204 */
    return ORC_OK;
}

/* sequence_execution.go:65-114 */
int64_t orc_next_offset(int64_t h[3], uint32_t v, int32_t ll)
{
    int64_t off;
    if (v <= 3 && ll > 0) { /* :68-82 */
        if (v == 1) {
            off = h[0];
        } else if (v == 2) {
            off = h[1];
            h[1] = h[0];
            h[0] = off;
        } else {
            off = h[2];
            h[2] = h[1];
            h[1] = h[0];
            h[0] = off;
        }
    } else if (v <= 3) { /* :84-101, LL == 0 */
        if (v == 1) {
            off = h[1];
            h[1] = h[0];
            h[0] = off;
        } else if (v == 2) {
            off = h[2];
            h[2] = h[1];
            h[1] = h[0];
            h[0] = off;
        } else {
            off = h[0] - 1;
            h[2] = h[1];
            h[1] = h[0];
            h[0] = off;
        }
    } else { /* :102-111 */
        off = (int64_t)v - 3;
        h[2] = h[1];
        h[1] = h[0];
        h[0] = off;
    }
    return off;
}

/* ======================================================================= ring buffer */

static int ring_sink(orc_ring *rb, const uint8_t *p, int n)
{
    if (n <= 0) return 0;
    if (rb->dump_len + (size_t)n > rb->dump_cap) {
        size_t nc = rb->dump_cap ? rb->dump_cap * 2 : 256;
        while (nc < rb->dump_len + (size_t)n) nc *= 2;
        uint8_t *q = (uint8_t *)realloc(rb->dump, nc);
        if (!q) return ORC_ERR_UNSUPPORTED;
        rb->dump = q;
        rb->dump_cap = nc;
    }
    memcpy(rb->dump + rb->dump_len, p, (size_t)n);
    rb->dump_len += (size_t)n;
    return 0;
}

static int ring_dump(orc_ring *rb, int low, int high) /* ringbuffer.go:306-318 */
{
    return ring_sink(rb, rb->data + low, high - low);
}

int orc_ring_init(orc_ring *rb, int len) /* ringbuffer.go:24-34 */
{
    memset(rb, 0, sizeof(*rb));
    rb->data = (uint8_t *)calloc((size_t)(len > 0 ? len : 1), 1);
    rb->len = len;
    return rb->data ? 0 : ORC_ERR_UNSUPPORTED;
}
void orc_ring_free(orc_ring *rb)
{
    free(rb->data);
    free(rb->dump);
    memset(rb, 0, sizeof(*rb));
}

static void ring_dump_all_dirty(orc_ring *rb) /* :80-99 */
{
    if (rb->all_dirty) ring_dump(rb, rb->offset, rb->len);
    ring_dump(rb, 0, rb->offset);
    rb->offset = 0;
    rb->all_dirty = 0;
}

int orc_ring_push(orc_ring *rb, const uint8_t *d, int n) /* :102-178 */
{
    if (n >= rb->len) { /* :106-124 */
        ring_dump_all_dirty(rb);
        /* bytes that would be overwritten anyway go straight to the sink */
        int direct = n - rb->len;
        ring_sink(rb, d, direct);
        memcpy(rb->data, d + direct, (size_t)rb->len);
        rb->offset = 0;
        rb->all_dirty = 1;
        return 0;
    }
    int ol = n + rb->offset;
    if (ol <= rb->len) { /* :127-147 */
        if (rb->all_dirty) ring_dump(rb, rb->offset, ol);
        memcpy(rb->data + rb->offset, d, (size_t)n);
        rb->offset += n;
        if (rb->offset >= rb->len) rb->all_dirty = 1;
        rb->offset %= rb->len;
        return 0;
    }
    int above = rb->len - rb->offset; /* :150-177 wrap */
    if (rb->all_dirty) ring_dump(rb, rb->offset, rb->len);
    memcpy(rb->data + rb->offset, d, (size_t)above);
    int rest = n - above;
    ring_dump(rb, 0, rest);
    memcpy(rb->data, d + above, (size_t)rest);
    rb->offset = rest;
    rb->all_dirty = 1;
    return 0;
}

int orc_ring_repeat(orc_ring *rb, int n, int after) /* :197-233 */
{
    uint8_t *buf = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    if (!buf) return ORC_ERR_UNSUPPORTED;
    int start = rb->offset - after;
    int lower = start - n;
    if (lower >= 0 && start >= 0) {
        memcpy(buf, rb->data + lower, (size_t)n);
    } else {
        if (!rb->all_dirty) { free(buf); return ORC_ERR_OFFSET; } /* :206-214 */
        int from_top = -lower;
        if (lower < 0 && start >= 0) { /* :216-218 */
            memcpy(buf, rb->data + rb->len - from_top, (size_t)from_top);
            memcpy(buf + from_top, rb->data, (size_t)start);
        } else { /* :220-222 */
            int skip_top = -start;
            memcpy(buf, rb->data + rb->len - from_top, (size_t)(from_top - skip_top));
        }
    }
    orc_ring_push(rb, buf, n);
    free(buf);
    return 0;
}

int orc_ring_repeat_before_index(orc_ring *rb, int n, int oldest) /* :242-277 */
{
    int skip = oldest - n;
    if (skip < 0) { /* overlapping: generate byte by byte (:250-273) */
        /* Output-equivalent restatement: bytes leave the window in order; the
         * reference pre-dumps the cells it is about to overwrite (quirk 1 of
         * SURVEY 8a for n >= Len is NOT reproduced). */
        for (int i = 0; i < n; i++) {
            int idx = rb->offset - oldest;
            if (idx < 0) idx += rb->len;
            uint8_t b = rb->data[idx];
            if (rb->all_dirty) ring_dump(rb, rb->offset, rb->offset + 1);
            rb->data[rb->offset] = b;
            rb->offset = (rb->offset + 1) % rb->len;
            if (rb->offset == 0) rb->all_dirty = 1;
        }
        return 0;
    }
    return orc_ring_repeat(rb, n, skip);
}

void orc_ring_flush(orc_ring *rb) { ring_dump_all_dirty(rb); } /* :326-328 */

int orc_ring_string(const orc_ring *rb, uint8_t *out) /* :331-337 */
{
    if (rb->all_dirty) {
        memcpy(out, rb->data + rb->offset, (size_t)(rb->len - rb->offset));
        memcpy(out + (rb->len - rb->offset), rb->data, (size_t)rb->offset);
        return rb->len;
    }
    memcpy(out, rb->data, (size_t)rb->offset);
    return rb->offset;
}

/* ======================================================================= XXH64 (published spec) */

#define XP1 0x9E3779B185EBCA87ULL
#define XP2 0xC2B2AE3D27D4EB4FULL
#define XP3 0x165667B19E3779F9ULL
#define XP4 0x85EBCA77C2B2AE63ULL
#define XP5 0x27D4EB2F165667C5ULL
static uint64_t x_rotl(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
static uint64_t x_rd64(const uint8_t *p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
static uint64_t x_rd32(const uint8_t *p) { return (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24); }
static uint64_t x_round(uint64_t acc, uint64_t in) { return x_rotl(acc + in * XP2, 31) * XP1; }
static uint64_t x_merge(uint64_t h, uint64_t v) { return (h ^ x_round(0, v)) * XP1 + XP4; }

uint64_t orc_xxh64(const uint8_t *p, size_t n, uint64_t seed)
{
    const uint8_t *end = p + n;
    uint64_t h;
    if (n >= 32) { /* spec step 1-2: four accumulators over 32-byte stripes */
        uint64_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        while ((size_t)(end - p) >= 32) {
            v1 = x_round(v1, x_rd64(p));
            v2 = x_round(v2, x_rd64(p + 8));
            v3 = x_round(v3, x_rd64(p + 16));
            v4 = x_round(v4, x_rd64(p + 24));
            p += 32;
        }
        h = x_rotl(v1, 1) + x_rotl(v2, 7) + x_rotl(v3, 12) + x_rotl(v4, 18); /* step 3: convergence */
        h = x_merge(h, v1);
        h = x_merge(h, v2);
        h = x_merge(h, v3);
        h = x_merge(h, v4);
    } else {
        h = seed + XP5;
    }
    h += (uint64_t)n; /* step 4 */
    while ((size_t)(end - p) >= 8) { /* step 5: remaining input */
        h ^= x_round(0, x_rd64(p));
        h = x_rotl(h, 27) * XP1 + XP4;
        p += 8;
    }
    if ((size_t)(end - p) >= 4) {
        h ^= x_rd32(p) * XP1;
        h = x_rotl(h, 23) * XP2 + XP3;
        p += 4;
    }
    while (p < end) {
        h ^= (uint64_t)(*p++) * XP5;
        h = x_rotl(h, 11) * XP1;
    }
    h ^= h >> 33; /* step 6: avalanche */
    h *= XP2;
    h ^= h >> 29;
    h *= XP3;
    h ^= h >> 32;
    return h;
}

/* ======================================================================= frames */

/* frame.go:23-127, framedecompressor.go:130-150,306-374 */
int orc_parse_frame_header(const uint8_t *src, size_t n, orc_frame_header *h)
{
    memset(h, 0, sizeof(*h));
    if (n < 5) return ORC_ERR_SRC_TRUNCATED;
    if (!(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) return ORC_ERR_MAGIC;
    uint8_t d = src[4];
    h->single_segment = (d >> 5) & 1;        /* frame.go:101-103 */
    h->checksum_flag = (d >> 2) & 1;         /* :106-108 */
    static const int dict_sizes[4] = {0, 1, 2, 4}; /* :113-127 */
    h->dict_id_bytes = dict_sizes[d & 3];
    int fcsflag = d >> 6; /* :79-98 */
    h->fcs_bytes = fcsflag == 0 ? (h->single_segment ? 1 : 0) : (fcsflag == 1 ? 2 : (fcsflag == 2 ? 4 : 8));
    size_t p = 5;
    size_t need = (h->single_segment ? 0 : 1) + (size_t)h->dict_id_bytes + (size_t)h->fcs_bytes;
    if (n < p + need) return ORC_ERR_SRC_TRUNCATED;
    if (!h->single_segment) { /* frame.go:28-36 */
        uint8_t b = src[p++];
        int exp = b >> 3;
        uint64_t mant = b & 7;
        uint64_t base = (uint64_t)1 << (10 + exp);
        h->window_size = base + (base / 8) * mant;
    }
    p += (size_t)h->dict_id_bytes; /* value read, ignored (no dictionary support) */
    if (h->fcs_bytes > 0) {        /* frame.go:49-61 */
        uint64_t v = 0;
        for (int i = 0; i < h->fcs_bytes; i++) v |= (uint64_t)src[p + i] << (8 * i);
        if (h->fcs_bytes == 2) v += 256;
        h->content_size = v;
        p += (size_t)h->fcs_bytes;
        if (h->single_segment) h->window_size = h->content_size; /* framedecompressor.go:358-360 */
    }
    h->header_bytes = (int)p;
    return ORC_OK;
}

void orc_trace_free(orc_trace *t)
{
    free(t->blocks);
    free(t->literals);
    free(t->seqs);
    free(t->resolved_offsets);
    memset(t, 0, sizeof(*t));
}

static orc_block_info *trace_new_block(orc_trace *t)
{
    if (!t) return NULL;
    if (t->n_blocks == t->cap_blocks) {
        int nc = t->cap_blocks ? t->cap_blocks * 2 : 16;
        t->blocks = (orc_block_info *)realloc(t->blocks, (size_t)nc * sizeof(orc_block_info));
        t->cap_blocks = nc;
    }
    orc_block_info *b = &t->blocks[t->n_blocks++];
    memset(b, 0, sizeof(*b));
    return b;
}
static void trace_add_literals(orc_trace *t, const uint8_t *p, size_t n)
{
    if (!t) return;
    if (t->n_literals + n > t->cap_literals) {
        size_t nc = t->cap_literals ? t->cap_literals * 2 : 1 << 16;
        while (nc < t->n_literals + n) nc *= 2;
        t->literals = (uint8_t *)realloc(t->literals, nc);
        t->cap_literals = nc;
    }
    memcpy(t->literals + t->n_literals, p, n);
    t->n_literals += n;
}
static void trace_add_seq(orc_trace *t, const orc_sequence *s, int64_t resolved)
{
    if (!t) return;
    if (t->n_seqs == t->cap_seqs) {
        size_t nc = t->cap_seqs ? t->cap_seqs * 2 : 1 << 12;
        t->seqs = (orc_sequence *)realloc(t->seqs, nc * sizeof(orc_sequence));
        t->resolved_offsets = (int64_t *)realloc(t->resolved_offsets, nc * sizeof(int64_t));
        t->cap_seqs = nc;
    }
    t->seqs[t->n_seqs] = *s;
    t->resolved_offsets[t->n_seqs] = resolved;
    t->n_seqs++;
}

/* Per-frame decoder state carried across blocks (framedecompressor.go:14-37,
 * :283-294: "previous" tables = last table actually used, per kind). */
typedef struct {
    orc_huf_table huf;
    int have_huf;
    orc_fse_table ll, of, ml;
    int have_ll, have_of, have_ml;
    int64_t hist[3];
    uint8_t *lit_buf;     /* 128 KiB (framedecompressor.go:29) */
    orc_sequence *seq_buf;
    int seq_cap;
} frame_state;

enum { MODE_PREDEF = 0, MODE_RLE = 1, MODE_COMPRESSED = 2, MODE_REPEAT = 3 };

/* sequences.go:275-366 for one of the three tables.  Returns bytes used or -err. */
static int decode_one_seq_table(orc_fse_table *t, int *have, int mode, int which, const uint8_t *src,
                                int64_t len)
{
    const int32_t *tr = which == 0 ? LL_BASE : (which == 2 ? ML_BASE : NULL);
    const uint8_t *ex = which == 0 ? LL_EXTRA : (which == 2 ? ML_EXTRA : NULL);
    int ntr = which == 0 ? 36 : (which == 2 ? 53 : 0);
    switch (mode) {
    case MODE_PREDEF: {
        if (*have) orc_fse_free(t);
        int rc = orc_fse_build_predefined(t, which);
        if (rc) return -rc;
        *have = 1;
        return 0;
    }
    case MODE_RLE: {
        if (len < 1) return -ORC_ERR_SRC_TRUNCATED;
        uint8_t b = src[0];
        if (*have) orc_fse_free(t);
        memset(t, 0, sizeof(*t));
        t->is_rle = 1;
        if (which == 1) { /* sequences.go:322-323 */
            t->rle_value = b;
            t->rle_additional_bits = 0;
        } else {
            if (b >= ntr) return -ORC_ERR_FSE_TABLE; /* Go: index out of range */
            t->rle_value = tr[b];
            t->rle_additional_bits = ex[b];
        }
        *have = 1;
        return 1;
    }
    case MODE_REPEAT:
        if (!*have) return -ORC_ERR_NO_PREV_TABLE;
        return 0;
    default: {
        orc_fse_table nt;
        int used = orc_fse_read_description(&nt, src, len);
        if (used < 0) return used;
        int rc = orc_fse_build(&nt, tr, ntr, ex, ntr);
        if (rc) { orc_fse_free(&nt); return -rc; }
        if (*have) orc_fse_free(t);
        *t = nt;
        *have = 1;
        return used;
    }
    }
}

/* One compressed block: literals (literals.go:209-373), sequences
 * (sequences.go:371-450), execution (sequence_execution.go:14-63). */
static int decode_compressed_block(frame_state *fs, const uint8_t *src, uint32_t bsize, uint8_t *dst,
                                   size_t cap, size_t *out_pos, orc_trace *trace, orc_block_info *bi)
{
    if (bsize < 1) return ORC_ERR_SRC_TRUNCATED;
    /* ---- literals section header: literals.go:67-204 */
    uint8_t b0 = src[0];
    int lit_type = b0 & 3;
    int sf = (b0 >> 2) & 3;
    uint32_t regen = 0, csize = 0;
    int nstreams = 1, hdr = 0;
    if (lit_type == 0 || lit_type == 1) {
        if (sf == 0 || sf == 2) { hdr = 1; regen = b0 >> 3; }
        else if (sf == 1) { hdr = 2; if (bsize < 2) return ORC_ERR_SRC_TRUNCATED; regen = (uint32_t)(b0 >> 4) + ((uint32_t)src[1] << 4); }
        else { hdr = 3; if (bsize < 3) return ORC_ERR_SRC_TRUNCATED; regen = (uint32_t)(b0 >> 4) + ((uint32_t)src[1] << 4) + ((uint32_t)src[2] << 12); }
        csize = lit_type == 0 ? regen : 1;
    } else {
        hdr = sf <= 1 ? 3 : (sf == 2 ? 4 : 5);
        if (bsize < (uint32_t)hdr) return ORC_ERR_SRC_TRUNCATED;
        uint8_t c[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4 && i < hdr; i++) c[i] = src[i];
        uint32_t sizes = ((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24)) >> 4;
        nstreams = 4;
        if (sf == 0) nstreams = 1;
        if (sf <= 1) { regen = sizes & 0x3FF; csize = (sizes >> 10) & 0x3FF; }
        else if (sf == 2) { regen = sizes & 0x3FFF; csize = (sizes >> 14) & 0x3FFF; }
        else { regen = sizes & 0x3FFFF; csize = ((sizes >> 18) & 0x3FFFF) + ((uint32_t)src[4] << 10); }
    }
    if (regen > 128 * 1024) return ORC_ERR_CORRUPT_SIZES;
    size_t p = (size_t)hdr;
    if (lit_type == 3 && !fs->have_huf) return ORC_ERR_NO_PREV_TABLE; /* literals.go:247-252 */
    int tree_bytes = 0;
    if (lit_type == 2) { /* :254-267 */
        uint8_t weights[256];
        int nw = 0;
        int used = orc_huf_read_weights(src + p, (int64_t)bsize - (int64_t)p, weights, &nw);
        if (used < 0) return -used;
        int rc = orc_huf_build(&fs->huf, weights, nw);
        if (rc) return rc;
        fs->have_huf = 1;
        if ((uint32_t)used > csize) return ORC_ERR_CORRUPT_SIZES;
        csize -= (uint32_t)used;
        tree_bytes = used;
        p += (size_t)used;
    }
    uint32_t ss[4] = {0, 0, 0, 0};
    if (nstreams == 4) { /* :270-279 jump table */
        if (p + 6 > bsize || csize < 6) return ORC_ERR_SRC_TRUNCATED;
        ss[0] = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8);
        ss[1] = (uint32_t)src[p + 2] | ((uint32_t)src[p + 3] << 8);
        ss[2] = (uint32_t)src[p + 4] | ((uint32_t)src[p + 5] << 8);
        p += 6;
        csize -= 6;
        if (ss[0] + ss[1] + ss[2] > csize) return ORC_ERR_CORRUPT_SIZES; /* literals.go:54-56 */
        ss[3] = csize - ss[0] - ss[1] - ss[2];                            /* :60-62 */
    }
    if (p + csize > bsize) return ORC_ERR_SRC_TRUNCATED;
    const uint8_t *cdata = src + p;
    const uint8_t *lit = NULL; /* literal bytes (raw / huffman) */
    uint8_t rle_byte = 0;
    if (lit_type == 0) lit = cdata;            /* :290-292 */
    else if (lit_type == 1) rle_byte = cdata[0];
    else { /* :295-371 */
        if (nstreams == 1) {
            int64_t n = orc_huf_decode_stream(&fs->huf, cdata, csize, fs->lit_buf, regen);
            if (n < 0) return (int)-n;
            /* 1-stream: the reference does not compare the count (:299-304) and would
             * serve stale buffer bytes on a short decode; the oracle reports it. */
            if (n != (int64_t)regen) return ORC_ERR_HUF_LENGTH;
        } else {
            uint32_t normal = (regen + 3) / 4; /* :306-307 */
            if (3 * normal > regen) return ORC_ERR_HUF_LENGTH;
            uint32_t last = regen - 3 * normal;
            uint32_t low = 0;
            int64_t tot = 0;
            for (int k = 0; k < 4; k++) {
                uint32_t want = k < 3 ? normal : last;
                int64_t n = orc_huf_decode_stream(&fs->huf, cdata + low, ss[k], fs->lit_buf + (size_t)k * normal, want);
                if (n < 0) return (int)-n;
                if (k < 3 && n != (int64_t)normal) return ORC_ERR_HUF_LENGTH; /* :320,332,349 */
                tot += n;
                low += ss[k];
            }
            if (tot != (int64_t)regen) return ORC_ERR_HUF_LENGTH; /* :366-369 */
        }
        lit = fs->lit_buf;
    }
    p += csize;
    if (bi) {
        bi->lit_type = lit_type;
        bi->lit_regen = regen;
        bi->lit_compressed = csize;
        bi->lit_streams = nstreams;
        bi->huf_max_bits = (lit_type >= 2) ? fs->huf.max_bits : 0;
    }
    if (trace) {
        if (lit_type == 1) {
            for (uint32_t i = 0; i < regen; i++) trace_add_literals(trace, &rle_byte, 1);
        } else {
            trace_add_literals(trace, lit, regen);
        }
    }
    (void)tree_bytes;

    /* ---- sequences section: sequences.go:371-450 */
    if (p >= bsize) return ORC_ERR_SRC_TRUNCATED;
    int n_seq = 0;
    uint8_t s0 = src[p];
    if (s0 == 0) { /* :395-400 */
        p += 1;
        n_seq = 0;
    } else {
        if (s0 < 128) { n_seq = s0; p += 1; }
        else if (s0 < 255) { if (p + 2 > bsize) return ORC_ERR_SRC_TRUNCATED; n_seq = ((int)(s0 - 128) << 8) + src[p + 1]; p += 2; }
        else { if (p + 3 > bsize) return ORC_ERR_SRC_TRUNCATED; n_seq = (int)src[p + 1] + ((int)src[p + 2] << 8) + 0x7F00; p += 3; }
        if (p >= bsize) return ORC_ERR_SRC_TRUNCATED;
        uint8_t modes = src[p++]; /* :406-412 */
        int llm = (modes >> 6) & 3, ofm = (modes >> 4) & 3, mlm = (modes >> 2) & 3; /* :228-232 */
        if (bi) { bi->ll_mode = llm; bi->of_mode = ofm; bi->ml_mode = mlm; }
        int used;
        used = decode_one_seq_table(&fs->ll, &fs->have_ll, llm, 0, src + p, (int64_t)bsize - (int64_t)p);
        if (used < 0) return -used;
        p += (size_t)used;
        used = decode_one_seq_table(&fs->of, &fs->have_of, ofm, 1, src + p, (int64_t)bsize - (int64_t)p);
        if (used < 0) return -used;
        p += (size_t)used;
        used = decode_one_seq_table(&fs->ml, &fs->have_ml, mlm, 2, src + p, (int64_t)bsize - (int64_t)p);
        if (used < 0) return -used;
        p += (size_t)used;
        if (p > bsize) return ORC_ERR_SRC_TRUNCATED;
        if (n_seq == 0) {
            /* b0 in 128..255 can encode 0 sequences; the reference then decodes an
             * empty list but still requires the bitstream cursor to end at -1. */
        }
        if (n_seq > fs->seq_cap) {
            free(fs->seq_buf);
            fs->seq_cap = n_seq + 1024;
            fs->seq_buf = (orc_sequence *)malloc(sizeof(orc_sequence) * (size_t)fs->seq_cap);
            if (!fs->seq_buf) { fs->seq_cap = 0; return ORC_ERR_UNSUPPORTED; }
        }
        int rc = orc_decode_sequences(&fs->ll, &fs->of, &fs->ml, src + p, (int64_t)bsize - (int64_t)p, n_seq, fs->seq_buf);
        if (rc) return rc;
        p = bsize;
    }
    if (p != bsize) return ORC_ERR_CORRUPT_SIZES; /* framedecompressor.go:114-123 */
    if (bi) bi->n_seq = n_seq;

    /* ---- execution: sequence_execution.go:14-63 on a flat output */
    size_t op = *out_pos;
    uint32_t lit_read = 0;
    for (int i = 0; i < n_seq; i++) {
        const orc_sequence *s = &fs->seq_buf[i];
        if (s->literal_length > 0) { /* :19-34 */
            uint32_t ll = (uint32_t)s->literal_length;
            if (lit_type != 1 && lit_read + ll > regen) return ORC_ERR_LITERALS;
            if (op + ll > cap) return ORC_ERR_DST_FULL;
            if (lit_type == 1) memset(dst + op, rle_byte, ll); /* literals.go:390-396 */
            else memcpy(dst + op, lit + lit_read, ll);
            lit_read += ll;
            op += ll;
        }
        int64_t off = orc_next_offset(fs->hist, s->offset, s->literal_length); /* :43 */
        trace_add_seq(trace, s, off);
        if (s->match_length > 0) { /* :44-49 */
            uint32_t ml = (uint32_t)s->match_length;
            if (off <= 0 || (uint64_t)off > op) return ORC_ERR_OFFSET;
            if (op + ml > cap) return ORC_ERR_DST_FULL;
            if ((uint64_t)off >= ml) memcpy(dst + op, dst + op - (size_t)off, ml);
            else for (uint32_t j = 0; j < ml; j++) dst[op + j] = dst[op + j - (size_t)off];
            op += ml;
        }
    }
    /* :55-59 rest of the literals */
    if (lit_type == 1) {
        if (lit_read > regen) return ORC_ERR_LITERALS; /* Go: negative make() panic */
    }
    uint32_t rest = regen - lit_read;
    if (op + rest > cap) return ORC_ERR_DST_FULL;
    if (lit_type == 1) memset(dst + op, rle_byte, rest);
    else if (rest) memcpy(dst + op, lit + lit_read, rest);
    op += rest;
    *out_pos = op;
    return ORC_OK;
}

static void frame_state_free(frame_state *fs)
{
    if (fs->have_ll) orc_fse_free(&fs->ll);
    if (fs->have_of) orc_fse_free(&fs->of);
    if (fs->have_ml) orc_fse_free(&fs->ml);
    free(fs->lit_buf);
    free(fs->seq_buf);
}

/* One frame with caller-provided scratch (`fs` zeroed or left over from a previous frame: the
 * literal and sequence buffers are reused, tables are reset per frame as framedecompressor.go:42-52
 * resets them). */
static int decode_frame_with(frame_state *fs, const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                             size_t *out_len, size_t *consumed, orc_trace *trace)
{
    orc_frame_header h;
    int rc = orc_parse_frame_header(src, n, &h);
    if (rc) return rc;
    if (fs->have_ll) orc_fse_free(&fs->ll);
    if (fs->have_of) orc_fse_free(&fs->of);
    if (fs->have_ml) orc_fse_free(&fs->ml);
    fs->have_ll = fs->have_of = fs->have_ml = fs->have_huf = 0;
    fs->hist[0] = 1; /* framedecompressor.go:48,59 */
    fs->hist[1] = 4;
    fs->hist[2] = 8;
    if (!fs->lit_buf) fs->lit_buf = (uint8_t *)malloc(128 * 1024 + 64);
    if (!fs->lit_buf) return ORC_ERR_UNSUPPORTED;
    size_t p = (size_t)h.header_bytes;
    size_t op = 0;
    int last = 0;
    rc = ORC_OK;
    while (!last) { /* framedecompressor.go:246-254 */
        if (p + 3 > n) { rc = ORC_ERR_SRC_TRUNCATED; break; }
        uint8_t r0 = src[p], r1 = src[p + 1], r2 = src[p + 2]; /* block.go:33-55 */
        p += 3;
        last = r0 & 1;
        int type = (r0 >> 1) & 3;
        uint32_t size = (uint32_t)(r0 >> 3) + ((uint32_t)r1 << 5) + ((uint32_t)r2 << 13);
        if (type >= 3) { rc = ORC_ERR_BLOCK_TYPE; break; }
        if (size > 128 * 1024) { rc = ORC_ERR_BLOCK_SIZE; break; }
        orc_block_info *bi = trace_new_block(trace);
        if (bi) { bi->block_type = type; bi->block_size = size; bi->out_begin = op; }
        if (type == 0) { /* framedecompressor.go:211-215 */
            if (p + size > n) { rc = ORC_ERR_SRC_TRUNCATED; break; }
            if (op + size > cap) { rc = ORC_ERR_DST_FULL; break; }
            memcpy(dst + op, src + p, size);
            op += size;
            p += size;
        } else if (type == 1) { /* :229-241 */
            if (p + 1 > n) { rc = ORC_ERR_SRC_TRUNCATED; break; }
            if (op + size > cap) { rc = ORC_ERR_DST_FULL; break; }
            memset(dst + op, src[p], size);
            op += size;
            p += 1;
        } else { /* :217-228 */
            if (p + size > n) { rc = ORC_ERR_SRC_TRUNCATED; break; }
            rc = decode_compressed_block(fs, src + p, size, dst, cap, &op, trace, bi);
            if (rc) break;
            p += size;
        }
        if (bi) bi->out_end = op;
    }
    if (out_len) *out_len = op;
    if (consumed) *consumed = p;
    return rc;
}

int orc_decode_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_len,
                     size_t *consumed, orc_trace *trace)
{
    frame_state *fs = (frame_state *)calloc(1, sizeof(frame_state));
    if (!fs) return ORC_ERR_UNSUPPORTED;
    int rc = decode_frame_with(fs, src, n, dst, cap, out_len, consumed, trace);
    frame_state_free(fs);
    free(fs);
    return rc;
}

int orc_decode_frames(const uint8_t *blob, const uint64_t *frame_off, const uint64_t *frame_len,
                      int n_frames, uint8_t *dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                      uint64_t *out_len, int32_t *status)
{
    int first = 0;
    /* scratch reused across the frames of the call (the per-frame >128 KiB mallocs would be served by
     * mmap/munmap, which serialises concurrent callers on the process-wide mapping lock) */
    frame_state *fs = (frame_state *)calloc(1, sizeof(frame_state));
    if (!fs) return ORC_ERR_UNSUPPORTED;
    for (int f = 0; f < n_frames; f++) {
        size_t ol = 0;
        int rc = decode_frame_with(fs, blob + frame_off[f], (size_t)frame_len[f], dst + dst_off[f],
                                   (size_t)dst_cap[f], &ol, NULL, NULL);
        if (out_len) out_len[f] = ol;
        if (status) status[f] = rc;
        if (rc && !first) first = rc;
    }
    frame_state_free(fs);
    free(fs);
    return first;
}

int orc_decode_frames_wsum(const uint8_t *blob, const uint64_t *frame_off, const uint64_t *frame_len,
                           int n_frames, uint8_t *dst, const uint64_t *dst_off, const uint64_t *dst_cap,
                           uint64_t *out_len, int32_t *status, uint64_t *wsum)
{
    int first = 0;
    frame_state *fs = (frame_state *)calloc(1, sizeof(frame_state));
    if (!fs) return ORC_ERR_UNSUPPORTED;
    for (int f = 0; f < n_frames; f++) {
        size_t ol = 0;
        uint8_t *d = dst + dst_off[f];
        int rc = decode_frame_with(fs, blob + frame_off[f], (size_t)frame_len[f], d, (size_t)dst_cap[f], &ol, NULL, NULL);
        if (out_len) out_len[f] = ol;
        if (status) status[f] = rc;
        if (rc && !first) first = rc;
        if (wsum) {
            uint64_t acc = 0, j = 0;
            size_t i = 0;
            for (; i + 8 <= ol; i += 8, j++) {
                uint64_t w;
                memcpy(&w, d + i, 8);
                acc += w * (2 * j + 1);
            }
            if (i < ol) {
                uint64_t w = 0;
                memcpy(&w, d + i, ol - i);
                acc += w * (2 * j + 1);
            }
            wsum[f] = acc;
        }
    }
    frame_state_free(fs);
    free(fs);
    return first;
}

const char *orc_strerror(int code)
{
    switch (code) {
    case ORC_OK: return "ok";
    case ORC_ERR_SRC_TRUNCATED: return "source truncated";
    case ORC_ERR_MAGIC: return "Magicnum is not correct";
    case ORC_ERR_BLOCK_TYPE: return "Illegal BlockType";
    case ORC_ERR_BLOCK_SIZE: return "Illegal block-size";
    case ORC_ERR_FSE_TABLE: return "bad FSE table description";
    case ORC_ERR_HUF_WEIGHTS: return "bad huffman weights";
    case ORC_ERR_NO_PREV_TABLE: return "no previous table to carry over";
    case ORC_ERR_BAD_PADDING: return "bad padding";
    case ORC_ERR_HUF_BITS: return "huffman stream: bits left over / over-read";
    case ORC_ERR_HUF_LENGTH: return "huffman stream decoded to wrong length";
    case ORC_ERR_SEQ_BITS: return "sequence bitstream: not all bits used";
    case ORC_ERR_CORRUPT_SIZES: return "section sizes do not add up";
    case ORC_ERR_LITERALS: return "not enough literal bytes";
    case ORC_ERR_OFFSET: return "cannot repeat bytes from before the first one";
    case ORC_ERR_DST_FULL: return "destination capacity exceeded";
    default: return "unsupported";
    }
}
