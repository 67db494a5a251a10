// mzd_parse.hip -- planning ON THE DEVICE: frame / block / section header parsing,
// FSE table descriptions, Huffman weights and the derivation of the work lists, one LANE per frame.
// It is the device restatement of what planner.cpp (host) + mzd_batch_upload do together, and
// follows the same reference lines:
//   frame header      structure/frame.go:23-127, decompression/framedecompressor.go:130-150,306-374
//   block header      structure/block.go:33-55
//   literals header   structure/literals.go:67-204,209-289, jump table :46-62
//   huffman weights   structure/huffman.go:40-107 (+ fse/fse.go:307-390), MaxBits :112-131
//   sequences header  structure/sequences.go:228-269,371-433
//   table selection   structure/sequences.go:275-366, carry-over framedecompressor.go:283-294
//   FSE description   fse/fse.go:28-130
// Round 5: the unit of the walk is a UNIT, not a frame -- a whole frame (the many-frames case: a lane per frame), or, for a LARGE
// frame, one of its blocks: k_parse_index walks such a frame's 3-byte block headers (one lane, the only serial part) and lists
// where every block starts; the blocks are then parsed side by side, a lane each.  What a block inherits from the blocks before
// it -- the Huffman table and the three FSE tables a Treeless / Repeat_Mode section reuses, and whether an earlier block had
// sequences (framedecompressor.go:283-294, literals.go:247-252, sequences.go:275-366) -- is not known to its lane in pass 0: the
// lane notes that it NEEDS it (UnitCount::need) and what it leaves behind (UnitCount::carry); the host, which turns the counts
// into offsets anyway, walks the units of the frame in order and hands every unit its inheritance for pass 1 (UnitBase::in).
// Two passes over the same walk: PASS 0 counts what every frame contributes (blocks, tasks, table
// cells, scratch, output bound) and finds parse errors; the host turns the counts into offsets;
// PASS 1 walks again and writes DFrame / DBlock / HufTask / SeqTask and the table build
// descriptors (counts / weights; k_fse_build and k_huf_build then build the tables).  Included by
// mzd_api.hip after mzd_kernels.hip.
#pragma once

namespace mzd {

struct ParseScratch {  // per lane, in global memory
    int16_t prob[256];
    uint8_t w[256];
    uint32_t cells[512];  // weight-stream FSE table: baseline | nbits << 16 | symbol << 24
    uint16_t nextv[256];
    uint8_t sym[512];
};

// a table a later section may reuse: src 0 = none, 1 = made by this unit (off = offset among the unit's own device cells),
// 2 = the predefined table of its kind, 3 (pass 1 only) = absolute (off = device cell offset)
struct PCarry {
    uint32_t off;
    uint8_t log, src, pad[2];
};
struct FrameCount {
    int32_t status;
    uint32_t n_blocks, n_seq, n_hufb, n_fse_tab, n_fse_src, n_fse_dev, n_huf_tab, n_huf_src, n_huf_dev, n_tile;
    uint32_t max_huf_bits, checksum, flags;
    uint32_t max_seq_logs;  // largest accuracy logs of the frame's sequence tables: LL | ML << 8 | OF << 16
    uint32_t n_raw, n_rle, n_comp, n_huf_streams;
    uint64_t n_rec, lit_bytes, out_bound, content_size, comp_bytes;
    // units of a large frame: what a later unit may reuse of this one's, which inheritance this unit could not do without
    // (bit k: carry kind k), whether its walk saw the frame's last block
    PCarry carry[4];
    uint32_t need, saw_last;
};
struct FrameBase {
    uint32_t block0, seq0, hufb0, fse_tab0, fse_src0, fse_dev0, huf_tab0, huf_src0, huf_dev0, tile0;
    uint64_t rec0, lit0, out_off, out_cap;
    // units of a large frame (pass 1): what the unit inherits -- [0] Huffman, [1 + MZD_FSE_*] the sequence tables --, whether an
    // earlier block of the frame had sequences, and for the frame's FIRST unit the totals of the frame it writes the DFrame from
    PCarry in[4];
    uint32_t seen_seq, frame_blocks, frame_block0;
    uint32_t pad;  // MZD_* status of the unit's frame (pass 1 skips the units of a frame that failed)
};
// one unit of the walk: a whole frame, or one block of a large frame
constexpr uint32_t kUnitFirst = 1;   // starts at the frame's magic number (parses the frame header)
constexpr uint32_t kUnitFinal = 2;   // runs to the frame's last block (and reads the checksum behind it)
struct ParseUnit {
    uint64_t begin, end;   // bytes of the blob: where the unit starts, where its frame ends
    uint32_t frame;        // index of its frame in the batch
    uint32_t flags;        // kUnit*
    uint32_t max_blocks;   // blocks this unit walks at most (a unit that is not final stops there)
    uint32_t pad;
};
struct ParseOut {
    DFrame *frames;
    DBlock *blocks;
    HufTask *huf_tasks;
    SeqTask *seq_tasks;
    FseBuildDesc *fse_tabs;
    uint32_t *fse_src;
    HufBuildDesc *huf_tabs;
    uint16_t *huf_src;
    uint32_t predef_off[3];  // device cell offset of the predefined LL / OF / ML tables (indexed by MZD_FSE_*)
};

__device__ __forceinline__ int p_highbit(uint32_t v) { return v ? 31 - __builtin_clz(v) : 0; }  // fse.go:235-249

// forward (LSB-first) bit reader: bitstream/bitstream.go:39-90
struct PFwd {
    const uint8_t *p;
    uint64_t nbits, pos;
    bool overrun;
    __device__ PFwd(const uint8_t *d, uint64_t len) : p(d), nbits(len * 8), pos(0), overrun(false) {}
    __device__ uint32_t read(int n)
    {
        if (pos + (uint64_t)n > nbits) {
            overrun = true;
            return 0;
        }
        uint32_t v = 0;
        for (int i = 0; i < n; i++, pos++) v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i;
        return v;
    }
};
// backward bit reader of the Huffman-weight stream: reversebitstream.go:17-88
struct PRev {
    const uint8_t *p;
    int64_t cursor;
    __device__ PRev(const uint8_t *d, int64_t len) : p(d), cursor(len * 8 - 1) {}
    __device__ uint32_t read(int n)
    {
        uint32_t v = 0;
        for (int i = 0; i < n; i++) {
            const int64_t b = cursor - i;
            v = (v << 1) | (b >= 0 ? (uint32_t)((p[b >> 3] >> (b & 7)) & 1u) : 0u);
        }
        cursor -= n;
        return v;
    }
};

// fse.go:28-130.  Returns bytes used, or <0: -MZD_ERR_*.
__device__ int p_read_fse_description(const uint8_t *src, uint64_t len, int16_t *prob, int &nsym, int &acc_log)
{
    PFwd bs(src, len);
    acc_log = 5 + (int)bs.read(4);
    nsym = 0;
    if (bs.overrun) return -MZD_ERR_TRUNCATED;
    if (acc_log > 9) return -MZD_ERR_UNSUPPORTED;
    int32_t remaining = 1 << acc_log;
    while (remaining > 0) {
        const int nb = p_highbit((uint32_t)remaining + 1) + 1;
        uint32_t v = bs.read(nb);
        if (bs.overrun) return -MZD_ERR_TRUNCATED;
        const uint32_t lower = (1u << (nb - 1)) - 1;
        const uint32_t thresh = (1u << nb) - 1 - (uint32_t)(remaining + 1);
        if ((v & lower) < thresh) {
            v &= lower;
            bs.pos--;  // "small" value: it used one bit less (fse.go:65-77)
        } else if (v > lower) {
            v -= thresh;
        }
        const int pr = (int)v - 1;
        if (nsym >= 256) return -MZD_ERR_FSE_TABLE;
        prob[nsym++] = (int16_t)pr;
        remaining -= pr < 0 ? 1 : pr;
        if (pr == 0) {  // zero-probability run lengths, 2 bits at a time (fse.go:96-117)
            uint32_t rep = 3;
            while (rep == 3) {
                rep = bs.read(2);
                if (bs.overrun) return -MZD_ERR_TRUNCATED;
                for (uint32_t i = 0; i < rep; i++) {
                    if (nsym >= 256) return -MZD_ERR_FSE_TABLE;
                    prob[nsym++] = 0;
                }
            }
        }
    }
    if (remaining != 0) return -MZD_ERR_FSE_TABLE;  // fse.go:126-128
    return (int)((bs.pos + 7) / 8);
}

// fse.go:136-230 for the (small) weight-stream table; the counts add up to the table size, so the
// spread cannot fail (same argument as k_fse_build)
__device__ void p_build_fse_cells(ParseScratch &sc, int nsym, int acc_log)
{
    const int size = 1 << acc_log;
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (sc.prob[s] == -1) {
            sc.sym[max(high, 0)] = (uint8_t)s;
            high--;
            sc.nextv[s] = 1;
        } else {
            sc.nextv[s] = (uint16_t)sc.prob[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < sc.prob[s]; i++) {
            sc.sym[pos] = (uint8_t)s;
            int guard = 0;
            do {
                pos = (pos + step) & mask;
            } while (pos > high && ++guard <= size);
        }
    }
    for (int i = 0; i < size; i++) {
        const uint32_t s = sc.sym[i];
        const uint32_t n = sc.nextv[s]++;
        const uint32_t nb = (uint32_t)acc_log - (uint32_t)p_highbit(n);
        sc.cells[i] = (((n << nb) - (uint32_t)size) & 0xFFFF) | (nb << 16) | (s << 24);
    }
}

// huffman.go:40-107: weights, 4-bit direct or FSE-compressed (two interleaved states, fse.go:307-390).
// Returns bytes used or <0.
__device__ int p_read_huffman_weights(const uint8_t *src, uint64_t len, ParseScratch &sc, int &nw)
{
    nw = 0;
    if (len < 1) return -MZD_ERR_TRUNCATED;
    const int header = src[0];
    if (header >= 128) {
        const int n = header - 127;
        const int nbytes = (n + 1) / 2;
        if (1 + (uint64_t)nbytes > len) return -MZD_ERR_TRUNCATED;
        for (int i = 0; i < n; i++) {
            const uint8_t b = src[1 + i / 2];
            sc.w[nw++] = (i & 1) ? (uint8_t)(b & 15) : (uint8_t)(b >> 4);
        }
        return 1 + nbytes;
    }
    if (1 + (uint64_t)header > len) return -MZD_ERR_TRUNCATED;
    int nsym, acc_log;
    const int used = p_read_fse_description(src + 1, (uint64_t)header, sc.prob, nsym, acc_log);
    if (used < 0) return used;
    p_build_fse_cells(sc, nsym, acc_log);
    const int64_t slen = header - used;
    if (slen <= 0) return -MZD_ERR_TRUNCATED;
    PRev rb(src + 1 + used, slen);
    int pad = 0;
    while (rb.read(1) == 0)
        if (++pad >= 8) return -MZD_ERR_BAD_PADDING;  // fse.go:314-325
    uint32_t st[2];
    st[0] = rb.read(acc_log);
    st[1] = rb.read(acc_log);
    for (int turn = 0;; turn ^= 1) {
        const uint32_t e = sc.cells[st[turn]];
        if (nw >= 255) return -MZD_ERR_HUF_WEIGHTS;
        sc.w[nw++] = (uint8_t)(e >> 24);
        st[turn] = (e & 0xFFFF) + rb.read((int)((e >> 16) & 0xFF));
        if (rb.cursor < -1) {  // over-read: flush the other state's symbol and stop (fse.go:363-383)
            if (nw >= 255) return -MZD_ERR_HUF_WEIGHTS;
            sc.w[nw++] = (uint8_t)(sc.cells[st[turn ^ 1]] >> 24);
            break;
        }
    }
    return 1 + header;
}

// huffman.go:112-131
__device__ int p_huffman_max_bits(const uint8_t *w, int nw, int &max_bits)
{
    uint32_t sum = 0;
    for (int i = 0; i < nw; i++) {
        if (w[i] > 11) return MZD_ERR_HUF_WEIGHTS;
        if (w[i]) sum += 1u << (w[i] - 1);
    }
    if (sum == 0) return MZD_ERR_HUF_WEIGHTS;
    max_bits = p_highbit(sum) + 1;
    const uint32_t left = (1u << max_bits) - sum;
    if (left & (left - 1)) return MZD_ERR_HUF_WEIGHTS;  // huffman.go:128-130
    if (max_bits > 11) return MZD_ERR_UNSUPPORTED;
    return MZD_OK;
}

struct PTabRef {  // "previous" table of one kind (framedecompressor.go:283-294): where its cells are
    uint32_t off;
    uint32_t log;  // accuracy log (FSE) / MaxBits (Huffman)
    bool valid;
};

// Walks one unit: a frame, or -- units of a large frame -- its blocks from `begin` on, `max_blocks` of them at most.  PASS 0: fills
// `cnt` (status + counts, and for the units of a large frame what the unit needs from / leaves to its neighbours).  PASS 1: `fb`
// holds the unit's offsets (and its inheritance), descriptors are written through `po` (`cnt` is not touched).
template <int PASS>
__device__ void p_walk_frame(const uint8_t *base, uint64_t begin, uint64_t end, ParseScratch &sc, FrameCount &cnt,
                             const FrameBase &fb, const ParseOut &po, uint32_t uflags = kUnitFirst | kUnitFinal,
                             uint32_t max_blocks = 0xFFFFFFFFu)
{
    const int kPMaxSym[3] = {35, 31, 52}, kPMaxLog[3] = {9, 8, 9}, kPDefLog[3] = {6, 5, 6};  // by MZD_FSE_*: LL, OF, ML
    FrameCount c{};
    c.content_size = MZD_UNKNOWN_SIZE;
    uint64_t p = begin;
    const bool first_unit = (uflags & kUnitFirst) != 0;
#define P_FAIL(code)                               \
    do {                                           \
        if (PASS == 0) {                           \
            FrameCount z_{};                       \
            z_.status = (code);                    \
            z_.content_size = MZD_UNKNOWN_SIZE;    \
            z_.need = c.need;                      \
            cnt = z_;                              \
        }                                          \
        return;                                    \
    } while (0)
    uint8_t fhd = 0;
    if (first_unit) {
        // ---- magic + frame header
        if (end - p < 5) P_FAIL(MZD_ERR_TRUNCATED);
        if (!(base[p] == 0x28 && base[p + 1] == 0xB5 && base[p + 2] == 0x2F && base[p + 3] == 0xFD)) P_FAIL(MZD_ERR_MAGIC);
        fhd = base[p + 4];
        p += 5;
        const bool single = (fhd >> 5) & 1;
        const int dict_bytes = (fhd & 3) == 3 ? 4 : (fhd & 3);
        const int fcs_flag = fhd >> 6;
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        if (end - p < (uint64_t)(!single) + dict_bytes + fcs_bytes) P_FAIL(MZD_ERR_TRUNCATED);
        if (!single) p++;  // window descriptor (frame.go:28-36): informational
        p += dict_bytes;   // dictionary id is read and ignored (no dictionary support: Readme.md:59-62)
        if (fcs_bytes) {
            uint64_t v = 0;
            for (int i = 0; i < fcs_bytes; i++) v |= (uint64_t)base[p + i] << (8 * i);
            if (fcs_bytes == 2) v += 256;  // frame.go:58-60
            c.content_size = v;
            p += fcs_bytes;
        }
        c.flags |= ((fhd >> 2) & 1) ? 0x80000000u : 0u;  // (the checksum flag, for the unit that reaches the frame's end)
    }
    PTabRef prev_huf{0, 0, false}, prev_t[3] = {{0, 0, false}, {0, 0, false}, {0, 0, false}};
    bool last = false, seen_seq = false;
    if (!first_unit && PASS == 1) {
        // a later unit of a large frame: what it inherits (the host resolved it from the units before: UnitBase::in)
        seen_seq = fb.seen_seq != 0;
        if (fb.in[0].src) prev_huf = PTabRef{fb.in[0].off, fb.in[0].log, true};
        for (int k = 0; k < 3; k++)
            if (fb.in[1 + k].src) prev_t[k] = PTabRef{fb.in[1 + k].src == 2 ? po.predef_off[k] : fb.in[1 + k].off, fb.in[1 + k].log, true};
    }
    // ---- blocks (block.go:33-55, framedecompressor.go:198-303)
    while (!last && c.n_blocks < max_blocks) {
        if (end - p < 3) P_FAIL(MZD_ERR_TRUNCATED);
        const uint32_t h = base[p] | ((uint32_t)base[p + 1] << 8) | ((uint32_t)base[p + 2] << 16);
        p += 3;
        last = h & 1;
        const int type = (h >> 1) & 3;
        const uint32_t size = h >> 3;
        if (type == 3) P_FAIL(MZD_ERR_BLOCK_TYPE);
        if (size > kBlockMax) P_FAIL(MZD_ERR_BLOCK_SIZE);
        DBlock d{};
        d.type = (uint8_t)type;
        d.size = size;
        const uint32_t bi = fb.block0 + c.n_blocks;
        if (type == MZD_BLOCK_RAW) {
            if (end - p < size) P_FAIL(MZD_ERR_TRUNCATED);
            d.src_off = p;
            p += size;
            c.out_bound += size;
            c.comp_bytes += size;
            c.n_raw++;
        } else if (type == MZD_BLOCK_RLE) {
            if (end - p < 1) P_FAIL(MZD_ERR_TRUNCATED);
            d.src_off = p;
            p += 1;
            c.out_bound += size;
            c.comp_bytes += 1;
            c.n_rle++;
        } else {
            if (end - p < size) P_FAIL(MZD_ERR_TRUNCATED);
            c.n_comp++;
            c.out_bound += kBlockMax;
            const uint64_t lim = p + size;
            if (size < 1) P_FAIL(MZD_ERR_TRUNCATED);
            // ---- literals section header (literals.go:67-204)
            const uint8_t b0 = base[p];
            const int ltype = b0 & 3, sf = (b0 >> 2) & 3;
            uint32_t regen = 0, csize = 0;
            int hdr, streams = 1;
            if (ltype <= 1) {
                hdr = (sf == 1) ? 2 : (sf == 3 ? 3 : 1);
                if (p + hdr > lim) P_FAIL(MZD_ERR_TRUNCATED);
                if (hdr == 1) regen = b0 >> 3;
                else if (hdr == 2) regen = (b0 >> 4) + ((uint32_t)base[p + 1] << 4);
                else regen = (b0 >> 4) + ((uint32_t)base[p + 1] << 4) + ((uint32_t)base[p + 2] << 12);
                csize = ltype == 0 ? regen : 1;
            } else {
                hdr = sf <= 1 ? 3 : sf + 2;
                if (p + hdr > lim) P_FAIL(MZD_ERR_TRUNCATED);
                uint64_t v = 0;
                for (int i = 0; i < hdr; i++) v |= (uint64_t)base[p + i] << (8 * i);
                v >>= 4;
                const int bits = sf <= 1 ? 10 : (sf == 2 ? 14 : 18);
                regen = (uint32_t)(v & ((1u << bits) - 1));
                csize = (uint32_t)((v >> bits) & ((1u << bits) - 1));
                streams = sf == 0 ? 1 : 4;
            }
            if (regen > kBlockMax) P_FAIL(MZD_ERR_CORRUPT_SIZES);
            uint64_t q = p + hdr;
            d.lit_regen = regen;
            if (ltype == 0) {
                d.lit_type = MZD_LIT_RAW;
                if (q + regen > lim) P_FAIL(MZD_ERR_TRUNCATED);
                d.lit_src = q;
                q += regen;
                c.comp_bytes += regen;
            } else if (ltype == 1) {
                d.lit_type = MZD_LIT_RLE;
                if (q + 1 > lim) P_FAIL(MZD_ERR_TRUNCATED);
                d.lit_src = q;
                q += 1;
                c.comp_bytes += 1;
            } else {
                d.lit_type = MZD_LIT_HUF;
                if (q + csize > lim) P_FAIL(MZD_ERR_TRUNCATED);
                const uint64_t lit_end = q + csize;
                if (ltype == 3) {  // Treeless: literals.go:247-252
                    if (!prev_huf.valid) {
                        if (first_unit || PASS == 1) P_FAIL(MZD_ERR_NO_PREV_TABLE);
                        c.need |= 1u;  // (a later unit in pass 0: the table is an earlier unit's -- or nobody's: the host decides)
                        prev_huf.valid = true;
                    }
                } else {           // literals.go:254-267
                    int nw = 0;
                    const int used = p_read_huffman_weights(base + q, lit_end - q, sc, nw);
                    if (used < 0) P_FAIL(-used);
                    int mb = 0;
                    const int rc = p_huffman_max_bits(sc.w, nw, mb);
                    if (rc) P_FAIL(rc);
                    const uint32_t ncell = ((uint32_t)nw + 1) / 2;
                    if (PASS == 1) {
                        HufBuildDesc hd{};
                        hd.src_off = fb.huf_src0 + c.n_huf_src;
                        hd.dst_off = fb.huf_dev0 + c.n_huf_dev;
                        hd.max_bits = (uint8_t)mb;
                        hd.n_weights = (uint8_t)nw;
                        hd.ok = 1;
                        po.huf_tabs[fb.huf_tab0 + c.n_huf_tab] = hd;
                        for (uint32_t i = 0; i < ncell; i++)
                            po.huf_src[hd.src_off + i] =
                                (uint16_t)(sc.w[2 * i] | ((2 * i + 1 < (uint32_t)nw ? sc.w[2 * i + 1] : 0) << 8));
                    }
                    prev_huf = PTabRef{fb.huf_dev0 + c.n_huf_dev, (uint32_t)mb, true};
                    c.carry[0] = PCarry{c.n_huf_dev, (uint8_t)mb, 1, {0, 0}};
                    c.n_huf_tab++;
                    c.n_huf_src += ncell;
                    c.n_huf_dev += 1u << mb;
                    c.max_huf_bits = max(c.max_huf_bits, (uint32_t)mb);
                    q += (uint64_t)used;
                }
                uint32_t ssz[4] = {0, 0, 0, 0};
                if (streams == 4) {  // jump table: literals.go:46-62,270-279
                    if (q + 6 > lit_end) P_FAIL(MZD_ERR_TRUNCATED);
                    const uint32_t s1 = base[q] | (base[q + 1] << 8), s2 = base[q + 2] | (base[q + 3] << 8),
                                   s3 = base[q + 4] | (base[q + 5] << 8);
                    q += 6;
                    const uint64_t rest = lit_end - q;
                    if ((uint64_t)s1 + s2 + s3 > rest) P_FAIL(MZD_ERR_CORRUPT_SIZES);
                    ssz[0] = s1; ssz[1] = s2; ssz[2] = s3;
                    ssz[3] = (uint32_t)(rest - s1 - s2 - s3);
                    if (3 * ((regen + 3) / 4) > regen) P_FAIL(MZD_ERR_HUF_LENGTH);  // literals.go:306-307 would go negative
                } else {
                    ssz[0] = (uint32_t)(lit_end - q);
                }
                // the four stream tasks of the section (a 1-stream section uses lane 0 of its quad)
                const int ns = streams == 4 ? 4 : 1;
                const uint32_t normal = ns == 4 ? (regen + 3) / 4 : regen;
                d.lit_src = fb.lit0 + c.lit_bytes;
                if (PASS == 1) {
                    uint64_t ioff = q;
                    for (int sidx = 0; sidx < 4; sidx++) {
                        HufTask t{};
                        if (sidx < ns) {
                            t.in_off = ioff;
                            t.in_size = ssz[sidx];
                            t.out_off = fb.lit0 + c.lit_bytes + (uint64_t)sidx * normal;
                            t.out_size = ns == 4 ? (sidx < 3 ? normal : regen - 3 * normal) : regen;
                            ioff += t.in_size;
                        }
                        t.table_off = prev_huf.off;
                        t.max_bits = prev_huf.log;
                        t.block = bi;
                        po.huf_tasks[4 * (size_t)(fb.hufb0 + c.n_hufb) + sidx] = t;
                    }
                }
                c.n_hufb++;
                c.n_huf_streams += (uint32_t)ns;
                c.lit_bytes += ((uint64_t)regen + 15) & ~15ull;
                c.comp_bytes += lit_end - q;
                q = lit_end;
            }
            // ---- sequences section header (sequences.go:371-433)
            if (q >= lim) P_FAIL(MZD_ERR_TRUNCATED);
            const uint8_t s0 = base[q];
            if (s0 == 0) {  // sequences.go:395-400
                q += 1;
                if (q != lim) P_FAIL(MZD_ERR_CORRUPT_SIZES);  // framedecompressor.go:114-123
            } else {
                uint32_t nseq;
                if (s0 < 128) {
                    nseq = s0;
                    q += 1;
                } else if (s0 < 255) {
                    if (q + 2 > lim) P_FAIL(MZD_ERR_TRUNCATED);
                    nseq = ((uint32_t)(s0 - 128) << 8) + base[q + 1];
                    q += 2;
                } else {
                    if (q + 3 > lim) P_FAIL(MZD_ERR_TRUNCATED);
                    nseq = base[q + 1] + ((uint32_t)base[q + 2] << 8) + 0x7F00;
                    q += 3;
                }
                if (q >= lim) P_FAIL(MZD_ERR_TRUNCATED);
                const uint8_t modes = base[q++];  // sequences.go:228-232
                PTabRef use[3];
                for (int kidx = 0; kidx < 3; kidx++) {  // stream order: LL, OF, ML
                    const int kind = kidx;              // MZD_FSE_LL = 0, OF = 1, ML = 2
                    const int mode = (modes >> (6 - 2 * kidx)) & 3;
                    if (mode == 0) {  // Predefined
                        prev_t[kind] = PTabRef{po.predef_off[kind], (uint32_t)kPDefLog[kind], true};
                        c.carry[1 + kind] = PCarry{0, (uint8_t)kPDefLog[kind], 2, {0, 0}};
                    } else if (mode == 1) {  // RLE: one byte = the code (sequences.go:282-289,315-323,343-351)
                        if (q >= lim) P_FAIL(MZD_ERR_TRUNCATED);
                        const uint8_t code = base[q];
                        if (code > kPMaxSym[kind]) P_FAIL(MZD_ERR_FSE_TABLE);
                        if (PASS == 1) {
                            FseBuildDesc fd{};
                            fd.src_off = fb.fse_src0 + c.n_fse_src;
                            fd.dst_off = fb.fse_dev0 + c.n_fse_dev;
                            fd.acc_log = 0;
                            fd.n_sym = 0;
                            fd.ok = 1;
                            po.fse_tabs[fb.fse_tab0 + c.n_fse_tab] = fd;
                            po.fse_src[fd.src_off] = (uint32_t)code << 24;
                        }
                        prev_t[kind] = PTabRef{fb.fse_dev0 + c.n_fse_dev, 0, true};
                        c.carry[1 + kind] = PCarry{c.n_fse_dev, 0, 1, {0, 0}};
                        c.n_fse_tab++;
                        c.n_fse_src += 1;
                        c.n_fse_dev += 1;
                        q += 1;
                    } else if (mode == 3) {  // Repeat
                        if (!prev_t[kind].valid) {
                            if (first_unit || PASS == 1) P_FAIL(MZD_ERR_NO_PREV_TABLE);
                            c.need |= 2u << kind;
                            prev_t[kind].valid = true;
                        }
                    } else {  // Compressed
                        int nsym = 0, al = 0;
                        const int used = p_read_fse_description(base + q, lim - q, sc.prob, nsym, al);
                        if (used < 0) P_FAIL(-used);
                        if (al > kPMaxLog[kind]) P_FAIL(MZD_ERR_UNSUPPORTED);
                        // (as the host planner: a description that runs on in zeros beyond the kind's codes is the table without them; one
                        // that gives such a symbol cells is outside what the code tables translate -- a documented limit, fse.go:219-224)
                        while (nsym > kPMaxSym[kind] + 1 && sc.prob[nsym - 1] == 0) nsym--;
                        if (nsym > kPMaxSym[kind] + 1) P_FAIL(MZD_ERR_UNSUPPORTED);
                        const uint32_t ncell = ((uint32_t)nsym + 1) / 2;
                        if (PASS == 1) {
                            FseBuildDesc fd{};
                            fd.src_off = fb.fse_src0 + c.n_fse_src;
                            fd.dst_off = fb.fse_dev0 + c.n_fse_dev;
                            fd.acc_log = (uint8_t)al;
                            fd.n_sym = (uint8_t)nsym;
                            fd.ok = 1;
                            po.fse_tabs[fb.fse_tab0 + c.n_fse_tab] = fd;
                            for (uint32_t i = 0; i < ncell; i++)
                                po.fse_src[fd.src_off + i] = (uint32_t)(uint16_t)sc.prob[2 * i] |
                                                             ((2 * i + 1 < (uint32_t)nsym ? (uint32_t)(uint16_t)sc.prob[2 * i + 1] : 0u) << 16);
                        }
                        prev_t[kind] = PTabRef{fb.fse_dev0 + c.n_fse_dev, (uint32_t)al, true};
                        c.carry[1 + kind] = PCarry{c.n_fse_dev, (uint8_t)al, 1, {0, 0}};
                        c.n_fse_tab++;
                        c.n_fse_src += ncell;
                        c.n_fse_dev += 1u << al;
                        q += (uint64_t)used;
                    }
                    use[kind] = prev_t[kind];
                }
                if (q > lim) P_FAIL(MZD_ERR_TRUNCATED);
                if (q == lim) P_FAIL(MZD_ERR_BAD_PADDING);  // empty bitstream: the reference would spin at sequences.go:133
                if (nseq == 0) {
                    // zero sequences in the two-byte form (planner.cpp parse_compressed_block): padding + initial states must use the
                    // bitstream up (sequences.go:126-208); the execution stage reports the verdict in the sequence stage's place
                    const uint8_t top = base[lim - 1];
                    const int used = (8 - (31 - __clz((int)max((uint32_t)top, 1u)))) + (int)use[MZD_FSE_LL].log + (int)use[MZD_FSE_OF].log + (int)use[MZD_FSE_ML].log;
                    d.pad[1] = top == 0 ? (uint8_t)MZD_ERR_BAD_PADDING : ((long long)(lim - q) * 8 == (long long)used ? (uint8_t)MZD_OK : (uint8_t)MZD_ERR_SEQ_BITS);
                }
                d.n_seq = nseq;
                d.rec_off = fb.rec0 + c.n_rec;
                d.tile_off = fb.tile0 + c.n_tile;
                if (PASS == 1) {
                    SeqTask t{};
                    t.in_off = q;
                    t.in_size = (uint32_t)(lim - q);
                    t.n_seq = nseq;
                    t.rec_off = d.rec_off;
                    t.tile_off = d.tile_off;
                    t.block = bi;
                    t.ll_off = use[MZD_FSE_LL].off;
                    t.of_off = use[MZD_FSE_OF].off;
                    t.ml_off = use[MZD_FSE_ML].off;
                    t.ll_log = (uint8_t)use[MZD_FSE_LL].log;
                    t.of_log = (uint8_t)use[MZD_FSE_OF].log;
                    t.ml_log = (uint8_t)use[MZD_FSE_ML].log;
                    t.hist_known = seen_seq ? 0 : 1;
                    po.seq_tasks[fb.seq0 + c.n_seq] = t;
                }
                seen_seq = true;
                c.max_seq_logs = max(c.max_seq_logs & 0xFFu, (uint32_t)use[MZD_FSE_LL].log) |
                                 (max((c.max_seq_logs >> 8) & 0xFFu, (uint32_t)use[MZD_FSE_ML].log) << 8) |
                                 (max((c.max_seq_logs >> 16) & 0xFFu, (uint32_t)use[MZD_FSE_OF].log) << 16);
                c.n_seq++;
                c.n_rec += nseq;
                c.n_tile += (nseq + 63) / 64;
                c.comp_bytes += lim - q;
            }
            p = lim;
        }
        if (PASS == 1) po.blocks[bi] = d;
        c.n_blocks++;
    }
    c.saw_last = last ? 1u : 0u;
    // the content checksum is not part of what the reference consumes (framereader.go:84-94)
    // (a later unit of a large frame does not know the frame header's flag: it reports the four bytes behind the last block, the
    // host keeps them if the first unit saw the flag)
    if (last && (((fhd >> 2) & 1) || !first_unit) && end - p >= 4) {
        c.checksum = base[p] | ((uint32_t)base[p + 1] << 8) | ((uint32_t)base[p + 2] << 16) | ((uint32_t)base[p + 3] << 24);
        // (a later unit: "the four bytes were there" -- without it the host would turn a frame cut inside its checksum into
        // has_checksum = 1, checksum = 0, where the single-lane walk and the host planner leave the flag clear)
        c.flags |= first_unit ? MZD_FRAME_HAS_CHECKSUM : 0x40000000u;
    }
    // never more than the blocks can regenerate: a (corrupt) header may declare any content size
    if (c.content_size != MZD_UNKNOWN_SIZE && (uflags & kUnitFinal)) c.out_bound = min(c.out_bound, c.content_size);
    if (PASS == 0) {
        c.status = MZD_OK;
        cnt = c;
    } else if ((uflags & kUnitFirst) && (uflags & kUnitFinal)) {  // (the frame of several units: the host writes its DFrame)
        DFrame df{};
        df.out_offset = fb.out_off;
        df.out_capacity = fb.out_cap;
        df.content_size = c.content_size;
        df.first_block = fb.block0;
        df.n_blocks = c.n_blocks;
        df.plan_status = MZD_OK;
        df.checksum = c.checksum;
        df.has_checksum = (c.flags & MZD_FRAME_HAS_CHECKSUM) ? 1 : 0;
        po.frames[0] = df;  // `po.frames` is pre-offset to this frame by the caller
    }
#undef P_FAIL
}

template <int PASS>
__global__ __launch_bounds__(64) void k_parse(const uint8_t *__restrict__ in, uint64_t in_size, const ParseUnit *__restrict__ units,
                                              uint32_t n_units, ParseScratch *scratch, FrameCount *counts,
                                              const FrameBase *__restrict__ bases, ParseOut po)
{
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x, nthr = gridDim.x * 64;
    ParseScratch &sc = scratch[tid];
    for (uint32_t u = tid; u < n_units; u += nthr) {
        const ParseUnit un = units[u];
        uint64_t off = un.begin, end = un.end;
        if (off > in_size || end > in_size || end < off) { off = 0; end = 0; }  // out of the blob: reported as truncated
        if (PASS == 0) {
            FrameCount c{};
            p_walk_frame<0>(in, off, end, sc, c, FrameBase{}, po, un.flags, un.max_blocks);
            counts[u] = c;
        } else {
            const FrameBase fb = bases[u];
            ParseOut pf = po;
            pf.frames = po.frames + un.frame;
            if (fb.pad != MZD_OK) {  // the status of the unit's FRAME as the host found it (a later unit's error fails the first one's too)
                if (un.flags & kUnitFirst) {
                    DFrame df{};
                    df.out_offset = fb.out_off;
                    df.content_size = MZD_UNKNOWN_SIZE;
                    df.first_block = fb.block0;
                    df.plan_status = (int32_t)fb.pad;
                    pf.frames[0] = df;
                }
            } else {
                FrameCount dummy;
                p_walk_frame<1>(in, off, end, sc, dummy, fb, pf, un.flags, un.max_blocks);
            }
        }
    }
}

// Large frames: where does every block start?  One lane per listed frame walks the 3-byte block headers (block.go:33-55) -- a chain
// of dependent loads, a block apart; the only serial part of planning such a frame -- and writes the offset of every block AFTER
// the first into starts[cap_off[j] ...] (room for cap_off[j + 1] - cap_off[j] of them; a frame with more is left to one lane:
// n_found = ~0).  It stops at the last block, or where a header cannot be followed (truncated input, a reserved block type, a
// size above the limit): the unit that starts at the last offset found meets that header again and reports it.
__global__ __launch_bounds__(64) void k_parse_index(const uint8_t *__restrict__ in, uint64_t in_size, const uint64_t *__restrict__ frame_off,
                                                    const uint64_t *__restrict__ frame_len, const uint32_t *__restrict__ cap_off,
                                                    uint32_t n_large, uint64_t *__restrict__ starts, uint32_t *__restrict__ n_found)
{
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_large) return;
    uint64_t p = frame_off[j], end = frame_off[j] + frame_len[j];
    uint32_t n = 0;
    const uint32_t cap = cap_off[j + 1] - cap_off[j];
    uint64_t *out = starts + cap_off[j];
    if (p > in_size || end > in_size || end - p < 5) {
        n_found[j] = 0;
        return;
    }
    {
        const uint8_t fhd = in[p + 4];
        const bool single = (fhd >> 5) & 1;
        const int dict_bytes = (fhd & 3) == 3 ? 4 : (fhd & 3);
        const int fcs_flag = fhd >> 6;
        p += 5 + (single ? 0 : 1) + dict_bytes + (fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag));
    }
    bool first = true;
    while (p + 3 <= end) {
        if (!first) {
            if (n >= cap) {
                n = 0xFFFFFFFFu;
                break;
            }
            out[n++] = p;
        }
        first = false;
        const uint32_t h = in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
        const int type = (h >> 1) & 3;
        const uint32_t size = h >> 3;
        if ((h & 1) || type == 3 || size > kBlockMax) break;
        const uint64_t adv = 3 + (type == MZD_BLOCK_RLE ? 1ull : (uint64_t)size);
        if (end - p < adv) break;
        p += adv;
    }
    n_found[j] = n;
}

}  // namespace mzd
