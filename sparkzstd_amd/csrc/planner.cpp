// planner.cpp -- host side of the mzd boundary: turns zstd frames into the flat
// block / table descriptors of include/mzd.h.
//
// This is the C++ stand-in for sparkzstd's Go host code (no Go toolchain in this
// image).  It keeps exactly what BASELINE.json's north_star leaves on the host:
// frame/block header parsing, literals/sequences section header parsing and
// FSE/Huffman table construction.  Nothing here decodes a Huffman literal stream or
// a sequence bitstream, and nothing here copies output bytes -- that is device work.
//
// Reference behaviour followed (file:line in /root/reference):
//   frame header      structure/frame.go:23-127, decompression/framedecompressor.go:130-150,306-374
//   block header      structure/block.go:33-55
//   literals header   structure/literals.go:67-204,209-289, jump table :46-62
//   huffman weights   structure/huffman.go:40-107 (+ fse/fse.go:307-390)
//   huffman table     structure/huffman.go:112-190
//   sequences header  structure/sequences.go:228-269,371-433
//   table selection   structure/sequences.go:275-366, carry-over framedecompressor.go:283-294
//   FSE description   fse/fse.go:28-130, table build fse/fse.go:136-230, predefined.go
#include "../../include/mzd.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kBlockMax = 128 * 1024;  // block.go:50
constexpr uint32_t kPredefLL = 0xFFFFFFF0u, kPredefOF = 0xFFFFFFF1u, kPredefML = 0xFFFFFFF2u;
// "the table the blocks BEFORE this range left" (a frame's blocks parsed in ranges on several threads: parse_blocks_parallel)
constexpr uint32_t kInheritLL = 0xFFFFFFE0u, kInheritOF = 0xFFFFFFE1u, kInheritML = 0xFFFFFFE2u, kInheritHuf = 0xFFFFFFE3u;

inline int highbit(uint32_t v) { return v ? 31 - __builtin_clz(v) : 0; }  // fse.go:235-249

// forward (LSB-first) bit reader over a byte range: bitstream/bitstream.go:39-90
struct FwdBits {
    const uint8_t *p;
    uint64_t nbits;
    uint64_t pos = 0;
    bool overrun = false;
    FwdBits(const uint8_t *d, uint64_t len) : p(d), nbits(len * 8) {}
    uint32_t read(int n)
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

// backward bit reader used only for the (tiny) Huffman-weight stream: reversebitstream.go:17-88
struct RevBits {
    const uint8_t *p;
    int64_t cursor;  // index of next bit; -1 == empty; < -1 == over-read
    RevBits(const uint8_t *d, int64_t len) : p(d), cursor(len * 8 - 1) {}
    uint32_t read(int n)
    {
        uint32_t v = 0;
        for (int i = 0; i < n; i++) {
            int64_t b = cursor - i;
            v = (v << 1) | (b >= 0 ? (uint32_t)((p[b >> 3] >> (b & 7)) & 1u) : 0u);
        }
        cursor -= n;
        return v;
    }
};

struct NormCounts {
    int acc_log = 0;
    std::vector<int16_t> prob;  // -1 == "less than one"
};

// fse.go:28-130.  Returns bytes used, or <0: -MZD_ERR_*.
int read_fse_description(const uint8_t *src, uint64_t len, NormCounts &nc)
{
    FwdBits bs(src, len);
    nc.acc_log = 5 + (int)bs.read(4);
    nc.prob.clear();
    if (bs.overrun) return -MZD_ERR_TRUNCATED;
    if (nc.acc_log > 9) return -MZD_ERR_UNSUPPORTED;  // spec max: LL 9, ML 9, OF 8, weights 6
    int32_t remaining = 1 << nc.acc_log;
    while (remaining > 0) {
        int nb = highbit((uint32_t)remaining + 1) + 1;
        uint32_t v = bs.read(nb);
        if (bs.overrun) return -MZD_ERR_TRUNCATED;
        uint32_t lower = (1u << (nb - 1)) - 1;
        uint32_t thresh = (1u << nb) - 1 - (uint32_t)(remaining + 1);
        if ((v & lower) < thresh) {
            v &= lower;
            bs.pos--;  // "small" value: it used one bit less (fse.go:65-77)
        } else if (v > lower) {
            v -= thresh;
        }
        int prob = (int)v - 1;
        if (nc.prob.size() >= 256) return -MZD_ERR_FSE_TABLE;
        nc.prob.push_back((int16_t)prob);
        remaining -= prob < 0 ? 1 : prob;
        if (prob == 0) {  // zero-probability run lengths, 2 bits at a time (fse.go:96-117)
            uint32_t rep = 3;
            while (rep == 3) {
                rep = bs.read(2);
                if (bs.overrun) return -MZD_ERR_TRUNCATED;
                for (uint32_t i = 0; i < rep; i++) {
                    if (nc.prob.size() >= 256) return -MZD_ERR_FSE_TABLE;
                    nc.prob.push_back(0);
                }
            }
        }
    }
    if (remaining != 0) return -MZD_ERR_FSE_TABLE;  // fse.go:126-128
    return (int)((bs.pos + 7) / 8);
}

// fse.go:136-230: spread symbols, then derive nbBits / baseline per cell.
int build_fse_cells(const NormCounts &nc, std::vector<mzd_fse_entry> &cells)
{
    const int size = 1 << nc.acc_log;
    const int nsym = (int)nc.prob.size();
    cells.assign((size_t)size, mzd_fse_entry{0, 0, 0});
    std::vector<uint8_t> taken((size_t)size, 0);
    std::vector<uint16_t> next((size_t)nsym, 0);
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (nc.prob[s] == -1) {
            if (high < 0) return MZD_ERR_FSE_TABLE;
            cells[(size_t)high].symbol = (uint8_t)s;
            taken[(size_t)high] = 1;
            high--;
            next[(size_t)s] = 1;
        } else {
            next[(size_t)s] = (uint16_t)nc.prob[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3;
    const int mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < nc.prob[s]; i++) {
            if (taken[(size_t)pos]) return MZD_ERR_FSE_TABLE;  // fse.go:166-169
            cells[(size_t)pos].symbol = (uint8_t)s;
            taken[(size_t)pos] = 1;
            int guard = 0;
            do {
                pos = (pos + step) & mask;
                if (++guard > size + 1) return MZD_ERR_FSE_TABLE;
            } while (pos > high);
        }
    }
    if (pos != 0) return MZD_ERR_FSE_TABLE;  // fse.go:186-189
    for (int i = 0; i < size; i++) {
        uint32_t n = next[cells[(size_t)i].symbol]++;
        int nb = nc.acc_log - highbit(n);
        cells[(size_t)i].nbits = (uint8_t)nb;
        cells[(size_t)i].baseline = (uint16_t)((n << nb) - (uint32_t)size);
    }
    return MZD_OK;
}

// predefined.go:3-16,34-45,64-68
const int16_t kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
const int16_t kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
constexpr int kMaxSym[3] = {35, 31, 52};  // LL, OF, ML highest legal code
constexpr int kMaxLog[3] = {9, 8, 9};

// huffman.go:40-107: weights, either 4-bit direct or FSE-compressed (two
// interleaved states on one table, fse.go:307-390).  Returns bytes used or <0.
int read_huffman_weights(const uint8_t *src, uint64_t len, std::vector<uint8_t> &w)
{
    w.clear();
    if (len < 1) return -MZD_ERR_TRUNCATED;
    const int header = src[0];
    if (header >= 128) {
        const int n = header - 127;
        const int nbytes = (n + 1) / 2;
        if (1 + (uint64_t)nbytes > len) return -MZD_ERR_TRUNCATED;
        for (int i = 0; i < n; i++) {
            uint8_t b = src[1 + i / 2];
            w.push_back((i & 1) ? (uint8_t)(b & 15) : (uint8_t)(b >> 4));
        }
        return 1 + nbytes;
    }
    if (1 + (uint64_t)header > len) return -MZD_ERR_TRUNCATED;
    NormCounts nc;
    int used = read_fse_description(src + 1, (uint64_t)header, nc);
    if (used < 0) return used;
    std::vector<mzd_fse_entry> cells;
    int rc = build_fse_cells(nc, cells);
    if (rc) return -rc;
    const int64_t slen = header - used;
    if (slen <= 0) return -MZD_ERR_TRUNCATED;
    RevBits rb(src + 1 + used, slen);
    int pad = 0;
    while (rb.read(1) == 0)
        if (++pad >= 8) return -MZD_ERR_BAD_PADDING;  // fse.go:314-325
    uint32_t st[2];
    st[0] = rb.read(nc.acc_log);
    st[1] = rb.read(nc.acc_log);
    for (int turn = 0;; turn ^= 1) {
        const mzd_fse_entry &e = cells[st[turn]];
        if (w.size() >= 255) return -MZD_ERR_HUF_WEIGHTS;
        w.push_back(e.symbol);
        st[turn] = e.baseline + rb.read(e.nbits);
        if (rb.cursor < -1) {  // over-read: flush the other state's symbol and stop (fse.go:363-383)
            if (w.size() >= 255) return -MZD_ERR_HUF_WEIGHTS;
            w.push_back(cells[st[turn ^ 1]].symbol);
            break;
        }
    }
    return 1 + header;
}

// huffman.go:112-131: MaxBits from the weights; the checks that make the fill of :163-187 succeed
int huffman_max_bits(const std::vector<uint8_t> &w, int &max_bits)
{
    uint32_t sum = 0;
    for (uint8_t x : w) {
        if (x > 11) return MZD_ERR_HUF_WEIGHTS;
        if (x) sum += 1u << (x - 1);
    }
    if (sum == 0) return MZD_ERR_HUF_WEIGHTS;
    max_bits = highbit(sum) + 1;
    const uint32_t left = (1u << max_bits) - sum;
    if (left & (left - 1)) return MZD_ERR_HUF_WEIGHTS;  // huffman.go:128-130
    if (max_bits > 11) return MZD_ERR_UNSUPPORTED;      // format limit; device table slot is 2048 cells
    return MZD_OK;
}

// huffman.go:112-190
int build_huffman_cells(const std::vector<uint8_t> &w, std::vector<mzd_huf_entry> &cells, int &max_bits)
{
    int rc0 = huffman_max_bits(w, max_bits);
    if (rc0) return rc0;
    uint32_t sum = 0;
    for (uint8_t x : w)
        if (x) sum += 1u << (x - 1);
    const uint32_t left = (1u << max_bits) - sum;
    std::vector<uint8_t> len(w.size() + 1);
    for (size_t i = 0; i < w.size(); i++) len[i] = w[i] ? (uint8_t)(max_bits + 1 - w[i]) : 0;
    len[w.size()] = (uint8_t)(max_bits + 1 - (highbit(left) + 1));
    for (uint8_t l : len)
        if (l > max_bits) return MZD_ERR_HUF_WEIGHTS;
    cells.assign((size_t)1 << max_bits, mzd_huf_entry{0, 0});
    // longest codes first from cell 0, ascending symbol inside a length (huffman.go:163-187)
    size_t at = 0;
    for (int l = max_bits; l >= 1; l--) {
        const size_t span = (size_t)1 << (max_bits - l);
        for (size_t s = 0; s < len.size(); s++) {
            if (len[s] != l) continue;
            if (at + span > cells.size()) return MZD_ERR_HUF_WEIGHTS;
            for (size_t j = 0; j < span; j++) cells[at + j] = mzd_huf_entry{(uint8_t)s, (uint8_t)l};
            at += span;
        }
    }
    if (at != cells.size()) return MZD_ERR_HUF_WEIGHTS;  // huffman.go:173-175
    return MZD_OK;
}

// Everything one frame contributes; indices are frame-local until merged.
struct FramePart {
    int status = MZD_OK;
    uint64_t src_begin = 0;  // offset of the frame in the input blob
    uint64_t consumed = 0;
    uint64_t content_size = MZD_UNKNOWN_SIZE;
    uint64_t window_size = 0;
    uint32_t checksum = 0, flags = 0;  // content checksum after the last block, MZD_FRAME_*
    uint64_t out_bound = 0;  // upper bound of the regenerated size
    std::vector<mzd_block_desc> blocks;
    std::vector<mzd_fse_table_desc> fse_tables;
    std::vector<mzd_fse_entry> fse_entries;
    std::vector<mzd_huf_table_desc> huf_tables;
    std::vector<mzd_huf_entry> huf_entries;
};

struct FrameParser {
    const uint8_t *base;  // input blob
    uint64_t begin, end;  // frame extent inside the blob
    FramePart &out;
    // "previous" tables = last table actually used, per kind (framedecompressor.go:283-294)
    uint32_t prev_huf = MZD_NO_TABLE, prev_ll = MZD_NO_TABLE, prev_of = MZD_NO_TABLE, prev_ml = MZD_NO_TABLE;

    bool device_tables = false;  // emit FSE tables as normalised counts (MZD_FSE_FROM_COUNTS)
    unsigned threads = 1;        // run(): a large frame's blocks are parsed in ranges on this many host threads

    FrameParser(const uint8_t *b, uint64_t off, uint64_t len, FramePart &o, bool dev = false)
        : base(b), begin(off), end(off + len), out(o), device_tables(dev) {}

    uint32_t add_fse_table(const std::vector<mzd_fse_entry> &cells, int acc_log, int kind)
    {
        mzd_fse_table_desc d{};
        d.entries_off = (uint32_t)out.fse_entries.size();
        d.acc_log = (uint8_t)acc_log;
        d.kind = (uint8_t)kind;
        out.fse_entries.insert(out.fse_entries.end(), cells.begin(), cells.end());
        out.fse_tables.push_back(d);
        return (uint32_t)out.fse_tables.size() - 1;
    }

    // sequences.go:275-366 for one table kind.  Returns bytes used or <0.
    int select_table(int mode, int kind, uint64_t p, uint64_t lim, uint32_t &prev, uint32_t &chosen)
    {
        switch (mode) {
        case 0:  // Predefined
            chosen = prev = (kind == MZD_FSE_LL ? kPredefLL : kind == MZD_FSE_OF ? kPredefOF : kPredefML);
            return 0;
        case 1: {  // RLE: one byte = the code (sequences.go:282-289,315-323,343-351)
            if (p >= lim) return -MZD_ERR_TRUNCATED;
            uint8_t code = base[p];
            if (code > kMaxSym[kind]) return -MZD_ERR_FSE_TABLE;
            std::vector<mzd_fse_entry> one{mzd_fse_entry{0, 0, code}};
            chosen = prev = add_fse_table(one, 0, kind);
            return 1;
        }
        case 3:  // Repeat
            if (prev == MZD_NO_TABLE) return -MZD_ERR_NO_PREV_TABLE;
            chosen = prev;
            return 0;
        default: {  // Compressed
            NormCounts nc;
            int used = read_fse_description(base + p, lim - p, nc);
            if (used < 0) return used;
            if (nc.acc_log > kMaxLog[kind]) return -MZD_ERR_UNSUPPORTED;
            // More symbols than the kind has codes (LL 36 / OF 32 / ML 53).  The reference builds such a table all the same: a symbol
            // beyond its translation array stays untranslated, with no extra bits (fse.go:219-224, `if len(symbolTranslation) > symbol`).
            // Symbols of probability 0 never get a cell: a description that merely RUNS ON in zeros is the table without them (what the
            // reference decodes it as).  One that gives such a symbol cells is not zstd (libzstd rejects it) and is outside what the
            // device's code tables translate: MZD_ERR_UNSUPPORTED, a documented limit -- not MZD_ERR_FSE_TABLE, which would claim that
            // the reference fails it too (found by the soak of round 6, seed 6: an ML table whose symbol 53 is "less than one").
            while ((int)nc.prob.size() > kMaxSym[kind] + 1 && nc.prob.back() == 0) nc.prob.pop_back();
            if ((int)nc.prob.size() > kMaxSym[kind] + 1) return -MZD_ERR_UNSUPPORTED;
            std::vector<mzd_fse_entry> cells;
            if (device_tables) {
                // counts only: two int16 per cell; read_fse_description has checked that they sum to the
                // table size, which is all the spread of fse.go:136-190 needs to succeed
                cells.assign((nc.prob.size() + 1) / 2, mzd_fse_entry{0, 0, 0});
                for (size_t s = 0; s < nc.prob.size(); s++) {
                    const uint16_t c = (uint16_t)nc.prob[s];
                    mzd_fse_entry &e = cells[s >> 1];
                    if (s & 1) { e.nbits = (uint8_t)(c & 0xFF); e.symbol = (uint8_t)(c >> 8); }
                    else e.baseline = c;
                }
                chosen = prev = add_fse_table(cells, nc.acc_log, kind);
                out.fse_tables.back().build = (uint16_t)(MZD_FSE_FROM_COUNTS | nc.prob.size());
                return used;
            }
            int rc = build_fse_cells(nc, cells);
            if (rc) return -rc;
            chosen = prev = add_fse_table(cells, nc.acc_log, kind);
            return used;
        }
        }
    }

    int parse_compressed_block(uint64_t p, uint32_t bsize, mzd_block_desc &bd)
    {
        const uint64_t lim = p + bsize;
        if (bsize < 1) return MZD_ERR_TRUNCATED;
        // ---- literals section header (literals.go:67-204)
        const uint8_t b0 = base[p];
        const int ltype = b0 & 3, sf = (b0 >> 2) & 3;
        uint32_t regen = 0, csize = 0;
        int hdr, streams = 1;
        if (ltype <= 1) {
            hdr = (sf == 1) ? 2 : (sf == 3 ? 3 : 1);
            if (p + hdr > lim) return MZD_ERR_TRUNCATED;
            if (hdr == 1) regen = b0 >> 3;
            else if (hdr == 2) regen = (b0 >> 4) + ((uint32_t)base[p + 1] << 4);
            else regen = (b0 >> 4) + ((uint32_t)base[p + 1] << 4) + ((uint32_t)base[p + 2] << 12);
            csize = ltype == 0 ? regen : 1;
        } else {
            hdr = sf <= 1 ? 3 : sf + 2;
            if (p + hdr > lim) return MZD_ERR_TRUNCATED;
            uint64_t v = 0;
            for (int i = 0; i < hdr; i++) v |= (uint64_t)base[p + i] << (8 * i);
            v >>= 4;
            const int bits = sf <= 1 ? 10 : (sf == 2 ? 14 : 18);
            regen = (uint32_t)(v & ((1u << bits) - 1));
            csize = (uint32_t)((v >> bits) & ((1u << bits) - 1));
            streams = sf == 0 ? 1 : 4;
        }
        if (regen > kBlockMax) return MZD_ERR_CORRUPT_SIZES;
        uint64_t q = p + hdr;
        bd.lit_regen = regen;
        bd.lit_streams = (uint8_t)streams;
        bd.huf_table = MZD_NO_TABLE;
        if (ltype == 0) {
            bd.lit_type = MZD_LIT_RAW;
            if (q + regen > lim) return MZD_ERR_TRUNCATED;
            bd.lit_off = q;
            q += regen;
        } else if (ltype == 1) {
            bd.lit_type = MZD_LIT_RLE;
            if (q + 1 > lim) return MZD_ERR_TRUNCATED;
            bd.lit_off = q;
            q += 1;
        } else {
            bd.lit_type = MZD_LIT_HUF;
            if (q + csize > lim) return MZD_ERR_TRUNCATED;
            const uint64_t lit_end = q + csize;
            if (ltype == 3) {  // Treeless: literals.go:247-252
                if (prev_huf == MZD_NO_TABLE) return MZD_ERR_NO_PREV_TABLE;
            } else {           // literals.go:254-267
                std::vector<uint8_t> w;
                int used = read_huffman_weights(base + q, lit_end - q, w);
                if (used < 0) return -used;
                std::vector<mzd_huf_entry> cells;
                int mb = 0;
                uint32_t form = 0;
                if (device_tables) {
                    // weights only (two per cell slot); huffman_max_bits has made the checks under which the
                    // table fill of huffman.go:163-187 cannot fail
                    int rc = huffman_max_bits(w, mb);
                    if (rc) return rc;
                    cells.assign((w.size() + 1) / 2, mzd_huf_entry{0, 0});
                    for (size_t i = 0; i < w.size(); i++) {
                        if (i & 1) cells[i >> 1].nbits = w[i];
                        else cells[i >> 1].symbol = w[i];
                    }
                    form = MZD_HUF_FROM_WEIGHTS | ((uint32_t)w.size() << 8);
                } else {
                    int rc = build_huffman_cells(w, cells, mb);
                    if (rc) return rc;
                }
                if (out.huf_entries.size() & 1) out.huf_entries.push_back(mzd_huf_entry{0, 0});
                mzd_huf_table_desc d{(uint32_t)out.huf_entries.size(), (uint32_t)mb | form};
                out.huf_entries.insert(out.huf_entries.end(), cells.begin(), cells.end());
                out.huf_tables.push_back(d);
                prev_huf = (uint32_t)out.huf_tables.size() - 1;
                q += (uint64_t)used;
            }
            bd.huf_table = prev_huf;
            if (streams == 4) {  // jump table: literals.go:46-62,270-279
                if (q + 6 > lit_end) return MZD_ERR_TRUNCATED;
                uint32_t s1 = base[q] | (base[q + 1] << 8), s2 = base[q + 2] | (base[q + 3] << 8),
                         s3 = base[q + 4] | (base[q + 5] << 8);
                q += 6;
                const uint64_t rest = lit_end - q;
                if ((uint64_t)s1 + s2 + s3 > rest) return MZD_ERR_CORRUPT_SIZES;
                bd.lit_stream_size[0] = s1;
                bd.lit_stream_size[1] = s2;
                bd.lit_stream_size[2] = s3;
                bd.lit_stream_size[3] = (uint32_t)(rest - s1 - s2 - s3);
                if (3 * ((regen + 3) / 4) > regen) return MZD_ERR_HUF_LENGTH;  // literals.go:306-307 would go negative
            } else {
                bd.lit_stream_size[0] = (uint32_t)(lit_end - q);
            }
            bd.lit_off = q;
            q = lit_end;
        }
        // ---- sequences section header (sequences.go:371-433)
        if (q >= lim) return MZD_ERR_TRUNCATED;
        const uint8_t s0 = base[q];
        bd.ll_table = bd.of_table = bd.ml_table = MZD_NO_TABLE;
        if (s0 == 0) {  // sequences.go:395-400
            bd.n_seq = 0;
            bd.seq_off = q + 1;
            bd.seq_size = 0;
            q += 1;
            if (q != lim) return MZD_ERR_CORRUPT_SIZES;  // framedecompressor.go:114-123
            return MZD_OK;
        }
        uint32_t nseq;
        if (s0 < 128) {
            nseq = s0;
            q += 1;
        } else if (s0 < 255) {
            if (q + 2 > lim) return MZD_ERR_TRUNCATED;
            nseq = ((uint32_t)(s0 - 128) << 8) + base[q + 1];
            q += 2;
        } else {
            if (q + 3 > lim) return MZD_ERR_TRUNCATED;
            nseq = base[q + 1] + ((uint32_t)base[q + 2] << 8) + 0x7F00;
            q += 3;
        }
        if (q >= lim) return MZD_ERR_TRUNCATED;
        const uint8_t modes = base[q++];  // sequences.go:228-232
        int used = select_table((modes >> 6) & 3, MZD_FSE_LL, q, lim, prev_ll, bd.ll_table);
        if (used < 0) return -used;
        q += (uint64_t)used;
        used = select_table((modes >> 4) & 3, MZD_FSE_OF, q, lim, prev_of, bd.of_table);
        if (used < 0) return -used;
        q += (uint64_t)used;
        used = select_table((modes >> 2) & 3, MZD_FSE_ML, q, lim, prev_ml, bd.ml_table);
        if (used < 0) return -used;
        q += (uint64_t)used;
        if (q > lim) return MZD_ERR_TRUNCATED;
        if (q == lim) return MZD_ERR_BAD_PADDING;  // empty bitstream: the reference would spin at sequences.go:133
        if (nseq == 0) {
            // Zero sequences in the two-byte form (0x80 0x00).  The reference decodes the section like any other (sequences.go:126-208):
            // the padding, the three initial states, no sequence -- and then wants the bitstream used up to the bit.  What that comes to
            // is decided here (the device's sequence stage has no chain to run for the block) and reported by the execution stage in
            // the stage's place.  (Found by the chunk soak of round 6: such a block passed as its literals.)
            auto log_of = [&](uint32_t idx) -> int {
                if (idx == kPredefLL || idx == kPredefML) return 6;
                if (idx == kPredefOF) return 5;
                if (idx >= kInheritLL) return -1;  // (the table of an earlier range: settled when the ranges are stitched)
                return out.fse_tables[idx].acc_log;
            };
            const int ll = log_of(bd.ll_table), of = log_of(bd.of_table), ml = log_of(bd.ml_table);
            const uint8_t top = base[lim - 1];
            if (ll < 0 || of < 0 || ml < 0) bd.seq_status = 0xFF;  // (decided by parse_blocks_parallel's stitching)
            else if (top == 0) bd.seq_status = MZD_ERR_BAD_PADDING;  // sequences.go:141-143
            else bd.seq_status = (int64_t)(lim - q) * 8 - (8 - highbit(top)) - (ll + of + ml) == 0 ? MZD_OK : MZD_ERR_SEQ_BITS;
        }
        bd.n_seq = nseq;
        bd.seq_off = q;
        bd.seq_size = (uint32_t)(lim - q);
        return MZD_OK;
    }

    // magic + frame header (framedecompressor.go:130-150,306-374; frame.go) at p; p ends behind it
    int parse_header(uint64_t &p, uint8_t &fhd)
    {
        if (end - p < 5) return MZD_ERR_TRUNCATED;
        if (!(base[p] == 0x28 && base[p + 1] == 0xB5 && base[p + 2] == 0x2F && base[p + 3] == 0xFD)) return MZD_ERR_MAGIC;
        fhd = base[p + 4];
        p += 5;
        const bool single = (fhd >> 5) & 1;
        static const int kDict[4] = {0, 1, 2, 4};
        const int dict_bytes = kDict[fhd & 3];
        const int fcs_flag = fhd >> 6;
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        if (end - p < (uint64_t)(!single) + dict_bytes + fcs_bytes) return MZD_ERR_TRUNCATED;
        if (!single) {
            const uint8_t wd = base[p++];
            const uint64_t wbase = 1ull << (10 + (wd >> 3));
            out.window_size = wbase + (wbase / 8) * (wd & 7);
        }
        p += dict_bytes;  // dictionary id is read and ignored (no dictionary support: Readme.md:59-62)
        if (fcs_bytes) {
            uint64_t v = 0;
            for (int i = 0; i < fcs_bytes; i++) v |= (uint64_t)base[p + i] << (8 * i);
            if (fcs_bytes == 2) v += 256;  // frame.go:58-60
            out.content_size = v;
            p += fcs_bytes;
            if (single) out.window_size = v;
        }
        return MZD_OK;
    }

    // one block (block.go:33-55, framedecompressor.go:198-303) at p: appended to out.blocks, p ends behind it
    int parse_block(uint64_t &p, bool &last, uint64_t &bound)
    {
        if (end - p < 3) return MZD_ERR_TRUNCATED;
        const uint32_t h = base[p] | ((uint32_t)base[p + 1] << 8) | ((uint32_t)base[p + 2] << 16);
        p += 3;
        last = h & 1;
        const int type = (h >> 1) & 3;
        const uint32_t size = h >> 3;
        if (type == 3) return MZD_ERR_BLOCK_TYPE;
        if (size > kBlockMax) return MZD_ERR_BLOCK_SIZE;
        mzd_block_desc bd{};
        bd.type = (uint8_t)type;
        bd.size = size;
        bd.huf_table = bd.ll_table = bd.of_table = bd.ml_table = MZD_NO_TABLE;
        if (type == MZD_BLOCK_RAW) {
            if (end - p < size) return MZD_ERR_TRUNCATED;
            bd.src_off = p;
            p += size;
            bound += size;
        } else if (type == MZD_BLOCK_RLE) {
            if (end - p < 1) return MZD_ERR_TRUNCATED;
            bd.src_off = p;
            p += 1;
            bound += size;
        } else {
            if (end - p < size) return MZD_ERR_TRUNCATED;
            int rc = parse_compressed_block(p, size, bd);
            if (rc) return rc;
            p += size;
            bound += kBlockMax;
        }
        out.blocks.push_back(bd);
        return MZD_OK;
    }

    bool run_parallel(uint64_t &p, uint64_t &bound);  // (below: needs parse_blocks_parallel)

    void run()
    {
        out.src_begin = begin;
        uint64_t p = begin;
        auto fail = [&](int code) {
            out.status = code;
            out.blocks.clear();
            out.consumed = p - begin;
        };
        uint8_t fhd = 0;
        if (int rc = parse_header(p, fhd)) return fail(rc);
        bool last = false;
        uint64_t bound = 0;
        if (threads > 1 && end - p >= (1u << 20) && run_parallel(p, bound)) last = true;
        while (!last)
            if (int rc = parse_block(p, last, bound)) return fail(rc);
        out.consumed = p - begin;
        // the content checksum is not part of what the reference consumes (framereader.go:84-94); it is
        // recorded for the optional device-side verification
        if (((fhd >> 2) & 1) && end - p >= 4) {
            out.checksum = base[p] | ((uint32_t)base[p + 1] << 8) | ((uint32_t)base[p + 2] << 16) | ((uint32_t)base[p + 3] << 24);
            out.flags |= MZD_FRAME_HAS_CHECKSUM;
        }
        // never more than the blocks can regenerate: a (corrupt) header may declare any content size
        out.out_bound = out.content_size != MZD_UNKNOWN_SIZE ? std::min<uint64_t>(out.content_size, bound) : bound;
    }
};

// cells a table occupies in its array: built (1 << log) or in count / weight form ((n + 1) / 2)
inline uint32_t fse_cells_of(const mzd_fse_table_desc &d) { return (d.build & MZD_FSE_FROM_COUNTS) ? ((uint32_t)(d.build & ~MZD_FSE_FROM_COUNTS) + 1) / 2 : 1u << d.acc_log; }
inline uint32_t huf_cells_of(const mzd_huf_table_desc &d)
{
    return (d.max_bits & MZD_HUF_FROM_WEIGHTS) ? (((d.max_bits >> 8) & 0xFFFFu) + 1) / 2 : 1u << (d.max_bits & 0xFFu);
}

// The blocks of ONE frame whose 3-byte headers sit at pos[0 .. n), parsed in contiguous ranges on `threads` host threads and
// appended to `out` exactly as the serial walk would have left them (same tables, same order, same indices).  A range does not know
// the tables the blocks before it left (Repeat_Mode / Treeless: framedecompressor.go:283-294): its parser starts with the kInherit*
// marks in their place, and the marks are resolved when the ranges are stitched, in order.  carry[0..3] = the LL / OF / ML / Huffman
// table in force before pos[0] (an index into `out`, kPredef*, MZD_NO_TABLE); updated to what the last block leaves.  false: a block
// failed or wanted a table nobody left -- `out` is then as it was, and the caller walks the blocks serially to name the error the way
// the serial walk does.  (What the host planner costs a large frame is table construction: 512 blocks of 128 KiB, 6-7 ms on one thread.)
bool parse_blocks_parallel(const uint8_t *base, uint64_t end, const std::vector<uint64_t> &pos, bool dev, unsigned threads, FramePart &out,
                           uint32_t carry[4], uint64_t &bound)
{
    const size_t n = pos.size();
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n / 16));
    if (threads < 2) return false;
    struct Range {
        FramePart part;
        uint32_t prev[4];
        uint64_t bound = 0;
        int rc = MZD_OK;
    };
    std::vector<Range> rg(threads);
    auto work = [&](unsigned t) {
        Range &r = rg[t];
        FrameParser fp(base, 0, end, r.part, dev);
        fp.prev_ll = kInheritLL;
        fp.prev_of = kInheritOF;
        fp.prev_ml = kInheritML;
        fp.prev_huf = kInheritHuf;
        const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
        for (size_t i = lo; i < hi && r.rc == MZD_OK; i++) {
            uint64_t p = pos[i];
            bool last = false;
            r.rc = fp.parse_block(p, last, r.bound);
        }
        r.prev[0] = fp.prev_ll;
        r.prev[1] = fp.prev_of;
        r.prev[2] = fp.prev_ml;
        r.prev[3] = fp.prev_huf;
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    for (auto &r : rg)
        if (r.rc) return false;
    // what a mark means where it is used; a range that uses one nobody can honour fails the frame (serially, for the error's sake)
    {
        uint32_t c[4] = {carry[0], carry[1], carry[2], carry[3]};
        for (auto &r : rg) {
            for (auto &b : r.part.blocks) {
                if (b.type != MZD_BLOCK_COMPRESSED) continue;
                if (b.seq_status == 0xFF) return false;  // (zero sequences in the long form under an inherited table: the serial walk decides)
                if ((b.ll_table == kInheritLL && c[0] == MZD_NO_TABLE) || (b.of_table == kInheritOF && c[1] == MZD_NO_TABLE) ||
                    (b.ml_table == kInheritML && c[2] == MZD_NO_TABLE) || (b.huf_table == kInheritHuf && c[3] == MZD_NO_TABLE))
                    return false;
            }
            for (int k = 0; k < 4; k++)
                if (r.prev[k] < kInheritLL || r.prev[k] >= kPredefLL) c[k] = r.prev[k] == MZD_NO_TABLE ? c[k] : 0;  // (some table: which one is settled below)
        }
    }
    for (auto &r : rg) {
        const uint32_t fse_base = (uint32_t)out.fse_tables.size();
        for (const auto &d0 : r.part.fse_tables) {
            mzd_fse_table_desc d = d0;
            const uint32_t cells = fse_cells_of(d);
            d.entries_off = (uint32_t)out.fse_entries.size();
            out.fse_entries.insert(out.fse_entries.end(), r.part.fse_entries.begin() + d0.entries_off, r.part.fse_entries.begin() + d0.entries_off + cells);
            out.fse_tables.push_back(d);
        }
        const uint32_t huf_base = (uint32_t)out.huf_tables.size();
        for (const auto &d0 : r.part.huf_tables) {
            mzd_huf_table_desc d = d0;
            const uint32_t cells = huf_cells_of(d);
            if (out.huf_entries.size() & 1) out.huf_entries.push_back(mzd_huf_entry{0, 0});  // (as parse_compressed_block pads)
            d.entries_off = (uint32_t)out.huf_entries.size();
            out.huf_entries.insert(out.huf_entries.end(), r.part.huf_entries.begin() + d0.entries_off, r.part.huf_entries.begin() + d0.entries_off + cells);
            out.huf_tables.push_back(d);
        }
        auto fse = [&](uint32_t idx, int k) -> uint32_t {
            if (idx == MZD_NO_TABLE || (idx >= kPredefLL)) return idx;
            if (idx >= kInheritLL) return carry[k];
            return idx + fse_base;
        };
        auto huf = [&](uint32_t idx) -> uint32_t {
            if (idx == MZD_NO_TABLE) return idx;
            if (idx == kInheritHuf) return carry[3];
            return idx + huf_base;
        };
        for (auto b : r.part.blocks) {
            if (b.type == MZD_BLOCK_COMPRESSED) {
                b.ll_table = fse(b.ll_table, 0);
                b.of_table = fse(b.of_table, 1);
                b.ml_table = fse(b.ml_table, 2);
                b.huf_table = huf(b.huf_table);
            }
            out.blocks.push_back(b);
        }
        carry[0] = fse(r.prev[0], 0);
        carry[1] = fse(r.prev[1], 1);
        carry[2] = fse(r.prev[2], 2);
        carry[3] = huf(r.prev[3]);
        bound += r.bound;
    }
    return true;
}

// run()'s way in: the frame's block headers walked first (block.go:33-55: three bytes each), then the blocks in ranges
bool FrameParser::run_parallel(uint64_t &p, uint64_t &bound)
{
    std::vector<uint64_t> pos;
    uint64_t q = p;
    for (bool last = false; !last;) {
        if (end - q < 3) return false;
        const uint32_t h = base[q] | ((uint32_t)base[q + 1] << 8) | ((uint32_t)base[q + 2] << 16);
        const int type = (h >> 1) & 3;
        const uint32_t size = h >> 3;
        if (type == 3 || size > kBlockMax) return false;
        const uint64_t payload = type == MZD_BLOCK_RLE ? 1 : size;
        if (end - q - 3 < payload) return false;
        pos.push_back(q);
        q += 3 + payload;
        last = h & 1;
    }
    uint32_t carry[4] = {MZD_NO_TABLE, MZD_NO_TABLE, MZD_NO_TABLE, MZD_NO_TABLE};
    if (!parse_blocks_parallel(base, end, pos, device_tables, threads, out, carry, bound)) return false;
    prev_ll = carry[0];
    prev_of = carry[1];
    prev_ml = carry[2];
    prev_huf = carry[3];
    p = q;
    return true;
}

}  // namespace

struct mzd_plan {
    std::vector<uint8_t> owned_blob;
    const uint8_t *ext_blob = nullptr;
    uint64_t ext_size = 0;
    std::vector<int> frame_status;
    std::vector<uint64_t> frame_bound;
    std::vector<mzd_frame_desc> frames;
    std::vector<mzd_block_desc> blocks;
    std::vector<mzd_fse_table_desc> fse_tables;
    std::vector<mzd_fse_entry> fse_entries;
    std::vector<mzd_huf_table_desc> huf_tables;
    std::vector<mzd_huf_entry> huf_entries;
    uint32_t predef[3] = {MZD_NO_TABLE, MZD_NO_TABLE, MZD_NO_TABLE};
    bool device_tables = false;
    mzd_batch view{};

    uint32_t predefined(int kind)
    {
        if (predef[kind] != MZD_NO_TABLE) return predef[kind];
        NormCounts nc;
        if (kind == MZD_FSE_LL) { nc.acc_log = 6; nc.prob.assign(kLLDefault, kLLDefault + 36); }
        else if (kind == MZD_FSE_OF) { nc.acc_log = 5; nc.prob.assign(kOFDefault, kOFDefault + 29); }
        else { nc.acc_log = 6; nc.prob.assign(kMLDefault, kMLDefault + 53); }
        std::vector<mzd_fse_entry> cells;
        build_fse_cells(nc, cells);
        mzd_fse_table_desc d{};
        d.entries_off = (uint32_t)fse_entries.size();
        d.acc_log = (uint8_t)nc.acc_log;
        d.kind = (uint8_t)kind;
        fse_entries.insert(fse_entries.end(), cells.begin(), cells.end());
        fse_tables.push_back(d);
        return predef[kind] = (uint32_t)fse_tables.size() - 1;
    }

    void merge(FramePart &fp, uint64_t rebase)
    {
        const uint32_t fse_base = (uint32_t)fse_tables.size();
        const uint32_t fse_ent_base = (uint32_t)fse_entries.size();
        if (huf_entries.size() & 1) huf_entries.push_back(mzd_huf_entry{0, 0});
        const uint32_t huf_base = (uint32_t)huf_tables.size();
        const uint32_t huf_ent_base = (uint32_t)huf_entries.size();
        for (auto d : fp.fse_tables) {
            d.entries_off += fse_ent_base;
            fse_tables.push_back(d);
        }
        fse_entries.insert(fse_entries.end(), fp.fse_entries.begin(), fp.fse_entries.end());
        for (auto d : fp.huf_tables) {
            d.entries_off += huf_ent_base;
            huf_tables.push_back(d);
        }
        huf_entries.insert(huf_entries.end(), fp.huf_entries.begin(), fp.huf_entries.end());
        auto fix = [&](uint32_t idx) -> uint32_t {
            if (idx == MZD_NO_TABLE) return idx;
            if (idx == kPredefLL) return predefined(MZD_FSE_LL);
            if (idx == kPredefOF) return predefined(MZD_FSE_OF);
            if (idx == kPredefML) return predefined(MZD_FSE_ML);
            return idx + fse_base;
        };
        mzd_frame_desc fd{};
        fd.first_block = (uint32_t)blocks.size();
        fd.n_blocks = (uint32_t)fp.blocks.size();
        fd.content_size = fp.content_size;
        fd.window_size = fp.window_size;
        fd.checksum = fp.checksum;
        fd.flags = (fp.status ? 0 : fp.flags) | (((uint32_t)fp.status & 0xFFu) << MZD_FRAME_PLAN_STATUS_SHIFT);
        for (auto b : fp.blocks) {
            b.src_off += rebase;
            b.lit_off += rebase;
            b.seq_off += rebase;
            if (b.type == MZD_BLOCK_COMPRESSED) {
                b.ll_table = fix(b.ll_table);
                b.of_table = fix(b.of_table);
                b.ml_table = fix(b.ml_table);
                if (b.huf_table != MZD_NO_TABLE) b.huf_table += huf_base;
            }
            blocks.push_back(b);
        }
        frames.push_back(fd);
        frame_status.push_back(fp.status);
        frame_bound.push_back(fp.status ? 0 : fp.out_bound);
    }
};

// ---- one frame in CHUNKS of whole blocks (ABI 9).  The reference decodes a frame block by block into a ring of the frame's window
// size and hands the bytes on as they come (framedecompressor.go:198-303, ringbuffer.go:36-49, framereader.go:51-109); what carries
// from a block to the next is the window, the offset history (framedecompressor.go:23) and the tables last used
// (framedecompressor.go:283-294).  The cursor is the host's part of that: it walks the frame's blocks as their bytes arrive and cuts
// them into batches of ONE frame description each -- a chunk -- whose tables include the ones in force at its start.  The window and
// the history are the device's part (mzd_frame_desc.start / .hist, MZD_FRAME_CONTINUES; mzd_fstream_* in mzd_api.hip).
struct mzd_cursor {
    mzd_plan plan;    // the current chunk's batch
    FramePart carry;  // the tables in force behind the last chunk: what Repeat / Treeless of the next one refer to
    uint32_t prev_huf = MZD_NO_TABLE, prev_ll = MZD_NO_TABLE, prev_of = MZD_NO_TABLE, prev_ml = MZD_NO_TABLE;  // indices into carry, kPredef*, or none
    bool header_done = false, last_done = false, has_checksum = false, checksum_seen = false;
    uint64_t window = 0, content = MZD_UNKNOWN_SIZE, blocks_done = 0, bound_done = 0;
    uint32_t checksum = 0;
    int status = MZD_OK;
    unsigned threads = 0;  // host threads a chunk's blocks are parsed on (0: up to eight)
};

namespace {

// the table `idx` of `from` appended to `to`; -> its index there (sentinels pass through)
uint32_t carry_fse(const FramePart &from, uint32_t idx, FramePart &to)
{
    if (idx == MZD_NO_TABLE || idx >= kPredefLL) return idx;
    mzd_fse_table_desc d = from.fse_tables[idx];
    const uint32_t n = fse_cells_of(d);
    const uint32_t off = d.entries_off;
    d.entries_off = (uint32_t)to.fse_entries.size();
    to.fse_entries.insert(to.fse_entries.end(), from.fse_entries.begin() + off, from.fse_entries.begin() + off + n);
    to.fse_tables.push_back(d);
    return (uint32_t)to.fse_tables.size() - 1;
}

uint32_t carry_huf(const FramePart &from, uint32_t idx, FramePart &to)
{
    if (idx == MZD_NO_TABLE) return idx;
    mzd_huf_table_desc d = from.huf_tables[idx];
    const uint32_t n = huf_cells_of(d);
    const uint32_t off = d.entries_off;
    if (to.huf_entries.size() & 1) to.huf_entries.push_back(mzd_huf_entry{0, 0});
    d.entries_off = (uint32_t)to.huf_entries.size();
    to.huf_entries.insert(to.huf_entries.end(), from.huf_entries.begin() + off, from.huf_entries.begin() + off + n);
    to.huf_tables.push_back(d);
    return (uint32_t)to.huf_tables.size() - 1;
}

}  // namespace

extern "C" {

mzd_plan *mzd_plan_create(void) { return new mzd_plan(); }
void mzd_plan_destroy(mzd_plan *p) { delete p; }
void mzd_plan_reset(mzd_plan *p)
{
    if (!p) return;
    const bool dev = p->device_tables;
    *p = mzd_plan();
    p->device_tables = dev;
}
void mzd_plan_set_device_tables(mzd_plan *p, int on)
{
    if (p) p->device_tables = on != 0;
}

int mzd_plan_add_frame(mzd_plan *p, const uint8_t *frame, uint64_t len, uint64_t *consumed)
{
    if (!p || !frame || p->ext_blob) return MZD_ERR_INVALID_ARG;
    const uint64_t at = p->owned_blob.size();
    // parse in place first, then copy only what the frame used (+ nothing of the checksum)
    FramePart fp;
    {
        FrameParser parser(frame, 0, len, fp, p->device_tables);
        parser.threads = len >= (1u << 20) ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;  // (a large frame: its blocks in ranges)
        parser.run();
    }
    const uint64_t keep = fp.status ? 0 : fp.consumed;
    p->owned_blob.insert(p->owned_blob.end(), frame, frame + keep);
    p->merge(fp, at);
    if (consumed) *consumed = fp.consumed;
    return fp.status;
}

int mzd_plan_add_frames(mzd_plan *p, const uint8_t *blob, const uint64_t *frame_off,
                        const uint64_t *frame_len, uint32_t n_frames, uint32_t n_threads)
{
    if (!p || !blob || !frame_off || !frame_len) return MZD_ERR_INVALID_ARG;
    if (!p->owned_blob.empty() || (p->ext_blob && p->ext_blob != blob)) return MZD_ERR_INVALID_ARG;
    p->ext_blob = blob;  // adopted, not copied: the caller keeps it alive
    uint64_t hi = p->ext_size;
    for (uint32_t i = 0; i < n_frames; i++) hi = std::max(hi, frame_off[i] + frame_len[i]);
    p->ext_size = hi;
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t all_threads = n_threads;
    n_threads = std::min<uint32_t>(n_threads, std::max<uint32_t>(1, n_frames / 64));
    // (few frames: what is left of the threads goes INTO the large ones, whose blocks are then parsed in ranges)
    const uint32_t inner = std::min<uint32_t>(8, std::max<uint32_t>(1, all_threads / n_threads));
    std::vector<FramePart> parts(n_frames);
    const bool dev = p->device_tables;
    auto work = [&](uint32_t t) {
        for (uint32_t i = t; i < n_frames; i += n_threads) {
            FrameParser parser(blob, frame_off[i], frame_len[i], parts[i], dev);
            parser.threads = inner;
            parser.run();
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    int first = MZD_OK;
    for (uint32_t i = 0; i < n_frames; i++) {
        if (parts[i].status && !first) first = parts[i].status;
        p->merge(parts[i], 0);
        parts[i] = FramePart();
    }
    return first;
}

const mzd_batch *mzd_plan_finalize(mzd_plan *p)
{
    if (!p) return nullptr;
    uint64_t at = 0;
    for (size_t i = 0; i < p->frames.size(); i++) {
        p->frames[i].out_offset = at;
        p->frames[i].out_capacity = p->frame_bound[i];
        at += (p->frame_bound[i] + 255) & ~255ull;
    }
    at += 256;  // tail slack: the execution kernel's 16-byte source loads may run past the last slab
    mzd_batch &v = p->view;
    v = mzd_batch{};
    v.abi_version = MZD_ABI_VERSION;
    v.flags = 0;
    v.in = p->ext_blob ? p->ext_blob : p->owned_blob.data();
    v.in_size = p->ext_blob ? p->ext_size : p->owned_blob.size();
    v.out = nullptr;
    v.out_size = at;
    v.frames = p->frames.data();
    v.n_frames = (uint32_t)p->frames.size();
    v.blocks = p->blocks.data();
    v.n_blocks = (uint32_t)p->blocks.size();
    v.fse_tables = p->fse_tables.data();
    v.n_fse_tables = (uint32_t)p->fse_tables.size();
    v.fse_entries = p->fse_entries.data();
    v.n_fse_entries = (uint32_t)p->fse_entries.size();
    v.huf_tables = p->huf_tables.data();
    v.n_huf_tables = (uint32_t)p->huf_tables.size();
    v.huf_entries = p->huf_entries.data();
    v.n_huf_entries = (uint32_t)p->huf_entries.size();
    return &v;
}

int mzd_plan_frame_status(const mzd_plan *p, uint32_t i)
{
    if (!p || i >= p->frame_status.size()) return MZD_ERR_INVALID_ARG;
    return p->frame_status[i];
}

// ---- frame boundaries of a stream of concatenated frames (multi-frame streams, skippable
// frames).  Header walk only: magic, frame header (frame.go:23-127), block headers (block.go:33-55); nothing
// of a block's content is looked at.  The reference itself reads ONE frame per reader and leaves what follows
// (framereader.go:84-94): this is the splitter a caller with a multi-frame file puts in front.
int mzd_split_frames(const uint8_t *blob, uint64_t size, uint64_t *frame_off, uint64_t *frame_len, uint64_t *out_bound,
                     uint32_t cap, uint32_t *n_frames, uint64_t *out_total)
{
    if ((!blob && size) || !n_frames) return MZD_ERR_INVALID_ARG;
    uint64_t p = 0, total = 0;
    uint32_t n = 0;
    int rc = MZD_OK;
    while (p < size) {
        if (size - p < 4) { rc = MZD_ERR_TRUNCATED; break; }
        const uint32_t magic = blob[p] | ((uint32_t)blob[p + 1] << 8) | ((uint32_t)blob[p + 2] << 16) | ((uint32_t)blob[p + 3] << 24);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame: 4-byte size, then that many bytes of user data
            if (size - p < 8) { rc = MZD_ERR_TRUNCATED; break; }
            const uint64_t sk = blob[p + 4] | ((uint64_t)blob[p + 5] << 8) | ((uint64_t)blob[p + 6] << 16) | ((uint64_t)blob[p + 7] << 24);
            if (size - p - 8 < sk) { rc = MZD_ERR_TRUNCATED; break; }
            p += 8 + sk;
            continue;
        }
        if (magic != 0xFD2FB528u) { rc = MZD_ERR_MAGIC; break; }
        const uint64_t begin = p;
        if (size - p < 5) { rc = MZD_ERR_TRUNCATED; break; }
        const uint8_t fhd = blob[p + 4];
        p += 5;
        const bool single = (fhd >> 5) & 1;
        static const int kDict[4] = {0, 1, 2, 4};
        const int fcs_flag = fhd >> 6;
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (1 << fcs_flag);
        const uint64_t hdr = (uint64_t)(!single) + kDict[fhd & 3] + fcs_bytes;
        if (size - p < hdr) { rc = MZD_ERR_TRUNCATED; break; }
        uint64_t content = MZD_UNKNOWN_SIZE;
        if (fcs_bytes) {
            const uint64_t q = p + hdr - fcs_bytes;
            uint64_t v = 0;
            for (int i = 0; i < fcs_bytes; i++) v |= (uint64_t)blob[q + i] << (8 * i);
            if (fcs_bytes == 2) v += 256;
            content = v;
        }
        p += hdr;
        uint64_t bound = 0;
        bool last = false;
        while (!last) {
            if (size - p < 3) { rc = MZD_ERR_TRUNCATED; break; }
            const uint32_t h = blob[p] | ((uint32_t)blob[p + 1] << 8) | ((uint32_t)blob[p + 2] << 16);
            p += 3;
            last = h & 1;
            const int type = (h >> 1) & 3;
            const uint32_t bs = h >> 3;
            if (type == 3) { rc = MZD_ERR_BLOCK_TYPE; break; }
            if (bs > kBlockMax) { rc = MZD_ERR_BLOCK_SIZE; break; }
            const uint64_t payload = type == MZD_BLOCK_RLE ? 1 : bs;
            if (size - p < payload) { rc = MZD_ERR_TRUNCATED; break; }
            p += payload;
            bound += type == MZD_BLOCK_COMPRESSED ? kBlockMax : bs;
        }
        if (rc) break;
        if ((fhd >> 2) & 1) {  // content checksum: belongs to the frame
            if (size - p < 4) { rc = MZD_ERR_TRUNCATED; break; }
            p += 4;
        }
        const uint64_t ob = content != MZD_UNKNOWN_SIZE ? std::min(content, bound) : bound;
        if (n < cap) {
            if (frame_off) frame_off[n] = begin;
            if (frame_len) frame_len[n] = p - begin;
            if (out_bound) out_bound[n] = ob;
        }
        total += (ob + 255) & ~255ull;
        n++;
    }
    *n_frames = n;  // frames found before the end / the defect; more than `cap`: call again with larger arrays
    if (out_total) *out_total = total;
    return rc;
}

// ---- the cursor's C entry points (include/mzd.h)
mzd_cursor *mzd_cursor_create(void) { return new mzd_cursor(); }
void mzd_cursor_destroy(mzd_cursor *c) { delete c; }

int mzd_cursor_next(mzd_cursor *c, const uint8_t *src, uint64_t len, uint64_t max_out, uint64_t start, const int32_t hist[3],
                    uint64_t *consumed, const mzd_batch **chunk, int *last_out)
{
    if (!c || (!src && len) || !consumed || !chunk) return MZD_ERR_INVALID_ARG;
    *consumed = 0;
    *chunk = nullptr;
    if (last_out) *last_out = 0;
    if (c->status) return c->status;
    if (c->last_done) return MZD_ERR_OUT_OF_BLOCKS;  // framedecompressor.go:196
    uint64_t p = 0;
    FramePart part;
    FrameParser fp(src, 0, len, part, false);
    if (!c->header_done) {
        uint8_t fhd = 0;
        const int rc = fp.parse_header(p, fhd);
        if (rc == MZD_ERR_TRUNCATED) return MZD_OK;  // not all of the header yet (at most 18 bytes): nothing consumed
        if (rc) return c->status = rc;
        c->header_done = true;
        c->window = part.window_size;
        c->content = part.content_size;
        c->has_checksum = (fhd >> 2) & 1;
    }
    // the tables in force at the chunk's start come first in its batch
    fp.prev_ll = carry_fse(c->carry, c->prev_ll, part);
    fp.prev_of = carry_fse(c->carry, c->prev_of, part);
    fp.prev_ml = carry_fse(c->carry, c->prev_ml, part);
    fp.prev_huf = carry_huf(c->carry, c->prev_huf, part);
    bool last = false;
    uint64_t bound = 0;
    // which blocks: the next ones that are here whole, as long as the chunk has room for what they can regenerate (one at least)
    std::vector<uint64_t> pos;
    {
        uint64_t q = p, room = 0;
        while (!last) {
            if (len - q < 3) break;
            const uint32_t h = src[q] | ((uint32_t)src[q + 1] << 8) | ((uint32_t)src[q + 2] << 16);
            const int type = (h >> 1) & 3;
            const uint32_t size = h >> 3;
            if (type == 3 || size > kBlockMax) {  // (parse_block names the defect when its turn comes)
                pos.push_back(q);
                break;
            }
            const uint64_t payload = type == MZD_BLOCK_RLE ? 1u : size;
            if (len - q - 3 < payload) break;
            const uint64_t makes = type == MZD_BLOCK_COMPRESSED ? kBlockMax : size;
            if (!pos.empty() && room + makes > max_out) break;
            pos.push_back(q);
            room += makes;
            q += 3 + payload;
            last = h & 1;
        }
        last = false;
    }
    bool parsed = false;
    if (pos.size() >= 32) {
        const unsigned threads = c->threads ? c->threads : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        uint32_t carry[4] = {fp.prev_ll, fp.prev_of, fp.prev_ml, fp.prev_huf};
        if (threads > 1 && parse_blocks_parallel(src, len, pos, false, threads, part, carry, bound)) {
            fp.prev_ll = carry[0];
            fp.prev_of = carry[1];
            fp.prev_ml = carry[2];
            fp.prev_huf = carry[3];
            const uint64_t q = pos.back();
            const uint32_t h = src[q] | ((uint32_t)src[q + 1] << 8) | ((uint32_t)src[q + 2] << 16);
            last = h & 1;
            p = q + 3 + (((h >> 1) & 3) == MZD_BLOCK_RLE ? 1u : (h >> 3));
            parsed = true;
        }
    }
    if (!parsed)
        for (size_t i = 0; i < pos.size(); i++) {
            const int rc = fp.parse_block(p, last, bound);
            if (rc) return c->status = rc;
        }
    *consumed = p;
    if (part.blocks.empty()) return MZD_OK;  // (the header at most: the caller comes back with more bytes)
    if (last) {
        c->last_done = true;
        if (c->has_checksum && len - p >= 4) {  // recorded, not consumed (framereader.go:84-94)
            c->checksum = src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16) | ((uint32_t)src[p + 3] << 24);
            c->checksum_seen = true;
        }
    }
    // what the next chunk's Repeat / Treeless blocks will mean
    {
        FramePart nc;
        c->prev_ll = carry_fse(part, fp.prev_ll, nc);
        c->prev_of = carry_fse(part, fp.prev_of, nc);
        c->prev_ml = carry_fse(part, fp.prev_ml, nc);
        c->prev_huf = carry_huf(part, fp.prev_huf, nc);
        c->carry = std::move(nc);
    }
    c->blocks_done += part.blocks.size();
    c->bound_done += bound;
    part.window_size = c->window;
    part.content_size = MZD_UNKNOWN_SIZE;  // (of the whole frame: the caller checks it at the frame's end)
    part.out_bound = start + bound;
    part.consumed = p;
    c->plan = mzd_plan();
    c->plan.ext_blob = src;
    c->plan.ext_size = p;
    c->plan.merge(part, 0);
    mzd_frame_desc &fd = c->plan.frames[0];
    fd.start = start;
    fd.flags |= MZD_FRAME_CONTINUES;  // (the frame's first chunk too: the library then reports the history behind it)
    fd.hist[0] = hist ? hist[0] : 1;
    fd.hist[1] = hist ? hist[1] : 4;
    fd.hist[2] = hist ? hist[2] : 8;
    *chunk = mzd_plan_finalize(&c->plan);
    if (last_out) *last_out = last ? 1 : 0;
    return MZD_OK;
}

void mzd_cursor_set_threads(mzd_cursor *c, uint32_t n_threads)
{
    if (c) c->threads = n_threads;
}

uint64_t mzd_cursor_window(const mzd_cursor *c) { return c && c->header_done ? c->window : 0; }
uint64_t mzd_cursor_content_size(const mzd_cursor *c) { return c && c->header_done ? c->content : MZD_UNKNOWN_SIZE; }
int mzd_cursor_checksum(const mzd_cursor *c, uint32_t *checksum)
{
    if (!c || !c->checksum_seen) return 0;
    if (checksum) *checksum = c->checksum;
    return 1;
}

}  // extern "C"
