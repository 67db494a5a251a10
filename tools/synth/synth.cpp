// synth.cpp -- synthetic workload generator for the benchmark / parity tests (NOT product code).
//
// Produces the batches SURVEY.md 8(d) defines for BASELINE.json's configs:
//   config 2: single-block RAW / RLE frames of 131072 bytes
//   config 3: frames with 4-stream Huffman literals and ZERO sequences
//   config 4: text-like frames, one Compressed block: Huffman literals + FSE sequences
// It contains a small, deterministic zstd-format ENCODER written from the format as the
// reference decodes it (greedy hash matcher, repeat offsets, length-limited Huffman with direct
// or FSE-compressed weights, FSE-compressed LL/OF/ML tables).  Everything it emits decodes with
// the reference algorithm (checked against the oracle and libzstd in tests/test_synth.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ deterministic PRNG
inline uint64_t splitmix64(uint64_t &x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

// ------------------------------------------------------------------ data generators

// Text-like: vocabulary of 4096 lowercase words (2-9 letters, fixed seed), Zipf(1/(rank+1))
// sampling, space separated.
struct Vocab {
    std::vector<std::string> words;
    std::vector<double> cdf;
    std::vector<uint16_t> quant;  // 2^20-entry inverse CDF: u >> 44 -> word index (O(1) Zipf sampling)
    Vocab()
    {
        uint64_t s = 0xC0FFEE1234ull;
        double tot = 0;
        for (int i = 0; i < 4096; i++) {
            int len = 2 + (int)(splitmix64(s) % 8);
            std::string w;
            for (int j = 0; j < len; j++) w.push_back((char)('a' + splitmix64(s) % 26));
            words.push_back(w);
            tot += 1.0 / (i + 1);
            cdf.push_back(tot);
        }
        for (auto &c : cdf) c /= tot;
        quant.resize((size_t)1 << 20);
        size_t w = 0;
        for (size_t q = 0; q < quant.size(); q++) {
            const double u = ((double)q + 0.5) / (double)quant.size();
            while (w + 1 < cdf.size() && cdf[w] < u) w++;
            quant[q] = (uint16_t)w;
        }
    }
};
const Vocab &vocab()
{
    static Vocab v;
    return v;
}

void gen_text(uint64_t seed, uint8_t *dst, size_t n)
{
    const Vocab &v = vocab();
    // the seed is HASHED into the initial state: with state = seed * G + c consecutive seeds would
    // just be the same splitmix stream advanced by one draw (frames shifted by one word)
    uint64_t s0 = seed ^ 0x7E57C0DE5EEDull;
    uint64_t s = splitmix64(s0) ^ (splitmix64(s0) << 1);
    size_t p = 0;
    while (p < n) {
        const std::string &w = v.words[v.quant[splitmix64(s) >> 44]];
        const size_t l = w.size();
        if (p + l + 1 <= n) {
            memcpy(dst + p, w.data(), l);
            dst[p + l] = ' ';
            p += l + 1;
        } else {
            for (char c : w) {
                if (p < n) dst[p++] = (uint8_t)c;
            }
            if (p < n) dst[p++] = ' ';
        }
    }
}

// config 3 content: bytes min(255, floor(Exp(lambda = 0.08)))
void gen_exp(uint64_t seed, uint8_t *dst, size_t n)
{
    uint64_t s0 = seed ^ 0xE7E7E7E7A5A5ull;
    uint64_t s = splitmix64(s0) ^ (splitmix64(s0) << 1);
    for (size_t i = 0; i < n; i++) {
        double u = ((double)(splitmix64(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        double x = -std::log(u) / 0.08;
        dst[i] = (uint8_t)(x >= 255.0 ? 255 : (int)x);
    }
}

void gen_random(uint64_t seed, uint8_t *dst, size_t n)
{
    uint64_t s = seed;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t r = splitmix64(s);
        memcpy(dst + i, &r, 8);
    }
    if (i < n) {
        uint64_t r = splitmix64(s);
        memcpy(dst + i, &r, n - i);
    }
}

// position-weighted 64-bit checksum over little-endian u64 words (tail zero padded):
// sum_j w_j * (2j + 1) mod 2^64.  The same formula runs on the device output in bench.py.
uint64_t checksum64(const uint8_t *p, size_t n)
{
    uint64_t acc = 0, j = 0;
    size_t i = 0;
    for (; i + 8 <= n; i += 8, j++) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        acc += w * (2 * j + 1);
    }
    if (i < n) {
        uint64_t w = 0;
        memcpy(&w, p + i, n - i);
        acc += w * (2 * j + 1);
    }
    return acc;
}

// ------------------------------------------------------------------ bit writers

struct FwdWriter {  // LSB-first (FSE table descriptions)
    std::vector<uint8_t> &out;
    uint64_t acc = 0;
    int nb = 0;
    explicit FwdWriter(std::vector<uint8_t> &o) : out(o) {}
    void put(uint32_t v, int n)
    {
        if (n == 0) return;
        acc |= (uint64_t)(v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1))) << nb;
        nb += n;
        while (nb >= 8) {
            out.push_back((uint8_t)acc);
            acc >>= 8;
            nb -= 8;
        }
    }
    void finish()
    {
        if (nb > 0) out.push_back((uint8_t)acc);
        acc = 0;
        nb = 0;
    }
    // backward-read streams end with a 1 marker bit then zero padding
    void finish_with_marker()
    {
        put(1, 1);
        finish();
    }
};

// ------------------------------------------------------------------ FSE

struct FseEnc {
    int log = 0;
    std::vector<int16_t> norm;          // per symbol, >= 0 here (never -1)
    std::vector<uint16_t> state_table;  // tableU16
    std::vector<int32_t> delta_nb, delta_find;
};

// counts -> normalized counts summing to 1 << log, every present symbol >= 1
void fse_normalize(const std::vector<uint32_t> &count, int log, std::vector<int16_t> &norm)
{
    const uint32_t size = 1u << log;
    uint64_t total = 0;
    for (uint32_t c : count) total += c;
    norm.assign(count.size(), 0);
    int64_t sum = 0;
    size_t big = 0;
    for (size_t s = 0; s < count.size(); s++) {
        if (!count[s]) continue;
        int64_t v = (int64_t)((uint64_t)count[s] * size / total);
        if (v < 1) v = 1;
        norm[s] = (int16_t)v;
        sum += v;
        if (count[s] > count[big] || !count[big]) big = s;
    }
    int64_t diff = (int64_t)size - sum;
    while (diff != 0) {
        // give / take at the symbol with the largest normalized count
        size_t m = 0;
        for (size_t s = 0; s < norm.size(); s++)
            if (norm[s] > norm[m]) m = s;
        if (diff > 0) {
            norm[m] = (int16_t)(norm[m] + diff);
            diff = 0;
        } else {
            int64_t take = std::min<int64_t>(-diff, norm[m] - 1);
            if (take <= 0) break;  // cannot happen when #present symbols <= size
            // do not flatten the largest below the second largest in one go: take at most half
            take = std::max<int64_t>(1, std::min<int64_t>(take, norm[m] / 2));
            norm[m] = (int16_t)(norm[m] - take);
            diff += take;
        }
    }
}

void fse_build_enc(FseEnc &e)
{
    const int size = 1 << e.log;
    const int nsym = (int)e.norm.size();
    std::vector<uint8_t> spread((size_t)size);
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < e.norm[s]; i++) {
            spread[(size_t)pos] = (uint8_t)s;
            pos = (pos + step) & mask;
        }
    std::vector<int> cumul((size_t)nsym + 1, 0);
    for (int s = 0; s < nsym; s++) cumul[(size_t)s + 1] = cumul[(size_t)s] + e.norm[s];
    e.state_table.assign((size_t)size, 0);
    {
        std::vector<int> c(cumul.begin(), cumul.end());
        for (int u = 0; u < size; u++) e.state_table[(size_t)c[spread[(size_t)u]]++] = (uint16_t)(size + u);
    }
    e.delta_nb.assign((size_t)nsym, 0);
    e.delta_find.assign((size_t)nsym, 0);
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        int n = e.norm[s];
        if (n == 0) {
            e.delta_nb[(size_t)s] = ((e.log + 1) << 16) - size;
        } else if (n == 1) {
            e.delta_nb[(size_t)s] = (e.log << 16) - size;
            e.delta_find[(size_t)s] = total - 1;
            total += 1;
        } else {
            int max_bits_out = e.log - highbit((uint32_t)n - 1);
            int min_state_plus = n << max_bits_out;
            e.delta_nb[(size_t)s] = (max_bits_out << 16) - min_state_plus;
            e.delta_find[(size_t)s] = total - n;
            total += n;
        }
    }
}

struct FseState {
    const FseEnc *t;
    uint32_t value;
    void init(const FseEnc &e, int sym)  // first symbol: no bits
    {
        t = &e;
        uint32_t nb = (uint32_t)(e.delta_nb[(size_t)sym] + (1 << 15)) >> 16;
        value = (nb << 16) - (uint32_t)e.delta_nb[(size_t)sym];
        value = e.state_table[(size_t)((value >> nb) + (uint32_t)e.delta_find[(size_t)sym])];
    }
    void encode(FwdWriter &w, int sym)
    {
        uint32_t nb = (uint32_t)(value + (uint32_t)t->delta_nb[(size_t)sym]) >> 16;
        w.put(value, (int)nb);
        value = t->state_table[(size_t)((value >> nb) + (uint32_t)t->delta_find[(size_t)sym])];
    }
    void flush(FwdWriter &w) { w.put(value, t->log); }
};

// inverse of the reference's ReadTabledescriptionFromBitstream (fse.go:28-130)
void fse_write_description(const FseEnc &e, std::vector<uint8_t> &out)
{
    std::vector<uint8_t> tmp;
    FwdWriter w(tmp);
    w.put((uint32_t)(e.log - 5), 4);
    int remaining = 1 << e.log;
    const int nsym = (int)e.norm.size();
    int s = 0;
    while (remaining > 0 && s < nsym) {
        int nb = highbit((uint32_t)remaining + 1) + 1;
        uint32_t lower = (1u << (nb - 1)) - 1;
        uint32_t thresh = (1u << nb) - 1 - (uint32_t)(remaining + 1);
        uint32_t v = (uint32_t)e.norm[(size_t)s] + 1;  // value = probability + 1
        if (v < thresh) w.put(v, nb - 1);
        else if (v <= lower) w.put(v, nb);
        else w.put(v + thresh, nb);
        remaining -= e.norm[(size_t)s];
        const bool was_zero = e.norm[(size_t)s] == 0;
        s++;
        if (was_zero) {
            int z = 0;
            while (s + z < nsym && e.norm[(size_t)(s + z)] == 0) z++;
            s += z;
            while (z >= 3) {
                w.put(3, 2);
                z -= 3;
            }
            w.put((uint32_t)z, 2);
        }
    }
    w.finish();
    out.insert(out.end(), tmp.begin(), tmp.end());
}

// ------------------------------------------------------------------ Huffman

struct HufEnc {
    int max_bits = 0;
    int max_sym = 0;
    uint8_t len[256];
    uint16_t code[256];
};

// code lengths limited to `limit` bits with an exactly full Kraft sum
bool huf_build_lengths(const uint32_t *count, int limit, HufEnc &h)
{
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> live;
    for (int s = 0; s < 256; s++)
        if (count[s]) {
            nodes.push_back({count[s], -1 - s, 0});
            live.push_back((int)nodes.size() - 1);
        }
    if (live.size() < 2) return false;
    auto cmp = [&](int a, int b) { return nodes[(size_t)a].w > nodes[(size_t)b].w || (nodes[(size_t)a].w == nodes[(size_t)b].w && a < b); };
    std::make_heap(live.begin(), live.end(), cmp);
    while (live.size() > 1) {
        std::pop_heap(live.begin(), live.end(), cmp);
        int a = live.back();
        live.pop_back();
        std::pop_heap(live.begin(), live.end(), cmp);
        int b = live.back();
        live.pop_back();
        nodes.push_back({nodes[(size_t)a].w + nodes[(size_t)b].w, a, b});
        live.push_back((int)nodes.size() - 1);
        std::push_heap(live.begin(), live.end(), cmp);
    }
    memset(h.len, 0, sizeof h.len);
    // depth first
    std::vector<std::pair<int, int>> st{{live[0], 0}};
    while (!st.empty()) {
        auto [n, d] = st.back();
        st.pop_back();
        if (nodes[(size_t)n].l < 0) {
            h.len[-1 - nodes[(size_t)n].l] = (uint8_t)std::max(d, 1);
        } else {
            st.push_back({nodes[(size_t)n].l, d + 1});
            st.push_back({nodes[(size_t)n].r, d + 1});
        }
    }
    // length limit: clamp, then repair the Kraft sum (in units of 2^-limit)
    int64_t kraft = 0;
    for (int s = 0; s < 256; s++)
        if (h.len[s]) {
            if (h.len[s] > limit) h.len[s] = (uint8_t)limit;
            kraft += (int64_t)1 << (limit - h.len[s]);
        }
    const int64_t full = (int64_t)1 << limit;
    // too full: lengthen the cheapest (least frequent) symbols that are shorter than the limit
    while (kraft > full) {
        int best = -1;
        for (int s = 0; s < 256; s++)
            if (h.len[s] && h.len[s] < limit && (best < 0 || count[s] < count[best] || (count[s] == count[best] && h.len[s] > h.len[best]))) best = s;
        if (best < 0) return false;
        kraft -= (int64_t)1 << (limit - h.len[best] - 1);
        h.len[best]++;
    }
    // not full: shorten the most frequent symbols whose shortening still fits
    while (kraft < full) {
        int best = -1;
        for (int s = 0; s < 256; s++)
            if (h.len[s] > 1 && kraft + ((int64_t)1 << (limit - h.len[s])) <= full &&
                (best < 0 || count[s] > count[best]))
                best = s;
        if (best < 0) return false;
        kraft += (int64_t)1 << (limit - h.len[best]);
        h.len[best]--;
    }
    h.max_bits = 0;
    h.max_sym = 0;
    for (int s = 0; s < 256; s++)
        if (h.len[s]) {
            h.max_bits = std::max<int>(h.max_bits, h.len[s]);
            h.max_sym = s;
        }
    // canonical codes in the decoder's table order (huffman.go:163-187): longest first from
    // cell 0, ascending symbol; code = cell index >> (max_bits - len)
    uint32_t cell = 0;
    for (int l = h.max_bits; l >= 1; l--)
        for (int s = 0; s < 256; s++)
            if (h.len[s] == l) {
                h.code[s] = (uint16_t)(cell >> (h.max_bits - l));
                cell += 1u << (h.max_bits - l);
            }
    return cell == (1u << h.max_bits);
}

// decode of FSE-compressed weights exactly as the reference does (fse.go:307-390), used as a
// self-check of the encoder's end-of-stream handling
bool weights_roundtrip(const std::vector<uint8_t> &desc_and_stream, size_t desc_len, const FseEnc &e,
                       const std::vector<uint8_t> &want)
{
    // decoding table
    const int size = 1 << e.log;
    std::vector<uint8_t> sym((size_t)size);
    std::vector<uint8_t> nbv((size_t)size);
    std::vector<uint16_t> base((size_t)size);
    {
        const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
        int pos = 0;
        for (size_t s = 0; s < e.norm.size(); s++)
            for (int i = 0; i < e.norm[s]; i++) {
                sym[(size_t)pos] = (uint8_t)s;
                pos = (pos + step) & mask;
            }
        std::vector<uint32_t> next(e.norm.begin(), e.norm.end());
        for (int i = 0; i < size; i++) {
            uint32_t n = next[sym[(size_t)i]]++;
            int nb = e.log - highbit(n);
            nbv[(size_t)i] = (uint8_t)nb;
            base[(size_t)i] = (uint16_t)((n << nb) - (uint32_t)size);
        }
    }
    const uint8_t *p = desc_and_stream.data() + desc_len;
    int64_t cursor = (int64_t)(desc_and_stream.size() - desc_len) * 8 - 1;
    auto rd = [&](int n) {
        uint32_t v = 0;
        for (int i = 0; i < n; i++) {
            int64_t b = cursor - i;
            v = (v << 1) | (b >= 0 ? (uint32_t)((p[b >> 3] >> (b & 7)) & 1) : 0u);
        }
        cursor -= n;
        return v;
    };
    int pad = 0;
    while (rd(1) == 0)
        if (++pad >= 8) return false;
    uint32_t st[2] = {rd(e.log), rd(e.log)};
    std::vector<uint8_t> got;
    for (int turn = 0;; turn ^= 1) {
        got.push_back(sym[st[turn]]);
        st[turn] = base[st[turn]] + rd(nbv[st[turn]]);
        if (cursor < -1) {
            got.push_back(sym[st[turn ^ 1]]);
            break;
        }
        if (got.size() > 300) return false;
    }
    return got == want;
}

// Huffman tree description (huffman.go:40-107): direct nibbles when possible, else FSE weights
bool huf_write_tree(const HufEnc &h, std::vector<uint8_t> &out)
{
    const int nw = h.max_sym;  // weights for symbols 0 .. max_sym-1; the last one is implied
    std::vector<uint8_t> w((size_t)nw);
    for (int s = 0; s < nw; s++) w[(size_t)s] = h.len[s] ? (uint8_t)(h.max_bits + 1 - h.len[s]) : 0;
    // try FSE-compressed weights
    if (nw >= 2) {
        std::vector<uint32_t> cnt(13, 0);
        for (uint8_t x : w) cnt[x]++;
        int maxw = 12;
        while (maxw > 0 && !cnt[(size_t)maxw]) maxw--;
        cnt.resize((size_t)maxw + 1);
        int present = 0;
        for (uint32_t c : cnt) present += c != 0;
        if (present >= 2) {
            for (int log = 6; log >= 5; log--) {
                FseEnc e;
                e.log = log;
                fse_normalize(cnt, log, e.norm);
                fse_build_enc(e);
                std::vector<uint8_t> body;
                fse_write_description(e, body);
                const size_t desc_len = body.size();
                std::vector<uint8_t> stream;
                FwdWriter bw(stream);
                // symbol i belongs to state (i & 1): state 0 is decoded first (fse.go:330-336)
                FseState s0, s1;
                int i = nw - 1;
                FseState *stt[2] = {&s0, &s1};
                stt[i & 1]->init(e, w[(size_t)i]);
                i--;
                stt[i & 1]->init(e, w[(size_t)i]);
                i--;
                for (; i >= 0; i--) stt[i & 1]->encode(bw, w[(size_t)i]);
                s1.flush(bw);
                s0.flush(bw);
                bw.finish_with_marker();
                body.insert(body.end(), stream.begin(), stream.end());
                if (body.size() < 128 && weights_roundtrip(body, desc_len, e, w) &&
                    (nw > 128 || body.size() < (size_t)(nw + 1) / 2)) {
                    out.push_back((uint8_t)body.size());
                    out.insert(out.end(), body.begin(), body.end());
                    return true;
                }
            }
        }
    }
    if (nw > 128) return false;
    out.push_back((uint8_t)(127 + nw));
    for (int i = 0; i < nw; i += 2) out.push_back((uint8_t)((w[(size_t)i] << 4) | (i + 1 < nw ? w[(size_t)i + 1] : 0)));
    return true;
}

// one Huffman stream, written so that the backward reader sees the FIRST symbol first
void huf_encode_stream(const HufEnc &h, const uint8_t *src, size_t n, std::vector<uint8_t> &out)
{
    std::vector<uint8_t> tmp;
    tmp.reserve(n);
    FwdWriter w(tmp);
    for (size_t i = n; i-- > 0;) w.put(h.code[src[i]], h.len[src[i]]);
    w.finish_with_marker();
    out.insert(out.end(), tmp.begin(), tmp.end());
}

// literals section (literals.go:67-371).  Returns false if Huffman is not applicable.
bool write_literals_huf(const uint8_t *lit, size_t n, std::vector<uint8_t> &out)
{
    if (n < 64) return false;
    uint32_t count[256] = {0};
    for (size_t i = 0; i < n; i++) count[lit[i]]++;
    HufEnc h;
    if (!huf_build_lengths(count, 11, h)) return false;
    std::vector<uint8_t> body;
    if (!huf_write_tree(h, body)) return false;
    const size_t normal = (n + 3) / 4;
    std::vector<uint8_t> streams[4];
    for (int k = 0; k < 4; k++) {
        size_t b = (size_t)k * normal, e = k < 3 ? b + normal : n;
        huf_encode_stream(h, lit + b, e - b, streams[k]);
    }
    for (int k = 0; k < 3; k++) {
        if (streams[k].size() > 0xFFFF) return false;
        body.push_back((uint8_t)streams[k].size());
        body.push_back((uint8_t)(streams[k].size() >> 8));
    }
    for (int k = 0; k < 4; k++) body.insert(body.end(), streams[k].begin(), streams[k].end());
    const size_t csize = body.size();
    if (csize >= n) return false;
    // header: type 2 (Compressed), 4 streams; size format by magnitude (literals.go:130-151)
    uint64_t hdr;
    int hbytes;
    if (n < 1024 && csize < 1024) { hdr = 2 | (1 << 2) | ((uint64_t)n << 4) | ((uint64_t)csize << 14); hbytes = 3; }
    else if (n < 16384 && csize < 16384) { hdr = 2 | (2 << 2) | ((uint64_t)n << 4) | ((uint64_t)csize << 18); hbytes = 4; }
    else { hdr = 2 | (3 << 2) | ((uint64_t)n << 4) | ((uint64_t)csize << 22); hbytes = 5; }
    for (int i = 0; i < hbytes; i++) out.push_back((uint8_t)(hdr >> (8 * i)));
    out.insert(out.end(), body.begin(), body.end());
    return true;
}

void write_literals_raw(const uint8_t *lit, size_t n, std::vector<uint8_t> &out)
{
    if (n < 32) out.push_back((uint8_t)(0 | (n << 3)));
    else if (n < 4096) { out.push_back((uint8_t)(0 | (1 << 2) | ((n & 15) << 4))); out.push_back((uint8_t)(n >> 4)); }
    else { out.push_back((uint8_t)(0 | (3 << 2) | ((n & 15) << 4))); out.push_back((uint8_t)(n >> 4)); out.push_back((uint8_t)(n >> 12)); }
    out.insert(out.end(), lit, lit + n);
}

void write_literals_rle(uint8_t b, size_t n, std::vector<uint8_t> &out)
{
    if (n < 32) out.push_back((uint8_t)(1 | (n << 3)));
    else if (n < 4096) { out.push_back((uint8_t)(1 | (1 << 2) | ((n & 15) << 4))); out.push_back((uint8_t)(n >> 4)); }
    else { out.push_back((uint8_t)(1 | (3 << 2) | ((n & 15) << 4))); out.push_back((uint8_t)(n >> 4)); out.push_back((uint8_t)(n >> 12)); }
    out.push_back(b);
}

void write_literals(const uint8_t *lit, size_t n, std::vector<uint8_t> &out)
{
    bool same = n > 0;
    for (size_t i = 1; i < n && same; i++) same = lit[i] == lit[0];
    if (same && n > 1) return write_literals_rle(lit[0], n, out);
    if (!write_literals_huf(lit, n, out)) write_literals_raw(lit, n, out);
}

// ------------------------------------------------------------------ sequences

struct Seq { uint32_t ll, ml, ofv; };

const uint32_t kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64,
                              0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
const uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const uint32_t kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                              30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099,
                              8195, 16387, 32771, 65539};
const uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                             1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

int ll_code(uint32_t ll)
{
    int c = 35;
    while (kLLBase[c] > ll) c--;
    return c;
}
int ml_code(uint32_t ml)
{
    int c = 52;
    while (kMLBase[c] > ml) c--;
    return c;
}

// returns mode byte contribution and appends the table description
int prepare_table(const std::vector<uint8_t> &codes, int max_log, int nsym_max, FseEnc &e, std::vector<uint8_t> &desc)
{
    std::vector<uint32_t> cnt((size_t)nsym_max, 0);
    for (uint8_t c : codes) cnt[c]++;
    int maxs = nsym_max - 1;
    while (maxs > 0 && !cnt[(size_t)maxs]) maxs--;
    cnt.resize((size_t)maxs + 1);
    int present = 0;
    for (uint32_t c : cnt) present += c != 0;
    if (present == 1) {  // RLE mode (sequences.go:282-289)
        desc.push_back((uint8_t)maxs);
        e.log = 0;
        return 1;
    }
    const uint32_t n = (uint32_t)codes.size();
    int log = std::min(max_log, std::max(5, highbit(n - 1) - 2));
    int min_log = std::min(highbit(n) + 1, highbit((uint32_t)maxs) + 2);
    log = std::max(log, min_log);
    while ((1 << log) < present) log++;
    log = std::min(log, max_log);
    e.log = log;
    fse_normalize(cnt, log, e.norm);
    fse_build_enc(e);
    fse_write_description(e, desc);
    return 2;  // Compressed mode
}

void write_sequences(const std::vector<Seq> &seqs, std::vector<uint8_t> &out)
{
    const size_t n = seqs.size();
    if (n == 0) {
        out.push_back(0);
        return;
    }
    if (n < 128) out.push_back((uint8_t)n);
    else if (n < 0x7F00) { out.push_back((uint8_t)((n >> 8) + 128)); out.push_back((uint8_t)n); }
    else { out.push_back(255); out.push_back((uint8_t)(n - 0x7F00)); out.push_back((uint8_t)((n - 0x7F00) >> 8)); }
    std::vector<uint8_t> llc(n), mlc(n), ofc(n);
    for (size_t i = 0; i < n; i++) {
        llc[i] = (uint8_t)ll_code(seqs[i].ll);
        mlc[i] = (uint8_t)ml_code(seqs[i].ml);
        ofc[i] = (uint8_t)highbit(seqs[i].ofv);
    }
    FseEnc ell, eof, eml;
    std::vector<uint8_t> dll, dof, dml;
    int mll = prepare_table(llc, 9, 36, ell, dll);
    int mof = prepare_table(ofc, 8, 32, eof, dof);
    int mml = prepare_table(mlc, 9, 53, eml, dml);
    out.push_back((uint8_t)((mll << 6) | (mof << 4) | (mml << 2)));
    out.insert(out.end(), dll.begin(), dll.end());
    out.insert(out.end(), dof.begin(), dof.end());
    out.insert(out.end(), dml.begin(), dml.end());
    std::vector<uint8_t> bits;
    bits.reserve(n * 4);
    FwdWriter w(bits);
    FseState sll{}, sof{}, sml{};
    const size_t last = n - 1;
    if (mml == 2) sml.init(eml, mlc[last]);
    if (mof == 2) sof.init(eof, ofc[last]);
    if (mll == 2) sll.init(ell, llc[last]);
    auto extras = [&](size_t i) {
        w.put(seqs[i].ll - kLLBase[llc[i]], kLLBits[llc[i]]);
        w.put(seqs[i].ml - kMLBase[mlc[i]], kMLBits[mlc[i]]);
        w.put(seqs[i].ofv - (1u << ofc[i]), ofc[i]);
    };
    extras(last);
    for (size_t i = last; i-- > 0;) {
        if (mof == 2) sof.encode(w, ofc[i]);
        if (mml == 2) sml.encode(w, mlc[i]);
        if (mll == 2) sll.encode(w, llc[i]);
        extras(i);
    }
    if (mml == 2) sml.flush(w);
    if (mof == 2) sof.flush(w);
    if (mll == 2) sll.flush(w);
    w.finish_with_marker();
    out.insert(out.end(), bits.begin(), bits.end());
}

// ------------------------------------------------------------------ matcher

inline uint32_t rd32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

// Greedy single-probe hash matcher with repeat-offset checks; emits raw offset VALUES with the
// decoder's repeat-code rules (sequence_execution.go:65-114) applied in reverse.
// `rep` is the frame's repeat-offset history: it persists across blocks (framedecompressor.go:23) and
// is updated only by blocks that carry sequences.
// The block is bytes [b0, b0 + bn) of the FRAME `src`; `table` (positions in the frame) lives as long as the frame: a block's
// matches reach back over its start into the blocks before it (ringbuffer.go:242-277 RepeatBeforeIndex), up to kMaxOffset --
// single-segment frames, the window is the content.  (A frame of one block starts with an empty table: unchanged.)
constexpr int kHashLog = 16;
static size_t g_max_offset = (size_t)1 << 27;  // (the device path takes offsets below 2^28); synth_set_max_offset: a smaller window
#define kMaxOffset g_max_offset
void find_sequences(const uint8_t *src, size_t b0, size_t bn, std::vector<int32_t> &table, std::vector<Seq> &seqs,
                    std::vector<uint8_t> &lits, int min_match, uint32_t rep[3])
{
    constexpr int HLOG = kHashLog;
    const size_t n = b0 + bn;  // the block's end
    size_t anchor = b0, ip = b0;
    const size_t limit = bn >= 8 ? n - 8 : b0;
    auto hash = [&](const uint8_t *p) { return (rd32(p) * 2654435761u) >> (32 - HLOG); };
    while (ip < limit) {
        size_t mlen = 0, mpos = ip;
        uint32_t off = 0;
        // repeat offset 0 at ip+1 (zstd-fast style)
        if (ip + 1 >= rep[0] && rd32(src + ip + 1) == rd32(src + ip + 1 - rep[0])) {
            mpos = ip + 1;
            off = rep[0];
            mlen = 4;
        } else {
            uint32_t h = hash(src + ip);
            int32_t cand = table[h];
            table[h] = (int32_t)ip;
            if (cand >= 0 && (size_t)cand < ip && ip - (size_t)cand <= kMaxOffset && rd32(src + cand) == rd32(src + ip)) {
                off = (uint32_t)(ip - (size_t)cand);
                mlen = 4;
            }
        }
        if (mlen) {
            while (mpos + mlen < n && src[mpos + mlen] == src[mpos + mlen - off]) mlen++;
            if ((int)mlen < min_match) mlen = 0;
        }
        if (!mlen) {
            ip++;
            continue;
        }
        ip = mpos;
        // extend backwards into pending literals
        while (ip > anchor && ip > off && src[ip - 1] == src[ip - 1 - off]) {
            ip--;
            mlen++;
        }
        const uint32_t ll = (uint32_t)(ip - anchor);
        uint32_t ofv;
        if (ll > 0) {
            if (off == rep[0]) ofv = 1;
            else if (off == rep[1]) { ofv = 2; std::swap(rep[0], rep[1]); }
            else if (off == rep[2]) { ofv = 3; uint32_t t = rep[2]; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = t; }
            else { ofv = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
        } else {
            if (off == rep[1]) { ofv = 1; std::swap(rep[0], rep[1]); }
            else if (off == rep[2]) { ofv = 2; uint32_t t = rep[2]; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = t; }
            else if (rep[0] > 1 && off == rep[0] - 1) { ofv = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
            else { ofv = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
        }
        lits.insert(lits.end(), src + anchor, src + ip);
        seqs.push_back(Seq{ll, (uint32_t)mlen, ofv});
        // index a couple of positions inside the match
        if (ip + 2 < limit) table[hash(src + ip + 2)] = (int32_t)(ip + 2);
        ip += mlen;
        if (ip >= 2 && ip - 2 < limit) table[hash(src + ip - 2)] = (int32_t)(ip - 2);
        anchor = ip;
    }
    lits.insert(lits.end(), src + anchor, src + n);
}

// ------------------------------------------------------------------ frames

// Optional content checksum (zstd frame format: low 32 bits of XXH64(content, 0) after the last
// block; xxHash specification).  Off by default: BASELINE's frame definitions carry none.
static bool g_content_checksum = false;
static uint64_t xrotl(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
static uint64_t xrd(const uint8_t *p, int n) { uint64_t v = 0; for (int i = n - 1; i >= 0; i--) v = (v << 8) | p[i]; return v; }
static uint64_t xxh64(const uint8_t *p, size_t n)
{
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    auto round = [&](uint64_t a, uint64_t in) { return xrotl(a + in * P2, 31) * P1; };
    const uint8_t *end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v[4] = {P1 + P2, P2, 0, 0 - P1};
        for (; end - p >= 32; p += 32)
            for (int k = 0; k < 4; k++) v[k] = round(v[k], xrd(p + 8 * k, 8));
        h = xrotl(v[0], 1) + xrotl(v[1], 7) + xrotl(v[2], 12) + xrotl(v[3], 18);
        for (int k = 0; k < 4; k++) h = (h ^ round(0, v[k])) * P1 + P4;
    } else {
        h = P5;
    }
    h += n;
    for (; end - p >= 8; p += 8) { h ^= round(0, xrd(p, 8)); h = xrotl(h, 27) * P1 + P4; }
    if (end - p >= 4) { h ^= xrd(p, 4) * P1; h = xrotl(h, 23) * P2 + P3; p += 4; }
    for (; p < end; p++) { h ^= *p * P5; h = xrotl(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

void put_frame_header(std::vector<uint8_t> &out, uint32_t content_size)
{
    // magic, FHD 0xA0 = single segment + 4-byte content size, no dictionary (+ 0x04 with a content checksum)
    const uint8_t h[9] = {0x28, 0xB5, 0x2F, 0xFD, (uint8_t)(g_content_checksum ? 0xA4 : 0xA0), (uint8_t)content_size, (uint8_t)(content_size >> 8),
                          (uint8_t)(content_size >> 16), (uint8_t)(content_size >> 24)};
    out.insert(out.end(), h, h + 9);
}
void put_block_header(std::vector<uint8_t> &out, uint32_t size, int type, bool last)
{
    uint32_t v = (size << 3) | ((uint32_t)type << 1) | (last ? 1 : 0);
    out.push_back((uint8_t)v);
    out.push_back((uint8_t)(v >> 8));
    out.push_back((uint8_t)(v >> 16));
}

// mode 0: full compressed block; 1: literals only (0 sequences); 2: raw block; 3: rle block
void encode_frame_body(const uint8_t *src, size_t n, int mode, std::vector<uint8_t> &out, uint32_t *n_seq_out, int min_match);
void encode_frame(const uint8_t *src, size_t n, int mode, std::vector<uint8_t> &out, uint32_t *n_seq_out, int min_match = 5)
{
    encode_frame_body(src, n, mode, out, n_seq_out, min_match);
    if (g_content_checksum) {
        const uint32_t c = (uint32_t)xxh64(src, n);
        for (int i = 0; i < 4; i++) out.push_back((uint8_t)(c >> (8 * i)));
    }
}
void encode_frame_body(const uint8_t *src, size_t n, int mode, std::vector<uint8_t> &out, uint32_t *n_seq_out, int min_match)
{
    put_frame_header(out, (uint32_t)n);
    if (n_seq_out) *n_seq_out = 0;
    // blocks of at most 128 KiB
    size_t p = 0;
    if (n == 0) {
        put_block_header(out, 0, 0, true);
        return;
    }
    uint32_t rep[3] = {1, 4, 8};  // framedecompressor.go:48
    std::vector<int32_t> table;  // the matcher's positions: one table per frame
    if (mode == 0) table.assign((size_t)1 << kHashLog, -1);
    while (p < n) {
        const size_t bn = std::min<size_t>(n - p, 128 * 1024);
        const bool last = p + bn == n;
        if (mode == 2) {
            put_block_header(out, (uint32_t)bn, 0, last);
            out.insert(out.end(), src + p, src + p + bn);
        } else if (mode == 3) {
            put_block_header(out, (uint32_t)bn, 1, last);
            out.push_back(src[p]);
        } else {
            std::vector<uint8_t> body;
            std::vector<Seq> seqs;
            std::vector<uint8_t> lits;
            uint32_t rep_new[3] = {rep[0], rep[1], rep[2]};
            if (mode == 0) find_sequences(src, p, bn, table, seqs, lits, min_match, rep_new);
            else lits.assign(src + p, src + p + bn);
            write_literals(lits.data(), lits.size(), body);
            write_sequences(seqs, body);
            if ((body.size() >= bn && mode == 0) || body.size() > 128 * 1024) {  // not compressible: raw block
                put_block_header(out, (uint32_t)bn, 0, last);
                out.insert(out.end(), src + p, src + p + bn);
            } else {
                put_block_header(out, (uint32_t)body.size(), 2, last);
                out.insert(out.end(), body.begin(), body.end());
                if (n_seq_out) *n_seq_out += (uint32_t)seqs.size();
                rep[0] = rep_new[0]; rep[1] = rep_new[1]; rep[2] = rep_new[2];  // history moves only with an emitted block
            }
        }
        p += bn;
    }
}

}  // namespace

extern "C" {

// frames produced from now on carry a content checksum (FHD bit 2 + 4 bytes after the last block)
void synth_set_content_checksum(int on) { g_content_checksum = on != 0; }

// matches of the frames produced from now on reach back at most `n` bytes (what zstd's windowLog does: 2^23 at its levels up
// to 19); 0 restores the default, 2^27
void synth_set_max_offset(uint64_t n) { g_max_offset = n ? (size_t)std::min<uint64_t>(n, (uint64_t)1 << 27) : (size_t)1 << 27; }

// kinds of content
enum { SYNTH_TEXT = 0, SYNTH_EXP = 1, SYNTH_RANDOM = 2, SYNTH_ZERO = 3 };

void synth_generate(int kind, uint64_t seed, uint8_t *dst, uint64_t n)
{
    if (kind == SYNTH_TEXT) gen_text(seed, dst, n);
    else if (kind == SYNTH_EXP) gen_exp(seed, dst, n);
    else if (kind == SYNTH_RANDOM) gen_random(seed, dst, n);
    else memset(dst, 0, n);
}

uint64_t synth_checksum64(const uint8_t *p, uint64_t n) { return checksum64(p, n); }

// Compress one buffer into one frame. mode: 0 full, 1 literals-only, 2 raw, 3 rle (first byte).
// Returns the frame size or 0 if cap is too small.
uint64_t synth_compress2(const uint8_t *src, uint64_t n, int mode, int min_match, uint8_t *dst, uint64_t cap, uint32_t *n_seq)
{
    std::vector<uint8_t> out;
    out.reserve(n / 2 + 64);
    encode_frame(src, n, mode, out, n_seq, min_match);
    if (out.size() > cap) return 0;
    memcpy(dst, out.data(), out.size());
    return out.size();
}

uint64_t synth_compress(const uint8_t *src, uint64_t n, int mode, uint8_t *dst, uint64_t cap, uint32_t *n_seq)
{
    std::vector<uint8_t> out;
    out.reserve(n / 2 + 64);
    encode_frame(src, n, mode, out, n_seq);
    if (out.size() > cap) return 0;
    memcpy(dst, out.data(), out.size());
    return out.size();
}

// BASELINE configs.  Frame i of config c is a pure function of (c, i):
//   config 2: even i raw block of splitmix64(0x5EED0000 + i), odd i rle of byte (37 i + 11) & 255
//   config 3: exp bytes, literals-only block
//   config 4: text, full block
// Writes frames back to back into blob (capacity blob_cap), fills off/len/checksum/n_seq
// (checksum = checksum64 of the ORIGINAL content).  Returns total bytes or 0 on overflow.
uint64_t synth_make_batch(int config, uint64_t first_frame, uint32_t count, uint32_t frame_bytes, uint8_t *blob,
                          uint64_t blob_cap, uint64_t *off, uint64_t *len, uint64_t *checksum, uint32_t *n_seq,
                          uint32_t threads)
{
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = std::min<uint32_t>(threads, std::max<uint32_t>(1, count));
    std::vector<std::vector<uint8_t>> frames(count);
    auto work = [&](uint32_t t) {
        std::vector<uint8_t> content(frame_bytes);
        for (uint32_t k = t; k < count; k += threads) {
            const uint64_t i = first_frame + k;
            int mode;
            if (config == 2) {
                if (i & 1) {
                    memset(content.data(), (int)((37 * i + 11) & 255), frame_bytes);
                    mode = 3;
                } else {
                    gen_random(0x5EED0000ull + i, content.data(), frame_bytes);
                    mode = 2;
                }
            } else if (config == 3) {
                gen_exp(i, content.data(), frame_bytes);
                mode = 1;
            } else {
                gen_text(i, content.data(), frame_bytes);
                mode = 0;
            }
            uint32_t ns = 0;
            encode_frame(content.data(), frame_bytes, mode, frames[k], &ns);
            checksum[k] = checksum64(content.data(), frame_bytes);
            if (n_seq) n_seq[k] = ns;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    uint64_t at = 0;
    for (uint32_t k = 0; k < count; k++) {
        if (at + frames[k].size() > blob_cap) return 0;
        memcpy(blob + at, frames[k].data(), frames[k].size());
        off[k] = at;
        len[k] = frames[k].size();
        at += frames[k].size();
        std::vector<uint8_t>().swap(frames[k]);
    }
    return at;
}

}  // extern "C"
