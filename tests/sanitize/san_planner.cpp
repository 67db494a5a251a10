// san_planner.cpp -- ASan / UBSan driver for the product's HOST code that parses untrusted bytes:
// the planner (sparkzstd_amd/csrc/planner.cpp: frame / block / section headers, FSE descriptions,
// Huffman weights, table builds) and mzd_split_frames.  CPU only, no HIP: planner.cpp is plain C++.
// Error model being exercised (the planner returns a status, it never faults):
//   structure/literals.go:43-44,206-207, fse/fse.go:133, decompression/framedecompressor.go:90.
// usage: san_planner <n_mutations_per_frame> file.zst...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mzd.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// reads everything the batch view points at (ASan checks every access against the planner's own allocations)
static uint64_t walk(const mzd_batch *b)
{
    uint64_t acc = b->in_size + b->out_size;
    for (uint32_t i = 0; i < b->n_frames; i++) acc += b->frames[i].first_block + b->frames[i].n_blocks + b->frames[i].out_offset;
    for (uint32_t i = 0; i < b->n_blocks; i++) {
        const mzd_block_desc &d = b->blocks[i];
        acc += d.type + d.size + d.lit_regen + d.n_seq;
        if (d.type == MZD_BLOCK_RAW && d.size) acc += b->in[d.src_off] + b->in[d.src_off + d.size - 1];
        if (d.type == MZD_BLOCK_RLE) acc += b->in[d.src_off];
        if (d.type == MZD_BLOCK_COMPRESSED) {
            if (d.lit_type == MZD_LIT_HUF) {
                uint64_t n = 0;
                for (int s = 0; s < (d.lit_streams == 4 ? 4 : 1); s++) n += d.lit_stream_size[s];
                if (n) acc += b->in[d.lit_off] + b->in[d.lit_off + n - 1];
                acc += b->huf_tables[d.huf_table].max_bits;
            } else if (d.lit_type == MZD_LIT_RAW && d.lit_regen) {
                acc += b->in[d.lit_off + d.lit_regen - 1];
            } else if (d.lit_type == MZD_LIT_RLE) {
                acc += b->in[d.lit_off];
            }
            if (d.n_seq) {
                acc += b->in[d.seq_off] + b->in[d.seq_off + d.seq_size - 1];
                acc += b->fse_tables[d.ll_table].acc_log + b->fse_tables[d.of_table].acc_log + b->fse_tables[d.ml_table].acc_log;
            }
        }
    }
    for (uint32_t i = 0; i < b->n_fse_tables; i++) {
        const mzd_fse_table_desc &t = b->fse_tables[i];
        const uint32_t n = (t.build & MZD_FSE_FROM_COUNTS) ? ((t.build & 0xFF) + 1) / 2 : 1u << t.acc_log;
        for (uint32_t j = 0; j < n; j++) acc += b->fse_entries[t.entries_off + j].baseline;
    }
    for (uint32_t i = 0; i < b->n_huf_tables; i++) {
        const mzd_huf_table_desc &t = b->huf_tables[i];
        const uint32_t n = (t.max_bits & MZD_HUF_FROM_WEIGHTS) ? (((t.max_bits >> 8) & 0xFF) + 1) / 2 : 1u << (t.max_bits & 0xFF);
        for (uint32_t j = 0; j < n; j++) acc += b->huf_entries[t.entries_off + j].nbits;
    }
    return acc;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int n_mut = atoi(argv[1]);
    std::vector<std::vector<uint8_t>> frames;
    for (int i = 2; i < argc; i++) {
        FILE *f = fopen(argv[i], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[i]); return 2; }
        std::vector<uint8_t> v;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
        fclose(f);
        frames.push_back(std::move(v));
    }
    uint64_t acc = 0, n_ok = 0, n_bad = 0, n_split_ok = 0;
    for (int device_tables = 0; device_tables < 2; device_tables++) {
        mzd_plan *p = mzd_plan_create();
        mzd_plan_set_device_tables(p, device_tables);
        // 1. the intact corpus: every frame must plan
        for (auto &f : frames) {
            uint64_t used = 0;
            // exact-size heap copy: one byte past the frame is an ASan error
            uint8_t *c = (uint8_t *)malloc(f.size());
            memcpy(c, f.data(), f.size());
            const int rc = mzd_plan_add_frame(p, c, f.size(), &used);
            free(c);
            if (rc != MZD_OK || used > f.size()) { fprintf(stderr, "intact frame failed: %d\n", rc); return 1; }
        }
        acc += walk(mzd_plan_finalize(p));
        mzd_plan_reset(p);
        // 2. mutated frames (tools/plan_soak.py's generator: truncation, 1-3 byte flips, headers included)
        for (size_t fi = 0; fi < frames.size(); fi++) {
            for (int m = 0; m < n_mut; m++) {
                std::vector<uint8_t> b = frames[fi];
                const double r = (double)(rnd() >> 11) / 9007199254740992.0;
                if (r < 0.2) b.resize(rnd() % (b.size() + 1));
                else {
                    const size_t lo = (rnd() % 10 < 3) ? 0 : 4;
                    const int flips = 1 + (int)(rnd() % 3);
                    for (int k = 0; k < flips && b.size() > lo; k++) b[lo + rnd() % (b.size() - lo)] ^= (uint8_t)(1 + rnd() % 255);
                }
                uint8_t *c = (uint8_t *)malloc(b.size() ? b.size() : 1);
                if (!b.empty()) memcpy(c, b.data(), b.size());
                uint64_t used = 0;
                const int rc = mzd_plan_add_frame(p, c, b.size(), &used);
                free(c);
                (rc == MZD_OK ? n_ok : n_bad)++;
                if (rc == MZD_OK && used > b.size()) { fprintf(stderr, "consumed beyond the frame\n"); return 1; }
            }
            if ((fi & 7) == 7) {
                acc += walk(mzd_plan_finalize(p));
                mzd_plan_reset(p);
            }
        }
        acc += walk(mzd_plan_finalize(p));
        mzd_plan_reset(p);
        // 3. every prefix of the small frames, sampled prefixes of the large ones
        for (auto &f : frames) {
            const size_t step = f.size() <= 512 ? 1 : f.size() / 97 + 1;
            for (size_t n = 0; n < f.size(); n += step) {
                uint8_t *c = (uint8_t *)malloc(n ? n : 1);
                memcpy(c, f.data(), n);
                uint64_t used = 0;
                const int rc = mzd_plan_add_frame(p, c, n, &used);
                free(c);
                if (rc == MZD_OK && used > n) { fprintf(stderr, "prefix %zu accepted past its end\n", n); return 1; }
            }
            acc += walk(mzd_plan_finalize(p));
            mzd_plan_reset(p);
        }
        // 4. many frames on several threads
        {
            std::vector<uint8_t> blob;
            std::vector<uint64_t> off, len;
            for (auto &f : frames) { off.push_back(blob.size()); len.push_back(f.size()); blob.insert(blob.end(), f.begin(), f.end()); }
            if (mzd_plan_add_frames(p, blob.data(), off.data(), len.data(), (uint32_t)off.size(), 4) != MZD_OK) { fprintf(stderr, "threaded plan failed\n"); return 1; }
            acc += walk(mzd_plan_finalize(p));
        }
        mzd_plan_destroy(p);
    }
    // 5. mzd_split_frames: concatenations with skippable frames in between, intact / mutated / truncated
    for (int round = 0; round < 40 + 4 * n_mut; round++) {
        std::vector<uint8_t> blob;
        const int k = 1 + (int)(rnd() % 6);
        for (int i = 0; i < k; i++) {
            if (rnd() % 4 == 0) {
                const uint32_t sk = (uint32_t)(rnd() % 40);
                const uint8_t h[8] = {(uint8_t)(0x50 + rnd() % 16), 0x2A, 0x4D, 0x18, (uint8_t)sk, 0, 0, 0};
                blob.insert(blob.end(), h, h + 8);
                for (uint32_t j = 0; j < sk; j++) blob.push_back((uint8_t)rnd());
            }
            auto &f = frames[rnd() % frames.size()];
            blob.insert(blob.end(), f.begin(), f.end());
        }
        if (round % 3 == 1 && !blob.empty()) blob.resize(rnd() % blob.size());
        if (round % 3 == 2)
            for (int j = 0; j < 3 && !blob.empty(); j++) blob[rnd() % blob.size()] ^= (uint8_t)(1 + rnd() % 255);
        uint8_t *c = (uint8_t *)malloc(blob.size() ? blob.size() : 1);
        if (!blob.empty()) memcpy(c, blob.data(), blob.size());
        std::vector<uint64_t> off(4), len(4), ob(4);  // deliberately smaller than the number of frames sometimes
        uint32_t n = 0;
        uint64_t total = 0;
        const int rc = mzd_split_frames(c, blob.size(), off.data(), len.data(), ob.data(), 4, &n, &total);
        for (uint32_t i = 0; i < n && i < 4; i++)
            if (off[i] + len[i] > blob.size()) { fprintf(stderr, "split: frame beyond the blob\n"); return 1; }
        if (rc == MZD_OK && round % 3 == 0) n_split_ok++;
        if (round % 3 == 0 && rc != MZD_OK) { fprintf(stderr, "split of intact frames failed: %d\n", rc); return 1; }
        free(c);
        acc += total + n;
    }
    // 6. the cursor (one frame in chunks of whole blocks, ABI 9): intact and mutated frames, the source handed over in exact-size heap
    // copies of what is left, small and large chunk bounds, the blocks of a chunk parsed serially and in ranges on eight threads
    uint64_t n_chunks = 0, n_cur_bad = 0;
    for (int threads = 1; threads <= 8; threads += 7) {
        for (size_t fi = 0; fi < frames.size(); fi++) {
            for (int m = 0; m < 1 + n_mut / 4; m++) {
                std::vector<uint8_t> b = frames[fi];
                if (m > 0) {
                    if (rnd() % 5 == 0) b.resize(rnd() % (b.size() + 1));
                    else
                        for (int k = 0; k < 2 && b.size() > 4; k++) b[4 + rnd() % (b.size() - 4)] ^= (uint8_t)(1 + rnd() % 255);
                }
                mzd_cursor *cur = mzd_cursor_create();
                mzd_cursor_set_threads(cur, (uint32_t)threads);
                const uint64_t max_out = (rnd() & 1) ? 131072 : (1ull << 30);
                size_t pos = 0;
                for (int guard = 0; guard < 100000; guard++) {
                    const size_t left = b.size() - pos;
                    uint8_t *c = (uint8_t *)malloc(left ? left : 1);
                    if (left) memcpy(c, b.data() + pos, left);
                    uint64_t used = 0;
                    const mzd_batch *chunk = nullptr;
                    int last = 0;
                    const int32_t hist[3] = {1, 4, 8};
                    const int rc = mzd_cursor_next(cur, c, left, max_out, 0, hist, &used, &chunk, &last);
                    if (rc == MZD_OK && used > left) { fprintf(stderr, "cursor consumed beyond its source\n"); return 1; }
                    if (rc == MZD_OK && chunk) {
                        acc += walk(chunk);
                        n_chunks++;
                        if (m == 0 && chunk->in_size != used) { fprintf(stderr, "cursor: a chunk's input is not what it consumed\n"); return 1; }
                    }
                    free(c);
                    pos += used;
                    if (rc != MZD_OK) { n_cur_bad++; break; }
                    if (last) break;
                    if (!chunk && used == 0) {  // wants more than there is: a cut frame
                        if (m == 0) { fprintf(stderr, "cursor starved on an intact frame\n"); return 1; }
                        break;
                    }
                }
                if (m == 0 && pos + 4 != b.size() && pos != b.size()) { fprintf(stderr, "cursor left %zu bytes of an intact frame\n", b.size() - pos); return 1; }
                mzd_cursor_destroy(cur);
            }
        }
    }
    printf("cursor: %llu chunks described, %llu frames ended in a status\n", (unsigned long long)n_chunks, (unsigned long long)n_cur_bad);
    printf("san_planner ok: %llu mutated frames planned, %llu rejected with a status, %llu intact concatenations split (acc %llx)\n",
           (unsigned long long)n_ok, (unsigned long long)n_bad, (unsigned long long)n_split_ok, (unsigned long long)acc);
    return 0;
}
