// mzd_huf.hip -- the Huffman literal stage: k_huf (a lane per stream; huffman.go:221-264, literals.go:290-371) and, for the parity
// tests only (-DMZD_TEST_KERNELS), k_huf_seg.  k_huf_w, the wavefront-per-stream kernel of round 6, is mzd_huf_w.hip.  Split out of
// mzd_kernels.hip in round 6 (one file per stage); included by mzd_api.hip behind it.
#pragma once

namespace mzd {

// ------------------------------------------------------------------------------------------
// k_huf: Huffman literal streams.  One wavefront per workgroup; lane = stream; 16 table slots.
//
// Restates huffman.go:221-264: after the padding marker the stream holds R data bits; each
// symbol is looked up with the next MaxBits unread bits (zero-extended below bit 0) and
// consumes NumberOfBits of them; the stream is valid iff exactly R bits are consumed when the
// expected number of symbols has been produced (:257-261 with literals.go:320,332,349,366).

constexpr int kHufQuads = 16;

// Staging area of the transposed bulk phase (tstage != 0): per lane a 128-byte ring of its stream (+ 8 bytes that repeat the
// first 8, for reads that cross the end), 64 bytes of regenerated symbols, and what the lanes tell each other.
constexpr int kHufTRing = 128, kHufTRow = kHufTRing + 16, kHufTOut = 64;  // (rows stay 16-byte aligned)
struct HufTMeta {
    uint32_t need[64];   // the stream wants its next chunk loaded
    int32_t chunk[64];   // ... this one (64-byte chunks of the stream, counted from its start; -1: the zeros below it)
    uint32_t bulk[64];   // the stream takes part in this iteration (its 64 symbols are to be stored)
    int32_t badj[64];    // stream start's offset in its 64-byte line: chunks are aligned in memory
    uint64_t in_off[64];
    uint64_t out_off[64];
};
constexpr int kHufTStageBytes = 64 * kHufTRow + 64 * kHufTOut + (int)sizeof(HufTMeta);

__global__ __launch_bounds__(64) void k_huf(const uint8_t *__restrict__ in, const HufTask *__restrict__ tasks,
                                            uint32_t n_tasks, const uint16_t *__restrict__ huf_entries,
                                            uint8_t *__restrict__ litbuf, uint8_t *out_blob, BlockSum *sums, uint32_t slot_cells,
                                            uint32_t tstage)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tbl_all = (uint16_t *)smem;
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x * 64 + lane;
    HufTask t;
    if (tid < n_tasks) t = tasks[tid];
    else { t.in_size = 0; t.out_size = 0; t.table_off = 0; t.max_bits = 0; t.in_off = 0; t.out_off = 0; t.block = 0; t.pad = 0; }
    // where the stream's symbols go: the literal scratch, or -- a block without sequences whose place in its frame is known at
    // upload (HufTask.pad) -- the output blob itself
    uint8_t *const obase = t.pad ? out_blob : litbuf;

    // stage the (up to) 16 tables of this wavefront: all 64 lanes copy each table
    for (int q = 0; q < kHufQuads; q++) {
        uint32_t off = (uint32_t)__shfl((int)t.table_off, q * 4, 64);
        uint32_t mb = (uint32_t)__shfl((int)t.max_bits, q * 4, 64);
        uint32_t live = (uint32_t)__shfl((int)(t.in_size | t.out_size), q * 4, 64);
        if (live == 0) continue;
        const uint32_t n32 = (1u << mb) >> 1;  // cells are 2 bytes; tables start on even cells; max_bits >= 1
        const uint32_t *src = (const uint32_t *)(huf_entries + off);
        uint32_t *dst = (uint32_t *)(tbl_all + (size_t)q * slot_cells);
        for (uint32_t i = lane; i < n32; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    // (all lanes still active here) largest MaxBits of the wavefront's tables: decides the bulk loop's refill spacing
    const uint32_t mbw = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(t.max_bits));
    const bool wide = mbw <= 7;
    const bool tmode = tstage != 0 && mbw <= 5;  // wave-uniform
    const bool nulltask = (t.in_size | t.out_size) == 0;
    if (nulltask && !tmode) return;  // (with the transposed phase every lane stays: it loads and stores for other lanes' streams)

    const uint16_t *tbl = tbl_all + (size_t)(lane >> 2) * slot_cells;
    const int mb = (int)t.max_bits;
    BackBits br;
    int rem = nulltask ? 0 : br.init(in + t.in_off, (int)t.in_size);
    int status = MZD_OK;
    if (rem < 0) status = MZD_ERR_BAD_PADDING;
    uint8_t *out = obase + t.out_off;
    uint32_t cnt = 0;
    const uint32_t want = t.out_size;

    if (tmode) {
        // ---- transposed bulk phase.  A lane per stream makes every load and store of the wavefront a 64-line scatter, and
        // the CU's address unit is what k_huf fills (TA_BUSY = its duration; beside the sequence stage it cost that stage
        // 2 ms of the pass).  Here global memory is touched only in 64-byte runs: FOUR lanes load a stream's next 64-byte
        // chunk into the stream's LDS ring (16 streams per instruction) and four lanes store a stream's 64 regenerated
        // bytes; the owner lane decodes from its ring (11 / 11 / 10 symbols between two 8-byte ring reads) into LDS.
        // An iteration regenerates 64 symbols for every stream that still has 64 symbols and 320 bits to go; what is left
        // of a stream takes the loops below.  Invariant at the start of an iteration: the ring holds the stream's bytes
        // [64 clow, 64 clow + 128) and ptr - 40 >= 64 clow (an iteration consumes at most 40 bytes).
        // The workgroup is ONE wavefront and a wavefront's LDS operations execute in order: what one lane wrote is there
        // for the lane that reads it in a later instruction.  Only the compiler has to keep the order -- a __syncthreads()
        // would also wait for the global stores of the iteration (7 900 of an iteration's 17 300 cycles).
        auto lds_order = []() { asm volatile("" ::: "memory"); };
        uint8_t *ringb = smem + tstage;
        uint8_t *ostb = ringb + 64 * kHufTRow;
        HufTMeta *mt = (HufTMeta *)(ostb + 64 * kHufTOut);
        uint8_t *myring = ringb + lane * kHufTRow;
        const int len = (int)t.in_size;
        bool inb = !nulltask && status == MZD_OK && 64u <= want && rem >= 64 * 5;
        uint64_t C = br.C;
        int k = br.k, ptr = br.ptr;
        // chunks are 64-byte aligned in MEMORY (every load is one aligned 16-byte piece of one line): positions in the ring
        // and chunk numbers are those of x + badj, x the stream-relative byte offset
        const int badj = (int)((uintptr_t)(in + t.in_off) & 63);
        int clow = ((len - 1 + badj) >> 6) + 1;  // nothing in the ring yet: the two fills below bring chunks ct and ct - 1
        mt->in_off[lane] = t.in_off - (uint64_t)badj;  // (of shifted position 0)
        mt->out_off[lane] = (uint64_t)(uintptr_t)(obase + t.out_off);  // (the address itself: streams of one wavefront may go to either place)
        mt->badj[lane] = badj;
        auto fill = [&](bool need) {  // the streams with `need` get chunk clow - 1 (cooperatively), clow moves down
            mt->need[lane] = need ? 1u : 0u;
            mt->chunk[lane] = clow - 1;
            lds_order();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int sidx = 16 * i + (lane >> 2), piece = lane & 3;
                if (mt->need[sidx]) {
                    const int xs = 64 * mt->chunk[sidx] + 16 * piece;  // shifted position of these 16 bytes
                    const int x0 = xs - mt->badj[sidx];                // stream-relative
                    U128U q{0, 0, 0, 0};
                    if (x0 > -16) {
                        const uint4 qa = *(const uint4 *)(in + mt->in_off[sidx] + xs);  // (the blob has MZD_IN_PAD readable bytes in front)
                        q = U128U{qa.x, qa.y, qa.z, qa.w};
                        if (x0 < 0) {  // bytes below the start of the stream read as zero (reversebitstream.go:23-27)
                            const int z = -x0;  // 1..15 bytes
                            uint64_t lo = (uint64_t)q.x | ((uint64_t)q.y << 32), hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
                            if (z >= 8) { lo = 0; hi = (hi >> (8 * (z - 8))) << (8 * (z - 8)); }
                            else lo = (lo >> (8 * z)) << (8 * z);
                            q = U128U{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                        }
                    }
                    uint8_t *r = ringb + sidx * kHufTRow;
                    const int ro = xs & (kHufTRing - 1);
                    *(uint4 *)(r + ro) = uint4{q.x, q.y, q.z, q.w};
                    if (ro == 0) *(uint2 *)(r + kHufTRing) = uint2{q.x, q.y};
                }
            }
            if (need) clow -= 1;
            lds_order();
        };
#ifdef MZD_HUF_RING_UNALIGNED
        auto ring64 = [&](int x) -> uint64_t { return ((const U64U *)(myring + ((x + badj) & (kHufTRing - 1))))->v; };
#else
        // (the 8 bytes at the cursor as TWO aligned 8-byte reads and a funnel shift: a byte-misaligned 8-byte LDS read holds the pipe a
        // cycle per active lane -- 64 cycles for this wavefront, six times per 64 symbols; the ring's spare bytes serve the second read
        // of a cursor in the last 8)
        auto ring64 = [&](int x) -> uint64_t {
            const uint32_t a = (uint32_t)(x + badj) & (uint32_t)(kHufTRing - 1);
            const uint64_t *p8 = (const uint64_t *)(myring + (a & ~7u));
            const uint64_t lo = p8[0], hi = p8[1];
            const uint32_t sh = 8u * (a & 7u);
            return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
        };
#endif
        if (__any(inb)) {
            fill(inb);
            fill(inb);
            uint32_t it = 0;
            // From here on a stream's next chunk is REQUESTED at the start of an iteration (when the cursor is within 88
            // bytes of the ring's low end), travels while the 64 symbols are decoded, and goes into the ring at the END of
            // the iteration -- if the chunk it replaces is dead by then (cursor + 8 <= 64 clow + 64; else it is dropped and
            // requested again: the cursor was still more than 48 bytes above the low end).  The global latency hides behind
            // the decode.
            do {
                mt->bulk[lane] = inb ? 1u : 0u;
                mt->need[lane] = (inb && ptr + badj - 88 < 64 * clow) ? 1u : 0u;
                mt->chunk[lane] = clow - 1;
                lds_order();
                U128U q[4];
                int qx[4], qz[4];
                bool qv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int sidx = 16 * i + (lane >> 2), piece = lane & 3;
                    qv[i] = mt->need[sidx] != 0;
                    qx[i] = 64 * mt->chunk[sidx] + 16 * piece;  // shifted position
                    // (always a load, from a harmless address when there is nothing to fetch: a conditional one would make the
                    // compiler wait for it right here; the blob has MZD_IN_PAD readable bytes in front of the first stream)
                    qz[i] = qx[i] - mt->badj[sidx];  // stream-relative: < 0 is below the start of the stream
                    const uint4 qa = *(const uint4 *)(qv[i] && qz[i] > -16 ? in + mt->in_off[sidx] + qx[i]
                                                                          : (const uint8_t *)((uintptr_t)in & ~(uintptr_t)15));
                    q[i] = U128U{qa.x, qa.y, qa.z, qa.w};
                }
                if (inb) {
                    uint32_t w[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) w[j] = 0;
#pragma unroll
                    for (int j = 0; j < 64; j++) {
                        if (j == 0 || j == 11 || j == 22 || j == 32 || j == 43 || j == 54) {
                            ptr -= k >> 3;
                            k &= 7;
                            C = ring64(ptr);
                        }
                        const uint32_t idx = (uint32_t)((C << k) >> (64 - mb));
                        const uint32_t e = tbl[idx];
                        w[j >> 2] |= (e & 0xFF) << (8 * (j & 3));
                        const int nb = (int)(e >> 8);
                        k += nb;
                        rem -= nb;
                    }
                    // (the owner lane storing its 64 symbols itself -- four scattered 16-byte stores, no staging: 1.43 vs 1.31 ms)
                    uint4 *o = (uint4 *)(ostb + lane * kHufTOut);
                    o[0] = uint4{w[0], w[1], w[2], w[3]};
                    o[1] = uint4{w[4], w[5], w[6], w[7]};
                    o[2] = uint4{w[8], w[9], w[10], w[11]};
                    o[3] = uint4{w[12], w[13], w[14], w[15]};
                    cnt += 64;
                }
                // does the requested chunk go in?  (the cursor after this iteration's last ring read: ptr; k < 64)
                const bool commit = mt->need[lane] != 0 && (ptr + badj - (k >> 3)) + 8 <= 64 * clow + 64;
                lds_order();  // everybody's ring reads and need / chunk reads are done; the staged symbols are in LDS
                mt->need[lane] = commit ? 1u : 0u;
                if (commit) clow -= 1;
                lds_order();
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int sidx = 16 * i + (lane >> 2), piece = lane & 3;
                    if (qv[i] && mt->need[sidx]) {
                        U128U qq = q[i];
                        const int x0 = qz[i];
                        if (x0 < 0) {  // bytes below the start of the stream read as zero (reversebitstream.go:23-27)
                            const int z = min(-x0, 16);
                            uint64_t lo = (uint64_t)qq.x | ((uint64_t)qq.y << 32), hi = (uint64_t)qq.z | ((uint64_t)qq.w << 32);
                            if (z >= 16) { lo = 0; hi = 0; }
                            else if (z >= 8) { lo = 0; hi = (hi >> (8 * (z - 8))) << (8 * (z - 8)); }
                            else lo = (lo >> (8 * z)) << (8 * z);
                            qq = U128U{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                        }
                        uint8_t *r = ringb + sidx * kHufTRow;
                        const int ro = qx[i] & (kHufTRing - 1);
                        *(uint4 *)(r + ro) = uint4{qq.x, qq.y, qq.z, qq.w};
                        if (ro == 0) *(uint2 *)(r + kHufTRing) = uint2{qq.x, qq.y};
                    }
                    if (mt->bulk[sidx]) {
                        const uint4 v = *(const uint4 *)(ostb + sidx * kHufTOut + 16 * piece);
                        *(U128U *)((uint8_t *)(uintptr_t)mt->out_off[sidx] + 64ull * it + 16 * piece) = U128U{v.x, v.y, v.z, v.w};
                    }
                }
                it++;
                inb = inb && cnt + 64 <= want && rem >= 64 * 5;
                lds_order();  // ring and staging are free for the next iteration
            } while (__any(inb));
            // back to the reader of the loops below: the 8 bytes at the cursor, whole consumed bytes dropped, lookahead
            if (!nulltask && status == MZD_OK) {
                ptr -= k >> 3;
                k &= 7;
                br.ptr = ptr;
                br.k = k;
                br.C = br.load_below(ptr);
                br.D = br.load_below(ptr - 8);
            }
        }
        if (nulltask) return;
    }

    if (status == MZD_OK) {
        // bulk: 16 symbols per iteration while at least 16*11 bits and 16 output slots remain.  A refill is a
        // per-lane gather (64 distinct lines per load) and k_huf shares the CU's address path with k_seq_pipe
        // (the faster k_huf is out of the way, the shorter the pass), so:
        //  - when every table of the wavefront has MaxBits <= 7, EIGHT symbols fit between two refills (k < 8 after
        //    a refill, 7 + 8 * 7 <= 64): half the gathers (same-box A/B of the pass: 26.68 -> 26.05 ms);
        //  - else four symbols per refill (7 + 4 * 11 + window), but a load brings SIXTEEN bytes and serves TWO
        //    refills: the first takes its top bytes, the second the bytes `s` below the top (s <= 7 = what the first
        //    consumed) and issues the next load.  Bytes below the stream's start may be in those 16; only indices
        //    >= 2 of them are ever taken.  (Config 3, MaxBits 11: 3.43 -> 3.18 ms; with eight symbols per refill
        //    the extra shifts cost more than the gathers saved: 25.5 -> 25.7 ms.)
        // A 16-byte load that serves TWO refills (the first takes its top bytes, the second the bytes `s` below the top --
        // s <= 7 = what the first consumed -- and issues the next load).  Bytes below the stream's start may be in those
        // 16; only indices >= 2 of them are ever taken.
        const uint8_t *sb = br.s;
        uint64_t Qhi = 0, Qlo = 0;
        uint32_t s8 = 0;  // 8 * (bytes of Q already taken)
        auto q_begin = [&]() {  // Q = the 16 bytes below the window; its upper half is the 8-byte lookahead the reader already holds
            Qhi = br.D;
            Qlo = ld64u(sb + max(br.ptr - 16, -16));
        };
        auto refill_first = [&]() {  // takes the top bytes of a fresh Q
            const int nb = br.k >> 3, sh = nb * 8;
            br.C = (br.C << sh) | ((Qhi >> 1) >> (63 - sh));
            br.ptr -= nb;
            br.k &= 7;
            s8 = (uint32_t)sh;
        };
        auto refill_second = [&]() {  // takes the bytes s below the top of Q, then requests the next Q
            const int nb = br.k >> 3, sh = nb * 8;
            const uint64_t M = (Qhi << s8) | ((Qlo >> 1) >> (63 - s8));
            br.C = (br.C << sh) | ((M >> 1) >> (63 - sh));
            br.ptr -= nb;
            br.k &= 7;
            const U128U q = *(const U128U *)(sb + max(br.ptr - 16, -16));  // ONE 16-byte gather
            Qlo = (uint64_t)q.x | ((uint64_t)q.y << 32);
            Qhi = (uint64_t)q.z | ((uint64_t)q.w << 32);
        };
        if (mbw <= 5) {
            // ELEVEN symbols fit between two refills (7 + 11 * 5 <= 64).  k_huf is bound by the CU's address unit (TA_BUSY =
            // the kernel's duration: every load and store of a wavefront is a 64-line scatter), so what counts is memory
            // INSTRUCTIONS per symbol: 64 symbols per iteration in groups of 11, 11, 10, 11, 11, 10 -- six refills fed by
            // three 16-byte loads -- and four 16-byte stores: 7 per 64 symbols (three 8-byte refill loads and two stores per
            // 32 symbols before: 10 per 64).
            if (cnt + 64 <= want && rem >= 64 * 5) {
                q_begin();
                do {
                    uint32_t w[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) w[j] = 0;
#pragma unroll
                    for (int j = 0; j < 64; j++) {
                        if (j == 0 || j == 22 || j == 43) refill_first();
                        if (j == 11 || j == 32 || j == 54) refill_second();
                        uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
                        uint32_t e = tbl[idx];
                        w[j >> 2] |= (e & 0xFF) << (8 * (j & 3));
                        int nb = (int)(e >> 8);
                        br.k += nb;
                        rem -= nb;
                        if ((j & 15) == 15) *(U128U *)(out + cnt + (j & ~15)) = U128U{w[(j >> 2) - 3], w[(j >> 2) - 2], w[(j >> 2) - 1], w[j >> 2]};
                    }
                    cnt += 64;
                } while (cnt + 64 <= want && rem >= 64 * 5);
                br.D = br.load_below(br.ptr - 8);  // back to the 8-byte lookahead of the loops below
            }
            // what is left of the stream above 32 symbols: three 8-byte refills per 32 symbols
            while (cnt + 32 <= want && rem >= 32 * 5) {
                uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    if (j == 0 || j == 11 || j == 22) br.refill();
                    uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
                    uint32_t e = tbl[idx];
                    w[j >> 2] |= (e & 0xFF) << (8 * (j & 3));
                    int nb = (int)(e >> 8);
                    br.k += nb;
                    rem -= nb;
                }
                *(U128U *)(out + cnt) = U128U{w[0], w[1], w[2], w[3]};
                *(U128U *)(out + cnt + 16) = U128U{w[4], w[5], w[6], w[7]};
                cnt += 32;
            }
        }
        if (wide) {
            while (cnt + 16 <= want && rem >= 16 * 11) {
                uint32_t w[4];
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    br.refill();
                    uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
                        uint32_t e = tbl[idx];
                        if (j < 4) acc0 |= (e & 0xFF) << (8 * j);
                        else acc1 |= (e & 0xFF) << (8 * (j - 4));
                        int nb = (int)(e >> 8);
                        br.k += nb;
                        rem -= nb;
                    }
                    w[g] = acc0;
                    w[g + 1] = acc1;
                }
                U128U v{w[0], w[1], w[2], w[3]};
                *(U128U *)(out + cnt) = v;
                cnt += 16;
            }
        } else if (cnt + 16 <= want && rem >= 16 * 11) {
            q_begin();
            do {
                uint32_t w[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if ((g & 1) == 0) refill_first();
                    else refill_second();
                    uint32_t acc = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
                        uint32_t e = tbl[idx];
                        acc |= (e & 0xFF) << (8 * j);
                        int nb = (int)(e >> 8);
                        br.k += nb;
                        rem -= nb;
                    }
                    w[g] = acc;
                }
                U128U v{w[0], w[1], w[2], w[3]};
                *(U128U *)(out + cnt) = v;
                cnt += 16;
            } while (cnt + 16 <= want && rem >= 16 * 11);
            br.D = br.load_below(br.ptr - 8);  // back to the 8-byte lookahead of the symbol-by-symbol tail
        }
        // tail: symbol by symbol
        while (cnt < want && rem > 0) {
            if (br.k + mb > 56) br.refill();  // only when the window runs low: every refill is a gather
            uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
            uint32_t e = tbl[idx];
            out[cnt++] = (uint8_t)(e & 0xFF);
            int nb = (int)(e >> 8);
            br.k += nb;
            rem -= nb;
        }
        // over-read: huffman.go:257-261.  Bits left over once the stream's share of the literals is full: the
        // reference decodes on until the bits run out (huffman.go:248-255), i.e. past the length
        // literals.go:320,332,349,366 expects -- the same sentinel as a stream that comes up short
        if (rem < 0) status = MZD_ERR_HUF_BITS;
        else if (rem > 0 || cnt != want) status = MZD_ERR_HUF_LENGTH;
    }
    // the reference decodes the streams of a section one after the other and stops at the first error
    // (literals.go:299-361), and the literals before the sequences: lowest stream index wins, and
    // k_exec lets a literals error win over the sequence stage's status
    if (status != MZD_OK) atomicMin(&sums[t.block].huf_err, ((tid & 3u) << 8) | (uint32_t)status);
}

#ifdef MZD_TEST_KERNELS  /* round 6: k_huf_w (mzd_huf_w.hip) took this kernel's place; kept for the parity tests (libmzd_test.so) */
// ------------------------------------------------------------------------------------------
// k_huf_seg: Huffman literal streams with INTRA-STREAM parallelism (huffman.go:221-264, same results
// and same end conditions as k_huf).  A stream is one serial chain of table lookups, so a batch of few
// long streams (BASELINE configs[2]: 16 384 streams of 32 768 symbols) leaves a lane-per-stream kernel
// with one wavefront per CU and ~190 cycles per symbol.  But Huffman codes SELF-SYNCHRONISE: a decoder
// started at a wrong bit position falls into step with the true sequence of code boundaries after a
// few symbols.  So: one WAVEFRONT per stream, the stream's R data bits cut into up to 64 segments of B
// bits, one lane each;
//   count pass   lane j starts kSegApproach bits BEFORE its segment (lane 0: at the exact start),
//                notes the first code boundary t_j at or after the segment's start, counts the symbols
//                that start in [t_j, end of segment) and notes where it leaves, e_j;
//   validation   the chain must close: e_j == t_{j+1} for every j.  Lane 0 is exact, so by induction
//                every lane then counted exactly its share of the true symbol sequence.  A lane whose
//                start disagrees takes its neighbour's exit and recounts; repeated until the chain
//                closes (each round fixes at least the first wrong lane: a code that never
//                synchronises degrades to the serial time, never to a wrong result);
//   write pass   an exclusive scan of the counts gives every lane its output offset.  Codes of seven bits and
//                more (round 3): the count pass has KEPT its symbols, four to a dword, in the top of the lane's
//                strip -- the bits up there are dead, the window only moves down -- and the write pass copies
//                them out, 16 bytes per store (a lane whose symbols caught up with its window, or that has to
//                recount, fills its strip again and decodes again).  Shorter codes make more symbols than
//                the bits they free have room for: those streams count only, and every lane decodes its
//                c_j symbols again from t_j (12 per store).
// Per round and wavefront (config 3; cycles, -DMZD_HUF_SEG_STATS): strip fill 13 k, approach + count 25 k, write-out
// 25 k -- the lookups are the smaller part: a lane's loads and stores are 48 to 72 bytes apart from its neighbours',
// every memory instruction is 64 separate requests to the address unit.  Approach run 128 bits and segment 384 bits
// (48 bytes: the lanes' 16-byte loads stay aligned to each other) measured best: 0.52 ms against 0.72 ms for two
// walks with 256 / 512 bits.
// The status is the one the serial loop gives (huffman.go:248-261, literals.go:320,332,349,366):
// all R bits decode to N symbols and leave rem = R - e_last <= 0 bits; N < want: rem < 0 ? "bits" :
// "length"; N == want: rem < 0 ? "bits" : ok; N > want: the serial loop stops at want with bits left:
// "length".  One workgroup = the (up to) four streams of a literals section = four wavefronts sharing the
// section's decode table in LDS (<= 4 KiB) + a 140-byte strip per lane: four workgroups per CU.

#ifdef MZD_HUF_SEG_STATS
// 0 rounds, 1 validation rounds, 2 lanes recounted, 3 active lanes, 5 lanes whose symbols did not fit; wavefront cycles: 8 strip
// fill, 9 approach + count, 10 validation, 11 scan, 12 write pass, 13 whole stream
__device__ unsigned long long g_huf_seg_stats[16];
#define SEG_CLK() __builtin_readcyclecounter()
#define SEG_ADD(i, v) do { if (lane == 0) atomicAdd(&g_huf_seg_stats[i], (unsigned long long)(v)); } while (0)
#else
#define SEG_CLK() 0ull
#define SEG_ADD(i, v) do { } while (0)
#endif
#ifndef MZD_SEG_APPROACH
#define MZD_SEG_APPROACH 128
#endif
#ifndef MZD_SEG_BITS
#define MZD_SEG_BITS 384
#endif
constexpr int kSegApproach = MZD_SEG_APPROACH;  // bits a lane decodes ahead of its segment to fall into step
constexpr int kSegBits = MZD_SEG_BITS;          // a lane's segment; a round of 64 lanes covers 64 times as much

// Bit window of one lane of k_huf_seg.  A lane's share of a round -- approach run, segment and lookahead,
// kSegLaneBytes of the stream -- is copied ONCE into the lane's own LDS strip (eight 16-byte loads per lane: the
// only reads of the stream; bytes below the start of the stream become zeros there, reversebitstream.go:23-27)
// and all passes read their bits from it with aligned dword reads.  (Refilling from global memory with per-lane
// loads cost the kernel its time: every such load or store is a 64-line gather that keeps the CU's address unit
// busy for ~80 cycles, and there were ~60 of them per lane and round: TA_BUSY = the kernel's duration.)
// The window is 64 bits wide and refilled in whole dwords: C = strip bytes [p, p + 8), p a multiple of 4,
// k = bits already consumed from its top; a refill shifts in the one or two dwords below once k >= 32.
// strip byte that holds the first bit the lane looks at in a round: approach run, segment and lookahead (a code of MaxBits,
// the window's alignment, a refill) lie below it; above it, the dead bits the count pass's symbols overwrite
constexpr int kSegTopByte = 12 + (kSegApproach + kSegBits + 11 + 7 + 32 + 7) / 8;
constexpr int kSegLaneBytes = (kSegTopByte + 1 + 15) / 16 * 16;  // stream bytes per strip, in 16-byte loads
#ifndef MZD_SEG_DWORDS
#define MZD_SEG_DWORDS 35
#endif
constexpr int kSegLaneDwords = MZD_SEG_DWORDS;  // strip stride (odd: the 64 strips start in different LDS banks); what lies above
                                                // the 32 dwords of stream bytes is room for the count pass's symbols
static_assert(kSegLaneDwords > kSegLaneBytes / 4 && (kSegLaneDwords & 1), "strip stride");
static_assert(8 * (kSegTopByte - 8 - 4) >= kSegApproach + kSegBits + 11 + 7 + 32, "a lane's strip covers its approach run, segment and lookahead");

template <int G>  // symbols between two refills: 31 + G * MaxBits <= 64
struct SegDec {
    const uint16_t *tbl;
    uint32_t *strip;  // the lane's LDS strip
    uint64_t C;
    int p, k, mb;     // p: strip byte offset of the window's low end (multiple of 4); k: bits consumed from its top
    int xb, len;      // strip byte r <-> stream byte xb + r

    // copies stream bytes [xb, xb + kSegLaneBytes) into the strip; a_top = absolute bit (from the top of the last
    // byte of the stream) the lane starts at.  The blob has MZD_IN_PAD readable bytes on both sides.
    __device__ __forceinline__ void fill(const uint8_t *s, int stream_len, int a_top)
    {
        U128U q[kSegLaneBytes / 16];
        fill_load(s, stream_len, a_top, q);
        fill_store(stream_len, a_top, q);
    }
    // the two halves of fill() (issuing the loads of the next round's strip before the stores of this one -- loads and stores
    // complete through one counter -- cut a wavefront's round from 69 k to 53 k cycles and the kernel's time not at all: with
    // sixteen wavefronts per CU nobody waits for a single wavefront's latency)
    static __device__ __forceinline__ void fill_load(const uint8_t *s, int stream_len, int a_top, U128U *q)
    {
        const int xb0 = (stream_len - 1 - (a_top >> 3)) - kSegTopByte;
#pragma unroll
        for (int c = 0; c < kSegLaneBytes / 16; c++) {
            const int x = min(max(xb0 + 16 * c, -16), stream_len);  // chunks entirely outside the stream: any readable address
            q[c] = *(const U128U *)(s + x);
        }
    }
    __device__ __forceinline__ void fill_store(int stream_len, int a_top, const U128U *q)
    {
        len = stream_len;
        xb = (len - 1 - (a_top >> 3)) - kSegTopByte;
#pragma unroll
        for (int c = 0; c < kSegLaneBytes / 16; c++) {
            const int x = xb + 16 * c;
            uint64_t lo = (uint64_t)q[c].x | ((uint64_t)q[c].y << 32), hi = (uint64_t)q[c].z | ((uint64_t)q[c].w << 32);
            if (x < 0) {  // bytes below the start of the stream read as zero
                const int z = min(-x, 16);
                if (z >= 8) { lo = 0; hi = z >= 16 ? 0ull : ((hi >> (8 * (z - 8))) << (8 * (z - 8))); }
                else lo = (lo >> (8 * z)) << (8 * z);
            }
            strip[4 * c + 0] = (uint32_t)lo;
            strip[4 * c + 1] = (uint32_t)(lo >> 32);
            strip[4 * c + 2] = (uint32_t)hi;
            strip[4 * c + 3] = (uint32_t)(hi >> 32);
        }
    }
    __device__ __forceinline__ void seek(int a)  // a = absolute bit
    {
        const int r = (len - 1 - (a >> 3)) - xb;  // strip byte that holds the bit
        p = (r & ~3) - 4;
        k = 8 * (p + 7 - r) + (a & 7);
        C = (uint64_t)strip[p >> 2] | ((uint64_t)strip[(p >> 2) + 1] << 32);
    }
    __device__ __forceinline__ void refill()  // k < 32 afterwards
    {
        const uint32_t d1 = strip[(p >> 2) - 1], d2 = strip[(p >> 2) - 2];
        const int n = k >> 5;  // 0, 1 or 2 dwords
        const uint64_t c1 = (C << 32) | d1, c2 = ((uint64_t)d1 << 32) | d2;
        C = n == 0 ? C : (n == 1 ? c1 : c2);
        p -= 4 * n;
        k &= 31;
    }
    __device__ __forceinline__ uint32_t sym()  // one lookup; returns the cell {symbol, nbits << 8}, advances the window
    {
        const uint32_t idx = (uint32_t)((C << k) >> (64 - mb));
        const uint32_t e = tbl[idx];
        k += (int)(e >> 8);
        return e;
    }
    __device__ __forceinline__ uint32_t one()
    {
        if (k >= 32) refill();
        return sym();
    }
    // count_until that also KEEPS the symbols: they go, four to a dword, into the part of the lane's own strip that the
    // window has left behind (dwords kSegLaneDwords - 1 downwards; the bits up there are dead: the window only moves down).
    // The strip's bits are gone afterwards -- whoever needs them again (a lane that recounts, a lane whose symbols did not
    // fit) fills the strip again.  `ovf`: the symbols caught up with the window (short codes: more than four symbols per
    // 32 bits for long enough); nothing is stored from then on and the lane decodes again in the write pass.
    __device__ __forceinline__ uint32_t decode_until(int &pos, int hi, bool &ovf)
    {
        uint32_t n = 0;
        int wd = kSegLaneDwords - 1;  // next dword to take symbols (the stride's spare dword first)
        while (pos + 4 * mb <= hi) {  // all four symbols start below hi
            refill();
            int k0 = k;
            const uint32_t e0 = sym(), e1 = sym();
            if (G < 4) {  // MaxBits 9..11: two symbols per refill
                pos += k - k0;
                refill();
                k0 = k;
            }
            const uint32_t e2 = sym(), e3 = sym();
            pos += k - k0;
            const uint32_t w = (e0 & 0xFF) | ((e1 & 0xFF) << 8) | ((e2 & 0xFF) << 16) | (e3 << 24);
            if (4 * wd >= p + 8) strip[wd] = w;
            else ovf = true;
            wd--;
            n += 4;
        }
        uint32_t w = 0, i = 0;
        while (pos < hi) {
            const uint32_t e = one();
            pos += (int)(e >> 8);
            w |= (e & 0xFF) << (8 * i);
            n++;
            if (++i == 4) {
                if (4 * wd >= p + 8) strip[wd] = w;
                else ovf = true;
                wd--;
                w = 0;
                i = 0;
            }
        }
        if (i) {
            if (4 * wd >= p + 8) strip[wd] = w;
            else ovf = true;
        }
        return n;
    }
    // decodes up to the first code boundary >= hi; returns the number of symbols that START in [pos, hi)
    __device__ __forceinline__ uint32_t count_until(int &pos, int hi)
    {
        uint32_t n = 0;
        while (pos + G * mb <= hi) {  // all G symbols start below hi
            refill();
            const int k0 = k;
#pragma unroll
            for (int g = 0; g < G; g++) sym();
            pos += k - k0;
            n += G;
        }
        while (pos < hi) {
            pos += (int)(one() >> 8);
            n++;
        }
        return n;
    }
};

struct __attribute__((packed, aligned(1))) U96U { uint32_t x, y, z; };
struct __attribute__((packed, aligned(1))) U16U { uint16_t v; };

template <int G>
__device__ __forceinline__ void huf_seg_stream(const uint8_t *__restrict__ in, const HufTask &t, const uint16_t *tbl,
                                               uint32_t *strip, uint8_t *obase, BlockSum *sums,
                                               uint32_t stream_idx, int lane)
{
    const uint8_t *s = in + t.in_off;
    const int len = (int)t.in_size, mb = (int)t.max_bits;
    const uint32_t want = t.out_size;
    // padding: zero bits above the marker and the marker itself (huffman.go:227-238)
    const uint32_t last = len > 0 ? s[len - 1] : 0u;
    int status = last == 0 ? MZD_ERR_BAD_PADDING : MZD_OK;
    const int a0 = last ? (int)__builtin_clz(last) - 24 + 1 : 8;
    const int R = 8 * len - a0;  // data bits
    // Codes of seven bits and more: the count pass KEEPS its symbols (in the dead top of the lane's strip) and the write pass
    // copies them out.  Shorter codes make more symbols than the bits they free have room for: those streams count only, and
    // every lane decodes its share again (the strip is intact then).
#ifdef MZD_SEG_TWO_WALKS  /* A/B: the kernel of round 2 */
    const bool keep = false;
#else
    const bool keep = mb >= 7;
#endif
    SegDec<G> d;
    d.tbl = tbl;
    d.strip = strip;
    d.mb = mb;
    // ROUNDS of 64 segments of kSegBits: a round reads one contiguous 4 KiB piece of the stream and writes one
    // contiguous piece of the literals.
    int p0 = 0;             // exact code boundary where the round starts
    uint32_t out_done = 0;  // symbols written by earlier rounds
    const unsigned long long c_begin = SEG_CLK();
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};  // (summed per stream: an atomic per round and phase throttles the kernel it measures)
    unsigned long long acc_rounds = 0, acc_lanes = 0;
    (void)c_begin;
    (void)acc;
    (void)acc_rounds;
    (void)acc_lanes;
    while (status == MZD_OK && p0 < R) {
        const unsigned long long c0 = SEG_CLK();
        unsigned long long c1 = c0;
        (void)c1;
        const int lo = p0 + lane * kSegBits;
        const int fill_pos = max(lo - kSegApproach, p0);  // where the lane's strip starts (p0 moves on before the write pass)
        const bool act = lo < R;
        const int hi = min(R, lo + kSegBits);
        int tpos = 0, epos = 0;
        uint32_t cnt = 0;
        bool ovf = false;  // the lane's symbols did not fit into its strip: it decodes again in the write pass
        if (act) {
            // ---- the lane's strip, then the count pass: approach, first boundary at or after lo, symbols up to hi
            int pos = fill_pos;
            d.fill(s, len, a0 + pos);
            d.seek(a0 + pos);
            c1 = SEG_CLK();
            while (pos + G * mb <= lo) {
                d.refill();
                const int k0 = d.k;
#pragma unroll
                for (int g = 0; g < G; g++) d.sym();
                pos += d.k - k0;
            }
            while (pos < lo) pos += (int)(d.one() >> 8);
            tpos = pos;
            if (keep) {
                ovf = false;
                cnt = d.decode_until(pos, hi, ovf);
            } else {
                ovf = true;
                cnt = d.count_until(pos, hi);
            }
            epos = pos;
        }
        const unsigned long long c2 = SEG_CLK();
        (void)c2;
        // ---- validation: the chain of boundaries must close (lanes run in lockstep here)
        for (int guard = 0; guard < 66; guard++) {
            const int tnext = __shfl_down(tpos, 1, 64);
            const bool nact = (bool)__shfl_down((int)act, 1, 64) && lane < 63;
            const bool bad = act && nact && epos != tnext;
            if (!__any(bad)) break;
#ifdef MZD_HUF_SEG_STATS
            { const unsigned long long bm = __ballot(bad); if (lane == 0) { atomicAdd(&g_huf_seg_stats[1], 1ull); atomicAdd(&g_huf_seg_stats[2], (unsigned long long)__popcll(bm)); } }
#endif
            const bool fix = (bool)__shfl_up((int)bad, 1, 64) && lane > 0;
            const int newt = __shfl_up(epos, 1, 64);
            if (fix) {  // newt < lo + MaxBits: inside the lane's strip (filled again: the symbols have overwritten its top)
                int pos = newt;
                if (keep) d.fill(s, len, a0 + fill_pos);
                d.seek(a0 + pos);
                tpos = pos;
                if (keep) {
                    ovf = false;
                    cnt = pos < hi ? d.decode_until(pos, hi, ovf) : 0u;
                } else {
                    cnt = pos < hi ? d.count_until(pos, hi) : 0u;
                }
                epos = pos;
            }
        }
#ifdef MZD_HUF_SEG_STATS
        acc_rounds += 1;
        acc_lanes += (unsigned long long)__popcll(__ballot(act));
#endif
        const unsigned long long c3 = SEG_CLK();
        (void)c3;
        const uint32_t incl = wave_incl_scan_u32(act ? cnt : 0u, lane);
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        const uint64_t am = __ballot(act);
        p0 = __shfl(epos, 63 - __builtin_clzll(am), 64);  // lane 0 is active: am != 0
        if (out_done + total > want) {  // the serial loop stops at `want` symbols with bits left (literals.go:320,332,349,366)
            status = MZD_ERR_HUF_LENGTH;
            break;
        }
        const unsigned long long c4 = SEG_CLK();
        (void)c4;
#ifdef MZD_HUF_SEG_STATS
        acc[5] += (unsigned long long)__popcll(__ballot(act && ovf));
#endif
        // ---- write pass: exactly cnt symbols to out + (symbols of the rounds and lanes below) -- from the lane's strip, where
        // the count pass left them ...
        if (act && cnt && !ovf) {
            uint8_t *out = obase + t.out_off + out_done + (incl - cnt);
            int rd = kSegLaneDwords - 1;
            uint32_t n = 0;
            for (; n + 16 <= cnt; n += 16, rd -= 4) *(U128U *)(out + n) = U128U{strip[rd], strip[rd - 1], strip[rd - 2], strip[rd - 3]};
            // the last r < 16 symbols: exactly r bytes leave (the next byte belongs to another lane) -- as ONE more 16-byte store
            // that ends at the lane's last byte and writes some of the bytes before it again (up to four exact stores for the tail
            // were a third of the kernel's store instructions; worth 1-2 %)
            uint32_t r = cnt - n;
#ifndef MZD_SEG_EXACT_TAILS
            if (r && cnt >= 16) {
                const uint32_t s0 = cnt - 16, sh = 8 * (s0 & 3);
                const int m = kSegLaneDwords - 1 - (int)(s0 >> 2);
                const uint32_t d0 = strip[m], d1 = strip[m - 1], d2 = strip[m - 2], d3 = strip[m - 3], d4 = strip[m - 4];
                *(U128U *)(out + s0) = U128U{__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                                             __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
                r = 0;
            }
#endif
            uint8_t *o = out + n;
            if (r & 8) {
                *(U64U *)o = U64U{(uint64_t)strip[rd] | ((uint64_t)strip[rd - 1] << 32)};
                o += 8;
                rd -= 2;
            }
            if (r & 4) {
                *(U32U *)o = U32U{strip[rd]};
                o += 4;
                rd -= 1;
            }
            if (r & 3) {
                uint32_t acc = strip[rd];
                if (r & 2) {
                    *(U16U *)o = U16U{(uint16_t)acc};
                    o += 2;
                    acc >>= 16;
                }
                if (r & 1) *o = (uint8_t)acc;
            }
        }
        // ... or decoded again from tpos (short codes: the symbols overtook the window)
        if (act && cnt && ovf) {
            uint8_t *out = obase + t.out_off + out_done + (incl - cnt);
            if (keep) d.fill(s, len, a0 + fill_pos);
            d.seek(a0 + tpos);
            uint32_t n = 0;
            constexpr int PER = 12;  // symbols per store
            while (n + PER <= cnt) {
                uint32_t w[3] = {0, 0, 0};
#pragma unroll
                for (int g = 0; g < PER / G; g++) {
                    d.refill();
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const int i = g * G + j;
                        w[i >> 2] |= (d.sym() & 0xFF) << (8 * (i & 3));
                    }
                }
                *(U96U *)(out + n) = U96U{w[0], w[1], w[2]};
                n += PER;
            }
            // the last r < 12 symbols: exactly r bytes leave (the next byte belongs to another lane)
            const uint32_t r = cnt - n;
            uint64_t acc = 0;
            uint32_t acc2 = 0;
            for (uint32_t i = 0; i < r; i++) {
                const uint64_t sy = d.one() & 0xFF;
                if (i < 8) acc |= sy << (8 * i);
                else acc2 |= (uint32_t)sy << (8 * (i - 8));
            }
            uint8_t *o = out + n;
            if (r & 8) {
                *(U64U *)o = U64U{acc};
                o += 8;
                acc = acc2;
            }
            if (r & 4) {
                *(U32U *)o = U32U{(uint32_t)acc};
                o += 4;
                acc >>= 32;
            }
            if (r & 2) {
                *(U16U *)o = U16U{(uint16_t)acc};
                o += 2;
                acc >>= 16;
            }
            if (r & 1) *o = (uint8_t)acc;
        }
        out_done += total;
        {
            const unsigned long long c5 = SEG_CLK();
            (void)c5;
            acc[0] += c1 - c0;
            acc[1] += c2 - c1;
            acc[2] += c3 - c2;
            acc[3] += c4 - c3;
            acc[4] += c5 - c4;
        }
    }
    for (int i = 0; i < 5; i++) SEG_ADD(8 + i, acc[i]);
    SEG_ADD(5, acc[5]);
    SEG_ADD(0, acc_rounds);
    SEG_ADD(3, acc_lanes);
    SEG_ADD(13, SEG_CLK() - c_begin);
    // ---- status of the whole stream: what the serial loop gives (see the kernel comment)
    if (status == MZD_OK) {
        const int rem = R - p0;
        if (out_done < want) status = rem < 0 ? MZD_ERR_HUF_BITS : MZD_ERR_HUF_LENGTH;
        else if (rem < 0) status = MZD_ERR_HUF_BITS;
    }
    if (status != MZD_OK && lane == 0) atomicMin(&sums[t.block].huf_err, (stream_idx << 8) | (uint32_t)status);
}

constexpr int kHufSegStripBytes = 4 * 64 * kSegLaneDwords * 4;  // four wavefronts of 64 strips

__global__ __launch_bounds__(256) void k_huf_seg(const uint8_t *__restrict__ in, const HufTask *__restrict__ tasks,
                                                 uint32_t n_tasks, const uint16_t *__restrict__ huf_entries,
                                                 uint8_t *__restrict__ litbuf, uint8_t *out_blob, BlockSum *sums, uint32_t table_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tbl = (uint16_t *)smem;                       // the section's decode table (table_bytes, a multiple of 16)
    uint32_t *strips = (uint32_t *)(smem + table_bytes);    // [wavefront][lane][kSegLaneDwords]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tid = blockIdx.x * 4 + wave;  // tasks come in quads that share one table
    HufTask t = tasks[min(tid, n_tasks - 1)];
    if (tid >= n_tasks) { t.in_size = 0; t.out_size = 0; }
    {
        const HufTask t0 = tasks[blockIdx.x * 4];
        const uint32_t n32 = (1u << t0.max_bits) >> 1;  // 2-byte cells, tables start on even cells, MaxBits >= 1
        const uint32_t *src = (const uint32_t *)(huf_entries + t0.table_off);
        uint32_t *dst = (uint32_t *)tbl;
        for (uint32_t i = threadIdx.x; i < n32; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    if ((t.in_size | t.out_size) == 0) return;  // null task (sections with one stream use the first wavefront only)
    uint32_t *strip = strips + (wave * 64 + lane) * kSegLaneDwords;
    uint8_t *const obase = t.pad ? out_blob : litbuf;  // (see k_huf)
    if (t.max_bits <= 5) huf_seg_stream<6>(in, t, tbl, strip, obase, sums, tid & 3u, lane);
    else if (t.max_bits <= 8) huf_seg_stream<4>(in, t, tbl, strip, obase, sums, tid & 3u, lane);
    else huf_seg_stream<3>(in, t, tbl, strip, obase, sums, tid & 3u, lane);
}

#endif  // MZD_TEST_KERNELS (k_huf_seg)

}  // namespace mzd
