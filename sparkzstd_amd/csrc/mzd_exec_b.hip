// mzd_exec_b.hip -- k_exec_b: sequence execution with ONE WAVEFRONT per frame and ONE LANE PER OUTPUT BYTE.
//
// Replaces decompression/sequence_execution.go:14-63 (ExecuteSequences), ringbuffer.go:102-277 (Push / Repeat /
// RepeatBeforeIndex) and the Raw / RLE block arms framedecompressor.go:211-215,229-241, like k_exec.  Same inputs
// (the 8-byte sequence records of the entropy stage, the regenerated literals) and the same statuses; another shape:
//
//   k_exec    a lane per SEQUENCE, 8 KiB of the block in LDS, a per-byte validity bitmap and a dataflow loop over
//             the pending matches.  Every copy is a byte-misaligned LDS access (a cycle per active lane in the LDS
//             pipe), and its 8.5 KiB per frame keep it from sharing a CU with the sequence stage.
//   k_exec_b  the output is produced strictly IN ORDER, 64 bytes per pass, lane j making byte P + j.  The work is
//             organised in STRETCHES (up to 1024 output bytes and 512 literals of one 64-sequence tile); a stretch's
//             setup leaves in LDS
//               * a bitmap with one bit per output byte: "a literal run or a match starts here" (a HEAD);
//               * a table with one 8-byte entry per head, in output order: {displacement D, mask M} such that the
//                 byte at position p of that run comes from LDS address (p + D) & M --
//                   literal        the stretch's literals, copied from the literal buffer into LDS in one go
//                   staged match   16 source bytes per sequence, loaded from the frame's slab into LDS by the setup:
//                                  every match whose source is final in memory.  All of a stretch's loads are in
//                                  flight together; no pass waits for memory
//                   window match   the last 2 KiB of output live in an LDS ring (position & 0x7ff, that is M)
//             and a pass is: read its 64 bits of the bitmap, count the heads at or below every lane (v_mbcnt), read
//             the table entry, read the byte, store it into the window ring -- ~13 vector and 5 LDS instructions,
//             all aligned whole-wavefront ones.  Two rare cases take a longer pass (the setup marks the passes): a
//             window match made by its own pass (offset <= lane; resolved between the lanes by pointer jumping, six
//             rounds at most) and a window match that is neither staged nor in the ring any more (a long far match:
//             the pass reads the slab itself).  The ring leaves for the slab in 512-byte units.
//             No byte-misaligned LDS access, no validity bitmap, no barrier; 4.7 KiB of LDS per frame, 32 frames per CU.
//
// Hazards are ordered by construction: a wavefront's LDS operations execute in order (a pass's reads precede its
// store); the slab is read only below `confirmed` (window units whose stores a wait on memory has seen complete) or
// after such a wait.
#pragma once

namespace mzd {

constexpr uint32_t kXbWin = 2048;      // window ring
constexpr uint32_t kXbLit = 512;       // literals of the current stretch
constexpr uint32_t kXbStage = 1024;    // staged matches of the current tile: 16 source bytes per sequence lane
constexpr uint32_t kXbStretch = 1024;  // output bytes per stretch at most (16 passes; one bit each in the bitmap)
constexpr uint32_t kXbFlush = 512;     // 64 lanes x 8 bytes leave for the slab at a time
constexpr int kXbNear = (int)kXbWin - 64;  // a window match byte less than this far behind its pass start is read from the ring

struct XbLds {
    uint8_t win[kXbWin];              // 0x000
    uint8_t lit[kXbLit];              // 0x800
    uint8_t stage[kXbStage];          // 0xa00
    uint2 table[130];                 // 0xe00: [0] the run that continues from the stretch before, [1 + k] head k of the stretch, [129] spare
    uint32_t bits[kXbStretch / 32];   // heads
    uint32_t special;                 // passes with a window match the plain pass cannot serve (bit = pass of the stretch)
    uint32_t pad[3];
};
static_assert(offsetof(XbLds, lit) == kXbWin && offsetof(XbLds, table) % 8 == 0 && offsetof(XbLds, bits) % 8 == 0 && sizeof(XbLds) % 16 == 0,
              "alignment of the LDS areas");
static_assert(kXbStretch + 64 + kXbFlush <= kXbWin, "a window unit is issued before the ring wraps onto it (a stretch starts with less than a unit pending)");

#ifdef MZD_XB_STATS
// tools/xb_stats.py: 0 tiles, 1 stretches, 2 plain passes, 3 special passes, 4 passes with a byte made by the pass itself,
// 5 passes that went to memory themselves, 6 staged matches, 7 matches, 8 cycles in tile setup, 9 cycles in stretch
// setup, 10 cycles in passes, 11 cycles total, 12 frames
__device__ unsigned long long g_xb_stats[16];
#define XB_STAT(i, n) (xbst[i] += (unsigned long long)(n))
#define XB_CLOCK() __builtin_readcyclecounter()
#else
#define XB_STAT(i, n) do { } while (0)
#define XB_CLOCK() 0ull
#endif

__device__ __forceinline__ void xb_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// bytes [flushed, upto) of the frame leave the window for the slab, byte by byte (block ends, unaligned remainders)
__device__ __forceinline__ void xb_flush_bytes(XbLds &sh, uint8_t *out, uint32_t &flushed, uint32_t upto, int lane)
{
    for (uint32_t x = flushed + (uint32_t)lane; x < upto; x += 64) out[x] = sh.win[x & (kXbWin - 1)];
    flushed = upto;
}

// the 512-byte unit at `flushed` (or the bytes up to the next unit boundary) leaves for the slab
__device__ __forceinline__ void xb_flush_step(XbLds &sh, uint8_t *out, uint32_t &flushed, int lane)
{
    if ((flushed & (kXbFlush - 1)) == 0) {
        const uint32_t x = flushed + 8u * (uint32_t)lane;
        const uint64_t v = *(const uint64_t *)&sh.win[x & (kXbWin - 1)];
        ((U64U *)(out + x))->v = v;
        flushed += kXbFlush;
    } else {
        xb_flush_bytes(sh, out, flushed, (flushed + kXbFlush) & ~(kXbFlush - 1), lane);
    }
}

// after a bulk write straight to the slab (Raw / RLE blocks, literal-only blocks): the window ring takes the last
// bytes of the frame back from memory so that the next block's window matches find them
__device__ __forceinline__ void xb_reload_window(XbLds &sh, const uint8_t *out, uint32_t outPos, uint32_t &flushed, uint32_t &confirmed,
                                                 int lane)
{
    xb_wait_vm();  // the bulk stores are in memory (same CU: visible to the loads below)
    const uint32_t lo = outPos > kXbWin ? outPos - kXbWin : 0u;
    const uint32_t lo4 = (lo + 3u) & ~3u;  // (the ring is trusted kXbNear + 63 bytes back: 1 to 3 bytes less than it holds here)
    const uint32_t hi4 = outPos & ~3u;
    for (uint32_t x = lo4 + 4u * (uint32_t)lane; x < hi4; x += 256) *(uint32_t *)&sh.win[x & (kXbWin - 1)] = ((const U32U *)(out + x))->v;
    for (uint32_t x = max(lo4, hi4) + (uint32_t)lane; x < outPos; x += 64) sh.win[x & (kXbWin - 1)] = out[x];
    flushed = outPos;
    confirmed = outPos;
}

__device__ __forceinline__ void xb_bulk_copy(uint8_t *dst, const uint8_t *src, uint32_t n, int lane)
{
    const uint32_t n16 = n >> 4;
    for (uint32_t i = (uint32_t)lane; i < n16; i += 64) *(U128U *)(dst + 16 * i) = *(const U128U *)(src + 16 * i);
    for (uint32_t i = (n16 << 4) + (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
}
__device__ __forceinline__ void xb_bulk_fill(uint8_t *dst, uint32_t byte, uint32_t n, int lane)
{
    const uint32_t v = byte * 0x01010101u;
    const U128U f{v, v, v, v};
    const uint32_t n16 = n >> 4;
    for (uint32_t i = (uint32_t)lane; i < n16; i += 64) *(U128U *)(dst + 16 * i) = f;
    for (uint32_t i = (n16 << 4) + (uint32_t)lane; i < n; i += 64) dst[i] = (uint8_t)v;
}

// lanes whose source byte is produced by this very pass (offset <= lane): pointer jumping between the lanes -- a lane either
// takes its source lane's byte or, while that one is still waiting itself, its source lane.  At most six rounds.
__device__ __noinline__ uint32_t xb_resolve_in_pass(uint32_t val, int r, uint32_t lane)
{
    uint32_t srcl = r >= 0 ? (uint32_t)r : lane;
    uint32_t done = r >= 0 ? 0u : 1u;
    do {
        const uint32_t v2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)val);
        const uint32_t d2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)done);
        const uint32_t s2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)srcl);
        if (!done) {
            if (d2) {
                val = v2;
                done = 1;
            } else {
                srcl = s2;
            }
        }
    } while (wave_any(!done));
    return val;
}


// index of the table entry that owns every byte of a pass: sbase + (heads of the pass at or below the lane) - 1
__device__ __forceinline__ uint32_t xb_owner(uint64_t H, uint32_t sbase, uint32_t lblo, uint32_t lbhi)
{
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H, sbase));
    const uint32_t own = ((uint32_t)H & lblo) | ((uint32_t)(H >> 32) & lbhi);
    return below - (own ? 0u : 1u);
}

__global__ __launch_bounds__(64, 8) void k_exec_b(const uint8_t *__restrict__ in, uint8_t *out_blob, const DFrame *__restrict__ frames,
                                                  const DBlock *__restrict__ blocks, const BlockSum *__restrict__ sums,
                                                  const uint64_t *__restrict__ recs, const uint8_t *__restrict__ litbuf,
                                                  int32_t *frame_status, uint64_t *frame_out_len,
                                                  const uint32_t *__restrict__ order, uint32_t first)
{
    __shared__ __attribute__((aligned(16))) XbLds sh;
    const uint8_t *const lds = (const uint8_t *)&sh;
    const int lane = threadIdx.x;
    // this wavefront's frame: in the batch's execution order when it has one (heterogeneous batches: the largest first)
    const uint32_t fidx = order ? order[first + blockIdx.x] : first + blockIdx.x;
    const DFrame fr = frames[fidx];
    uint8_t *out = out_blob + fr.out_offset;

    int error = fr.plan_status;
    uint32_t outPos = 0;         // bytes of this frame produced so far (frames of 4 GiB and more take k_exec)
    uint32_t flushed = 0;        // [0, flushed) has left for the slab (the youngest units may still be in flight)
    uint32_t confirmed = 0;      // [0, confirmed) has ARRIVED in the slab: a wait on memory came after its stores
    int H0 = 1, H1 = 4, H2 = 8;  // framedecompressor.go:48,59
    if (lane < (int)(kXbStretch / 32)) sh.bits[lane] = 0u;
    if (lane == 0) sh.special = 0u;
    // constants of the passes, in VGPRs (a vector instruction with a literal or scalar operand issues at half rate)
    uint32_t v7ff = kXbWin - 1;
    uint32_t lblo = lane < 32 ? 1u << lane : 0u, lbhi = lane < 32 ? 0u : 1u << (lane - 32);
    asm volatile("" : "+v"(v7ff), "+v"(lblo), "+v"(lbhi));
#ifdef MZD_XB_STATS
    unsigned long long xbst[16] = {0};
    const unsigned long long xb_t0 = XB_CLOCK();
#endif

    for (uint32_t bi = 0; bi < fr.n_blocks && error == MZD_OK; bi++) {
        const DBlock b = blocks[fr.first_block + bi];
        if (b.type != MZD_BLOCK_COMPRESSED) {
            // Raw (framedecompressor.go:211-215) / RLE (:229-241): straight copy / fill
            if ((uint64_t)outPos + b.size > fr.out_capacity) {
                error = MZD_ERR_DST_FULL;
                break;
            }
            xb_flush_bytes(sh, out, flushed, outPos, lane);
            if (b.type == MZD_BLOCK_RAW) xb_bulk_copy(out + outPos, in + b.src_off, b.size, lane);
            else xb_bulk_fill(out + outPos, in[b.src_off], b.size, lane);
            outPos += b.size;
            xb_reload_window(sh, out, outPos, flushed, confirmed, lane);
            continue;
        }

        const BlockSum bsum = sums[fr.first_block + bi];
        const uint32_t litTotal = bsum.lit_total, seqOut = bsum.out_total;
        int err = bsum.huf_err != 0xFFFFFFFFu ? (int)(bsum.huf_err & 0xFF) : bsum.status;
        if (err == MZD_OK && litTotal > b.lit_regen) err = MZD_ERR_LITERALS;  // sequence_execution.go:27-29
        const uint32_t blockOut = seqOut + (b.lit_regen - min(litTotal, b.lit_regen));
        if (err == MZD_OK && blockOut > kBlockMax) err = MZD_ERR_CORRUPT_SIZES;
        if (err == MZD_OK && (uint64_t)outPos + blockOut > fr.out_capacity) err = MZD_ERR_DST_FULL;
        if (err != MZD_OK) {
            error = err;
            break;
        }
        const uint8_t *lits = (b.lit_type == MZD_LIT_HUF ? litbuf : in) + b.lit_src;
        const bool litRle = b.lit_type == MZD_LIT_RLE;

        if (b.n_seq == 0) {
            // no sequences: the block IS its literals (sequence_execution.go:55-59) -- unless the Huffman stage has
            // already put them in place
            xb_flush_bytes(sh, out, flushed, outPos, lane);
            if (!b.pad[0]) {
                if (litRle) xb_bulk_fill(out + outPos, lits[0], b.lit_regen, lane);
                else xb_bulk_copy(out + outPos, lits, b.lit_regen, lane);
            }
            outPos += b.lit_regen;
            xb_reload_window(sh, out, outPos, flushed, confirmed, lane);
            continue;
        }
        if (litRle) {  // RLE literals (literals.go:390-396): every literal of every stretch is this byte
            const uint64_t v = lits[0] * 0x0101010101010101ull;
            *(uint64_t *)&sh.lit[8 * lane] = v;
        }

        // ---- tiles of 64 sequences; the literals after the last sequence (sequence_execution.go:55-59) ride along as
        // one more sequence without a match
        const uint32_t rest = b.lit_regen - litTotal;
        const uint32_t nps = b.n_seq + (rest ? 1u : 0u);
        const uint32_t ntiles = (nps + 63) >> 6;
        const uint64_t *brec = recs + b.rec_off;
        uint32_t tileStart = outPos;  // frame-relative position of the tile's first byte
        uint32_t litRun = 0;          // literals of the block that earlier tiles consumed
        uint64_t rec_n = (uint32_t)lane < b.n_seq ? brec[lane] : 0ull;
        for (uint32_t t = 0; t < ntiles; t++) {
            const unsigned long long xb_t1 = XB_CLOCK();
            (void)xb_t1;
            const uint64_t rec = rec_n;
            const uint32_t si = t * 64 + (uint32_t)lane;
            {
                const uint32_t sn = si + 64;
                rec_n = sn < b.n_seq ? brec[sn] : 0ull;
            }
            const bool isSeq = si < b.n_seq;
            uint32_t LL = (uint32_t)rec & kRecLlMask;
            const uint32_t ML = (uint32_t)(rec >> kRecMlShift) & kRecMlMask;  // (0 beyond the last sequence: rec == 0)
            const uint32_t offf = (uint32_t)(rec >> kRecOffShift) & kRecOffMask;
            if (si == b.n_seq) LL = rest;
            int off = (int)offf;
            if (offf & kRecOffSymbolic) {
                const uint32_t u = offf & (kRecOffSymbolic - 1);
                off = sel3(u & 3, H0, H1, H2) - (int)(u >> 2);
            }
            const uint32_t sLL = wave_incl_scan_dpp(LL), sOut = wave_incl_scan_dpp(LL + ML);
            const uint32_t tileLits = (uint32_t)__builtin_amdgcn_readlane((int)sLL, 63);
            const uint32_t tileOut = (uint32_t)__builtin_amdgcn_readlane((int)sOut, 63);
            const uint32_t mstart = tileStart + sOut - ML, lstart = mstart - LL;  // frame-relative
            const uint32_t lsrc = litRun + sLL - LL;                             // the sequence's first literal (index in the block)
            const bool bad = isSeq && ML > 0 && (off <= 0 || (uint32_t)off > mstart);  // ringbuffer.go:206-214
            if (wave_any(bad)) {
                error = MZD_ERR_OFFSET;
                break;
            }
            const uint32_t E = tileStart + tileOut;
            // a match byte at position p comes from LDS offset (p + md.x) & md.y: the window ring until a stretch setup stages it
            uint2 md = make_uint2((uint32_t)(-off), kXbWin - 1);
            const uint32_t litD = (uint32_t)offsetof(XbLds, lit) + lsrc - lstart;  // literal byte p: LDS offset p + litD - la
            const bool ringSpecial = off < 64 || off > kXbNear;  // (as long as the match stays a window match)
            uint32_t la = litRun;  // literal cursor at the stretch's start
            uint32_t P = tileStart;
            uint32_t p = P + (uint32_t)lane;
            XB_STAT(0, 1);
            XB_STAT(7, __popcll(wave_ballot(isSeq && ML > 0)));
            XB_STAT(8, XB_CLOCK() - xb_t1);

            while (P < E) {
                // ---- a stretch: setup for up to 1024 output bytes / 512 literals, then its passes
                const unsigned long long xb_t2 = XB_CLOCK();
                (void)xb_t2;
                uint32_t sEnd = min(E, P + kXbStretch);
                // literal cursor at the stretch's end: the last sequence that starts at or before it (the lanes' starts ascend;
                // lane 0 starts at tileStart); a stretch that would need more than 512 literals ends where the 512th does
                uint32_t lb;
                {
                    const int k = 63 - __builtin_clzll(wave_ballot(lstart <= sEnd));
                    const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)lsrc, k);
                    const uint32_t kn = (uint32_t)__builtin_amdgcn_readlane((int)LL, k);
                    lb = ks + min(kn, sEnd - kl);
                }
                if (lb - la > kXbLit) {
                    const int k = 63 - __builtin_clzll(wave_ballot(lsrc <= la + kXbLit));  // the run that holds literal la + 512
                    const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)lsrc, k);
                    sEnd = kl + (la + kXbLit - ks);
                    lb = la + kXbLit;
                }
                const uint32_t sLen = sEnd - P;
                const bool litIn = LL > 0 && lstart - P < sLen;   // the sequence's literal run / match starts in this stretch
                const bool mIn = ML > 0 && mstart - P < sLen;
                const bool contL = lstart < P && P < mstart;      // ... or continues from the stretch before (one lane at most)
                const bool contM = mstart < P && P < mstart + ML;
                // matches whose source is final in the slab: 16 source bytes into the stage, all of the stretch's loads in
                // flight together, so that no pass has to wait for memory itself
                // (a match not farther back than the ring reaches is served by the ring whatever pass it falls into)
                const uint32_t q0 = mstart - (uint32_t)off;
                const bool stg = mIn && ML <= 16 && off > kXbNear && q0 + ML <= confirmed;
                U128U sv{0, 0, 0, 0};
                if (stg) sv = *(const U128U *)(out + q0);
                // the stretch's literals: [la, la + 512) of the block's literals
                uint64_t lv = 0;
                {
                    const uint32_t li = la + 8u * (uint32_t)lane;
                    if (!litRle && li < lb) lv = ((const U64U *)(lits + li))->v;  // (lb <= lit_regen)
                }
                // table index of the sequence's heads: 1 + the heads of the lanes below (+ its own literal head)
                const uint64_t litMask = wave_ballot(litIn), mMask = wave_ballot(mIn);
                const uint32_t hb = __builtin_amdgcn_mbcnt_hi((uint32_t)(litMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)litMask, 1u)) +
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mMask, 0u));
                if (stg) md = make_uint2((uint32_t)offsetof(XbLds, stage) + 16u * (uint32_t)lane - mstart, 0xFFFFFFFFu);
                const uint2 ld = make_uint2(litD - la, 0xFFFFFFFFu);
                if (litIn || contL) sh.table[contL ? 0u : hb] = ld;
                if (mIn || contM) sh.table[contM ? 0u : hb + (litIn ? 1u : 0u)] = md;
                if (litIn) atomicOr(&sh.bits[(lstart - P) >> 5], 1u << ((lstart - P) & 31));
                if (mIn) atomicOr(&sh.bits[(mstart - P) >> 5], 1u << ((mstart - P) & 31));
                // passes that hold a byte of a window match with a very small or very large offset
                // (masked stores on purpose: 64 lanes on one LDS word serialise -- unconditional stores with a spare slot
                // for the lanes without a head made the kernel 12 % slower)
                const bool spec = (mIn || contM) && md.y == kXbWin - 1 && ringSpecial;
                if (wave_any(spec)) {
                    if (spec) {
                        const uint32_t f0 = (max(mstart, P) - P) >> 6, f1 = (min(mstart + ML, sEnd) - 1 - P) >> 6;
                        atomicOr(&sh.special, (2u << f1) - (1u << f0));
                    }
                }
#ifndef MZD_ABL_XB_LATE  /* ablation, timing only (wrong bytes): what the passes cost when the stretch's loads travel behind them */
                xb_wait_vm();  // the staged bytes and the literals are here, and so is every window unit issued before
                confirmed = flushed;
                if (stg) *(uint4 *)&sh.stage[16 * lane] = make_uint4(sv.x, sv.y, sv.z, sv.w);
                if (!litRle) *(uint64_t *)&sh.lit[8 * lane] = lv;
#else
                confirmed = flushed;
#endif
                while (P - flushed >= kXbFlush) xb_flush_step(sh, out, flushed, lane);
                uint32_t special = __builtin_amdgcn_readfirstlane((int)sh.special);
                XB_STAT(1, 1);
                XB_STAT(6, __popcll(wave_ballot(stg)));
                XB_STAT(9, XB_CLOCK() - xb_t2);
                const unsigned long long xb_t3 = XB_CLOCK();
                (void)xb_t3;

                uint32_t sbase = 1;
                const uint64_t *hbits = (const uint64_t *)sh.bits;
                for (uint32_t k = 0; P < sEnd; k++, P += 64, p += 64, special >>= 1) {
                    const uint64_t H = hbits[k];
                    const uint32_t own = xb_owner(H, sbase, lblo, lbhi);
                    sbase = (uint32_t)__builtin_amdgcn_readlane((int)own, 63) + 1u;
                    const uint2 e = sh.table[own];
                    const uint32_t s = p + e.x;
                    uint32_t val = lds[s & e.y];
                    if (special & 1u) {
                        // window matches of this pass: made by the pass itself (offset <= lane) or behind the ring
                        const uint32_t n = sEnd - P;
                        const bool act = (uint32_t)lane < n && e.y == kXbWin - 1;
                        const int r = act ? (int)(e.x + (uint32_t)lane) : -1;  // lane - offset
                        const bool far = act && r < -kXbNear;
                        XB_STAT(3, 1);
                        XB_STAT(4, wave_any(r >= 0));
                        XB_STAT(5, wave_any(far));
                        if (wave_any(far)) {
                            xb_wait_vm();  // every window unit issued so far has arrived in the slab
                            if (far) val = out[s];
                        }
                        if (wave_any(r >= 0)) {
                            // nearly always one level deep: the source lane's byte is there already
                            const uint32_t srcl = r >= 0 ? (uint32_t)r : (uint32_t)lane;
                            const uint32_t v2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcl << 2), (int)val);
                            const int r2 = __builtin_amdgcn_ds_bpermute((int)(srcl << 2), r);
                            if (!wave_any(r >= 0 && r2 >= 0)) val = r >= 0 ? v2 : val;
                            else val = xb_resolve_in_pass(val, r, (uint32_t)lane);
                        }
                    } else {
                        XB_STAT(2, 1);
                    }
                    sh.win[p & v7ff] = (uint8_t)val;  // (beyond the tile's end: bytes the next tile overwrites before anything reads them)
                }
#ifdef MZD_ABL_XB_LATE
                xb_wait_vm();
                if (stg) *(uint4 *)&sh.stage[16 * lane] = make_uint4(sv.x, sv.y, sv.z, sv.w);
                if (!litRle) *(uint64_t *)&sh.lit[8 * lane] = lv;
#endif
                P = sEnd;  // (a stretch cut short by its literals ends inside a pass)
                p = P + (uint32_t)lane;
                if (lane < (int)(kXbStretch / 64)) ((uint64_t *)sh.bits)[lane] = 0ull;  // every head of the stretch has been used
                if (lane == 0) sh.special = 0u;
                la = lb;
                XB_STAT(10, XB_CLOCK() - xb_t3);
            }
            tileStart = E;
            litRun += tileLits;
        }
        outPos += blockOut;  // (also for a block that failed on an offset: the length k_exec reports)
        if (error != MZD_OK) break;
        {
            // offset history carried to the next block (framedecompressor.go:23; persists across blocks)
            const int n0 = resolve_hist(bsum.hist[0], H0, H1, H2);
            const int n1 = resolve_hist(bsum.hist[1], H0, H1, H2);
            const int n2 = resolve_hist(bsum.hist[2], H0, H1, H2);
            H0 = n0; H1 = n1; H2 = n2;
        }
    }
    if (error == MZD_OK) xb_flush_bytes(sh, out, flushed, outPos, lane);
#ifdef MZD_XB_STATS
    xbst[11] = XB_CLOCK() - xb_t0;
    xbst[12] = 1;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_xb_stats[i], xbst[i]);
#endif
    if (lane == 0) {
        int e = error;
        if (e == MZD_OK && fr.content_size != MZD_UNKNOWN_SIZE && outPos != fr.content_size) e = MZD_ERR_DST_FULL;
        frame_status[fidx] = e;
        frame_out_len[fidx] = outPos;
    }
}

}  // namespace mzd
