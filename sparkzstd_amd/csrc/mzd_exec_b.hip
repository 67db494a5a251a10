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
//                   staged match   32 source bytes per sequence, loaded from the frame's slab into LDS by the setup:
//                                  every match whose source is final in memory.  All of a stretch's loads are in
//                                  flight together; no pass waits for memory
//                   window match   the last 4 KiB of output live in an LDS ring (position & 0xfff, that is M)
//             and a pass is: read its 64 bits of the bitmap, count the heads at or below every lane (v_mbcnt), read
//             the table entry, read the byte, store it into the window ring -- ~13 vector and 5 LDS instructions,
//             all aligned whole-wavefront ones.  Two rare cases take a longer pass (the setup marks the passes): a
//             window match made by its own pass (offset <= lane; resolved between the lanes by pointer jumping, six
//             rounds at most) and a window match that is neither staged nor in the ring any more (a long far match:
//             the pass reads the slab itself).  The ring leaves for the slab in 512-byte units.
//             No byte-misaligned LDS access, no validity bitmap, no barrier; 7.7 KiB of LDS per frame.
//
// Hazards are ordered by construction: a wavefront's LDS operations execute in order (a pass's reads precede its
// store); the slab is read only below `confirmed` (window units whose stores a wait on memory has seen complete) or
// after such a wait.
#pragma once

namespace mzd {

constexpr uint32_t kXbWin = 4096;      // window ring (what the stage cannot serve yet -- sources younger than the last wait on memory saw -- must still be in it)
constexpr uint32_t kXbLit = 512;       // literals of the current stretch
constexpr uint32_t kXbStage = 2048;    // staged matches of the current tile: 32 source bytes per sequence lane
constexpr uint32_t kXbStageMl = 32;    // longest match that is staged
constexpr uint32_t kXbStretch = 1024;  // output bytes per stretch at most (16 passes; one bit each in the bitmap)
constexpr uint32_t kXbFlush = 512;     // 64 lanes x 8 bytes leave for the slab at a time
constexpr int kXbNear = (int)kXbWin - 64;  // a window match byte less than this far behind its pass start is read from the ring

struct XbLds {
    uint8_t win[kXbWin];
    uint8_t lit[kXbLit];
    uint8_t stage[kXbStage];
    uint2 table[130];                 // [0] the run that continues from the stretch before, [1 + k] head k of the stretch, [129] spare
    uint32_t bits[kXbStretch / 32];   // heads
    uint32_t special;                 // passes with a window match the plain pass cannot serve (bit = pass of the stretch)
    uint32_t special2;                // passes with a byte that the pass before makes
    uint32_t pad[2];
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

// bytes [flushed, upto) of the frame leave the window for the slab, byte by byte (block ends, unaligned remainders).
// (`flushed` and the like travel by value: a variable whose address an outlined function takes lives in scratch memory, and
// every reload of it waits for ALL of the wavefront's memory operations -- the prefetched ones included)
__device__ __noinline__ uint32_t xb_flush_bytes(XbLds &sh, uint8_t *out, uint32_t flushed, uint32_t upto, int lane)
{
    for (uint32_t x = flushed + (uint32_t)lane; x < upto; x += 64) out[x] = sh.win[x & (kXbWin - 1)];
    return upto;
}

// the 512-byte unit at `flushed` (or the bytes up to the next unit boundary) leaves for the slab; -> the new `flushed`
__device__ __forceinline__ uint32_t xb_flush_step(XbLds &sh, uint8_t *out, uint32_t flushed, int lane)
{
    if ((flushed & (kXbFlush - 1)) == 0) {
        const uint32_t x = flushed + 8u * (uint32_t)lane;
        const uint64_t v = *(const uint64_t *)&sh.win[x & (kXbWin - 1)];
        ((U64U *)(out + x))->v = v;
        return flushed + kXbFlush;
    }
    return xb_flush_bytes(sh, out, flushed, (flushed + kXbFlush) & ~(kXbFlush - 1), lane);
}

// after a bulk write straight to the slab (Raw / RLE blocks, literal-only blocks): the window ring takes the last
// bytes of the frame back from memory so that the next block's window matches find them
__device__ __noinline__ void xb_reload_window(XbLds &sh, const uint8_t *out, uint32_t outPos, int lane)
{
    xb_wait_vm();  // the bulk stores are in memory (same CU: visible to the loads below)
    const uint32_t lo = outPos > kXbWin ? outPos - kXbWin : 0u;
    const uint32_t lo4 = (lo + 3u) & ~3u;  // (the ring is trusted kXbNear + 63 bytes back: 1 to 3 bytes less than it holds here)
    const uint32_t hi4 = outPos & ~3u;
    for (uint32_t x = lo4 + 4u * (uint32_t)lane; x < hi4; x += 256) *(uint32_t *)&sh.win[x & (kXbWin - 1)] = ((const U32U *)(out + x))->v;
    for (uint32_t x = max(lo4, hi4) + (uint32_t)lane; x < outPos; x += 64) sh.win[x & (kXbWin - 1)] = out[x];
}

__device__ __noinline__ void xb_bulk_copy(uint8_t *dst, const uint8_t *src, uint32_t n, int lane)
{
    const uint32_t n16 = n >> 4;
    for (uint32_t i = (uint32_t)lane; i < n16; i += 64) *(U128U *)(dst + 16 * i) = *(const U128U *)(src + 16 * i);
    for (uint32_t i = (n16 << 4) + (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
}
__device__ __noinline__ void xb_bulk_fill(uint8_t *dst, uint32_t byte, uint32_t n, int lane)
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

// BM (block mode, mzd_exec_blk.hip): the wavefront's job is ONE SEGMENT: block `first + blockIdx.x` of the batch when it is a
// head (kBjHead) and the blocks after it up to the next head -- a few consecutive blocks WITH sequences of one frame, executed in
// order (a Raw / RLE / literal-only block is a job of its own); `out_blob` is the plane of this pass.  Whatever lies before the
// segment's start S is read from the pass's pattern `bk.pat` instead of the slab -- the ring is preloaded with it, staged and
// far reads below S go to it.
struct XbBlk {
    const BJob *jobs;     // per block
    const uint32_t *heads;  // the jobs: [0] their number, [1 + j] the first block of job j
    BFrame *bframes;
    const uint8_t *pat;   // this pass's pattern, indexed by the frame-relative position (np > 0: pass 0's; the others follow at pstride)
    uint32_t pass;        // np == 0: the launch is ONE pass, this one
    // np > 0: the launch is ALL np passes -- workgroup b is pass b % np of job b / np (a job's passes next to each other: they read
    // the same records and literals), pass p > 0 writes to planes + (p - 1) * plane_stride instead of the kernel's output blob
    uint32_t np;
    uint64_t pstride, plane_stride;
    uint8_t *planes;
};

#ifdef MZD_TEST_KERNELS  /* round 6: a second implementation for the parity tests (libmzd_test.so); the helpers above are k_exec_c's too */
template <bool BM>
__global__ __launch_bounds__(64, 5) void k_exec_b(const uint8_t *__restrict__ in, uint8_t *out_blob, const DFrame *__restrict__ frames,
                                                  const DBlock *__restrict__ blocks, const BlockSum *__restrict__ sums,
                                                  const uint64_t *__restrict__ recs, const uint8_t *__restrict__ litbuf,
                                                  int32_t *frame_status, uint64_t *frame_out_len,
                                                  const uint32_t *__restrict__ order, uint32_t first, XbBlk bk)
{
    __shared__ __attribute__((aligned(16))) XbLds sh;
#ifdef MZD_SHIFT_XB  /* experiment: the whole instruction stream four bytes later */
    asm volatile("s_nop 0");
#endif
    const uint8_t *const lds = (const uint8_t *)&sh;
    const int lane = threadIdx.x;
    // this wavefront's frame: in the batch's execution order when it has one (heterogeneous batches: the largest first)
    uint32_t fidx, bi0 = 0;
    BJob jb{};
    if (BM) {
        // the job list k_blk_scan made: heads[0] = number of jobs, heads[1 + j] = the block job j starts at.  (A wavefront per
        // BLOCK that exits unless its block starts a job put the jobs on every second workgroup -- and so on half of the CUs:
        // a pass took twice as long.)
        if (blockIdx.x >= bk.heads[0]) return;
        const uint32_t g = bk.heads[1 + blockIdx.x];
        jb = bk.jobs[g];
        fidx = jb.frame;
        bi0 = g;  // (global for now)
    } else {
        fidx = order ? order[first + blockIdx.x] : first + blockIdx.x;
    }
    const DFrame fr = frames[fidx];
    uint8_t *out = out_blob + fr.out_offset;
    if (BM) {
        bi0 -= fr.first_block;
        // the passes after the first are for segments that can derive bytes from before their start
        if ((jb.flags & kBjSkip) || (bk.pass > 0 && (bi0 == 0 || (jb.flags & kBjDirect)))) return;
        // the pass of the position's high bits is for frames whose matches may reach back 8 MiB or more (k_blk_scan)
        if (bk.pass == 3 && !bk.bframes[fidx].high) return;
    }
    const uint32_t S = BM ? jb.start : 0u;  // the block's first byte (block mode)
    const uint8_t *const pat = bk.pat;

    int error = BM ? (int)MZD_OK : fr.plan_status;
    uint32_t outPos = S;         // bytes of this frame produced so far (frames of 4 GiB and more take k_exec)
    uint32_t flushed = S;        // [0, flushed) has left for the slab (the youngest units may still be in flight)
    uint32_t confirmed = S;      // [0, confirmed) has ARRIVED in the slab: a wait on memory came after its stores
    int H0 = 1, H1 = 4, H2 = 8;  // framedecompressor.go:48,59
    if (BM) {
        H0 = jb.H0;
        H1 = jb.H1;
        H2 = jb.H2;
    }
    if (lane < (int)(kXbStretch / 32)) sh.bits[lane] = 0u;
    if (lane == 0) sh.special = sh.special2 = 0u;
    if (BM && S > 0 && !(jb.flags & kBjDirect)) xb_reload_window(sh, pat, S, lane);  // the ring's view of the frame before the segment
    uint32_t bi = bi0;  // (declared out here: block mode reports the block an error belongs to -- every error leaves the loop by `break`)
    // constants of the passes, in VGPRs (a vector instruction with a literal or scalar operand issues at half rate)
    uint32_t vwmask = kXbWin - 1;
    uint32_t lblo = lane < 32 ? 1u << lane : 0u, lbhi = lane < 32 ? 0u : 1u << (lane - 32);
    asm volatile("" : "+v"(vwmask), "+v"(lblo), "+v"(lbhi));
#ifdef MZD_XB_STATS
    unsigned long long xbst[16] = {0};
    const unsigned long long xb_t0 = XB_CLOCK();
#endif

    for (; bi < fr.n_blocks && error == MZD_OK; bi++) {
        // (block mode: the job ends where the next one starts, or where the frame ended)
        if (BM && bi > bi0 && (bk.jobs[fr.first_block + bi].flags & (kBjHead | kBjSkip))) break;
        const DBlock b = blocks[fr.first_block + bi];
        if (b.type != MZD_BLOCK_COMPRESSED) {
            // Raw (framedecompressor.go:211-215) / RLE (:229-241): straight copy / fill
            if ((uint64_t)outPos + b.size > fr.out_capacity) {
                error = MZD_ERR_DST_FULL;
                break;
            }
            flushed = xb_flush_bytes(sh, out, flushed, outPos, lane);
            if (b.type == MZD_BLOCK_RAW) xb_bulk_copy(out + outPos, in + b.src_off, b.size, lane);
            else xb_bulk_fill(out + outPos, in[b.src_off], b.size, lane);
            outPos += b.size;
            if (!BM) xb_reload_window(sh, out, outPos, lane);
            flushed = confirmed = outPos;
            continue;
        }

        const BlockSum bsum = sums[fr.first_block + bi];
        const uint32_t litTotal = bsum.lit_total, seqOut = bsum.out_total;
        int err = bsum.huf_err != 0xFFFFFFFFu ? (int)(bsum.huf_err & 0xFF) : bsum.status;
        if (err == MZD_OK && b.n_seq == 0) err = b.pad[1];  // (zero sequences in the two-byte form: the planner's verdict, sequences.go:126-208)
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
            flushed = xb_flush_bytes(sh, out, flushed, outPos, lane);
            if (!b.pad[0]) {
                if (litRle) xb_bulk_fill(out + outPos, lits[0], b.lit_regen, lane);
                else xb_bulk_copy(out + outPos, lits, b.lit_regen, lane);
            }
            outPos += b.lit_regen;
            if (!BM) xb_reload_window(sh, out, outPos, lane);
            flushed = confirmed = outPos;
            continue;
        }
        if (litRle) {  // RLE literals (literals.go:390-396): every literal of every stretch is this byte
            const uint64_t v = lits[0] * 0x0101010101010101ull;
            *(uint64_t *)&sh.lit[8 * lane] = v;
        }

        // ---- tiles of 64 sequences; the literals after the last sequence (sequence_execution.go:55-59) ride along as
        // one more sequence without a match.  The loop is a SOFTWARE PIPELINE over stretches: the loads of the next
        // stretch (its staged matches, its literals) are issued before the passes of the current one, so that a wavefront
        // meets its own memory latency once per block instead of once per stretch (SQ_WAIT_ANY was 70 % of the wave cycles).
        const uint32_t rest = b.lit_regen - litTotal;
        const uint32_t nps = b.n_seq + (rest ? 1u : 0u);
        const uint32_t ntiles = (nps + 63) >> 6;
        const uint64_t *brec = recs + b.rec_off;

        // per-lane state of a tile (one sequence per lane)
        struct Tile {
            uint32_t LL, ML, lstart, mstart, lsrc, litD;
            int off;
            uint2 md;          // a match byte at position p comes from LDS offset (p + md.x) & md.y: the window ring until a stretch stages it
            uint32_t start, E, lits, litRun;  // wave-uniform: the tile's first byte, the byte after its last, its literals, literals before it
        };
        // a stretch whose loads are in flight.  Two of them, used alternately (the loop body below is instantiated for both
        // orders): a register that a load is still filling is never copied -- a copy would wait for the load on the spot
        struct Plan {
            uint32_t P, sEnd, la, lb;
            uint64_t stg;      // lanes whose match goes through the stage
            U128U sv, sv2;     // ... and its 32 source bytes
            uint64_t lv;       // the stretch's literals, 8 per lane
            uint64_t rec;      // the records of the tile AFTER the stretch's tile
        };
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        auto load_recs = [&](uint32_t tile) -> uint64_t {
            const uint32_t si = tile * 64 + (uint32_t)lane;
            return si < b.n_seq ? brec[si] : 0ull;
        };
        // tile t from its records; false: an offset beyond the produced data (ringbuffer.go:206-214)
        auto load_tile = [&](Tile &T, uint64_t rec, uint32_t t, uint32_t tileStart, uint32_t litRun) -> bool {
            const uint32_t si = t * 64 + (uint32_t)lane;
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
            T.lits = (uint32_t)__builtin_amdgcn_readlane((int)sLL, 63);
            T.E = tileStart + (uint32_t)__builtin_amdgcn_readlane((int)sOut, 63);
            T.start = tileStart;
            T.litRun = litRun;
            T.LL = LL;
            T.ML = ML;
            T.off = off;
            T.mstart = tileStart + sOut - ML;  // frame-relative
            T.lstart = T.mstart - LL;
            T.lsrc = litRun + sLL - LL;        // the sequence's first literal (index in the block)
            T.md = make_uint2((uint32_t)(-off), kXbWin - 1);
            T.litD = (uint32_t)offsetof(XbLds, lit) + T.lsrc - T.lstart;  // literal byte p: LDS offset p + litD - la
            XB_STAT(0, 1);
            XB_STAT(7, __popcll(wave_ballot(isSeq && ML > 0)));
            return !wave_any(isSeq && ML > 0 && (off <= 0 || (uint32_t)off > T.mstart));
        };
        // the stretch of tile T that starts at P with the literal cursor at la: its extent, and its loads on their way
        auto plan_stretch = [&](const Tile &T, uint32_t P, uint32_t la, Plan &N) {
            uint32_t sEnd = min(T.E, P + kXbStretch);
            // literal cursor at the stretch's end: the last sequence that starts at or before it (the lanes' starts ascend;
            // lane 0 starts at the tile's start); a stretch that would need more than 512 literals ends where the 512th does
            uint32_t lb;
            {
                const int k = 63 - __builtin_clzll(wave_ballot(T.lstart <= sEnd));
                const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)T.lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)T.lsrc, k);
                const uint32_t kn = (uint32_t)__builtin_amdgcn_readlane((int)T.LL, k);
                lb = ks + min(kn, sEnd - kl);
            }
            if (lb - la > kXbLit) {
                const int k = 63 - __builtin_clzll(wave_ballot(T.lsrc <= la + kXbLit));  // the run that holds literal la + 512
                const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)T.lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)T.lsrc, k);
                sEnd = kl + (la + kXbLit - ks);
                lb = la + kXbLit;
            }
            N.P = uni(P);
            N.sEnd = uni(sEnd);
            N.la = uni(la);
            N.lb = uni(lb);
            // matches whose source is final in the slab: 16 source bytes into the stage, all of the stretch's loads in flight
            // together (a match not farther back than the ring reaches is served by the ring whatever pass it falls into)
            const bool mIn = T.ML > 0 && T.mstart - N.P < N.sEnd - N.P;
            const uint32_t q0 = T.mstart - (uint32_t)T.off;
            // (block mode: a source that straddles the block's start is left to the pass)
            const bool stg = mIn && T.ML <= kXbStageMl && T.off > kXbNear && q0 + T.ML <= confirmed && (!BM || q0 >= S || q0 + T.ML <= S);
            const uint8_t *const rb = BM && q0 < S ? pat : (const uint8_t *)out;
            N.stg = wave_ballot(stg);
            N.sv = N.sv2 = U128U{0, 0, 0, 0};
#ifndef MZD_ABL_XB_NOSTAGE  /* ablation, timing only (wrong bytes): the kernel without its scattered reads of the slab */
            if (stg) N.sv = *(const U128U *)(rb + q0);
            if (stg && T.ML > 16) N.sv2 = *(const U128U *)(rb + q0 + 16);
#endif
            // the stretch's literals: [la, lb) of the block's literals
            N.lv = 0;
            const uint32_t li = N.la + 8u * (uint32_t)lane;
            if (!litRle && li < N.lb) N.lv = ((const U64U *)(lits + li))->v;  // (lb <= lit_regen)
        };

        Tile T;
        uint32_t t = 0;
        // One step of the pipeline: finish the setup of stretch C (its loads were issued a step ago), plan stretch N and issue
        // its loads, run C's passes.  -> false when C was the block's last stretch (or the block failed).
        auto step = [&](Plan &C, Plan &N) -> bool {
            // ---- the current stretch: everything a pass looks up goes to LDS
            const unsigned long long xb_t2 = XB_CLOCK();
            (void)xb_t2;
            const uint32_t P0 = C.P, sEnd = C.sEnd, sLen = sEnd - P0, la = C.la;
            const bool stg = (C.stg >> lane) & 1;
            const bool litIn = T.LL > 0 && T.lstart - P0 < sLen;   // the sequence's literal run / match starts in this stretch
            const bool mIn = T.ML > 0 && T.mstart - P0 < sLen;
            const bool contL = T.lstart < P0 && P0 < T.mstart;     // ... or continues from the stretch before (one lane at most)
            const bool contM = T.mstart < P0 && P0 < T.mstart + T.ML;
            // table index of the sequence's heads: 1 + the heads of the lanes below (+ its own literal head)
            const uint64_t litMask = wave_ballot(litIn), mMask = wave_ballot(mIn);
            const uint32_t hb = __builtin_amdgcn_mbcnt_hi((uint32_t)(litMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)litMask, 1u)) +
                                __builtin_amdgcn_mbcnt_hi((uint32_t)(mMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mMask, 0u));
            if (stg) T.md = make_uint2((uint32_t)offsetof(XbLds, stage) + 32u * (uint32_t)lane - T.mstart, 0xFFFFFFFFu);
            const uint2 ld = make_uint2(T.litD - la, 0xFFFFFFFFu);
            if (litIn || contL) sh.table[contL ? 0u : hb] = ld;
            if (mIn || contM) sh.table[contM ? 0u : hb + (litIn ? 1u : 0u)] = T.md;
            if (litIn) atomicOr(&sh.bits[(T.lstart - P0) >> 5], 1u << ((T.lstart - P0) & 31));
            if (mIn) atomicOr(&sh.bits[(T.mstart - P0) >> 5], 1u << ((T.mstart - P0) & 31));
            // passes in which a window match needs more than the plain pass does: a byte made by the pass itself (offset <= its
            // lane) or lying behind the ring -> `special`; a byte made by the pass before (the plain passes run in pairs whose
            // bytes are read together) -> `special2`.  Per match: its first pass holds lanes lo0 .. hi0, the passes in between
            // all lanes, its last pass lanes 0 .. hiL.  (masked stores on purpose: 64 lanes on one LDS word serialise)
            const bool spec = (mIn || contM) && T.md.y == kXbWin - 1 && (T.off < 128 || T.off > kXbNear);
            if (wave_any(spec)) {
                if (spec) {
                    const uint32_t a = max(T.mstart, P0) - P0, z = min(T.mstart + T.ML, sEnd) - 1 - P0;  // first and last byte, stretch-relative
                    const uint32_t f0 = a >> 6, f1 = z >> 6, lo0 = a & 63, hiL = z & 63, hi0 = f1 > f0 ? 63u : hiL;
                    const uint32_t off = (uint32_t)T.off;
                    const uint32_t mF = 1u << f0, mL = f1 > f0 ? 1u << f1 : 0u, mM = f1 > f0 + 1 ? (1u << f1) - (2u << f0) : 0u;
                    const bool far = off > (uint32_t)kXbNear;
                    uint32_t s1 = 0, s2 = 0;
                    if (hi0 >= off || (far && off > lo0 + (uint32_t)kXbNear)) s1 |= mF;
                    if (off <= 63 || far) s1 |= mM;
                    if (hiL >= off || far) s1 |= mL;
                    if (off > lo0 && off <= hi0 + 64) s2 |= mF;
                    if (off <= 127) s2 |= mM;
                    if (off <= hiL + 64) s2 |= mL;
                    atomicOr(&sh.special, s1);
                    atomicOr(&sh.special2, s2 & ~s1);
                }
            }
            xb_wait_vm();  // the staged bytes, the literals and the next tile's records are here; so is every window unit issued before
            confirmed = uni(flushed);
            if (stg) {
                *(uint4 *)&sh.stage[32 * lane] = make_uint4(C.sv.x, C.sv.y, C.sv.z, C.sv.w);
                *(uint4 *)&sh.stage[32 * lane + 16] = make_uint4(C.sv2.x, C.sv2.y, C.sv2.z, C.sv2.w);
            }
            if (!litRle) *(uint64_t *)&sh.lit[8 * lane] = C.lv;
            uint32_t special = uni(sh.special), special2 = uni(sh.special2);
            XB_STAT(1, 1);
            XB_STAT(6, __popcll(C.stg));
            XB_STAT(9, XB_CLOCK() - xb_t2);

            // ---- the NEXT stretch (of this tile, or the first of the next tile): planned, its loads issued
            Tile Tn;
            bool nextTile = false, haveNext = false;
            if (sEnd < T.E) {
                plan_stretch(T, sEnd, C.lb, N);
                N.rec = C.rec;
                haveNext = true;
            } else if (t + 1 < ntiles) {
                nextTile = true;
                if (!load_tile(Tn, C.rec, t + 1, T.E, T.litRun + T.lits)) {
                    error = MZD_ERR_OFFSET;
                } else {
                    plan_stretch(Tn, Tn.start, Tn.litRun, N);
                    N.rec = load_recs(t + 2);
                    haveNext = true;
                }
            }
            // the window units that are complete leave for the slab -- AFTER the next stretch's loads: whatever waits on
            // memory next (the step after this one) then finds loads and stores a whole stretch old
            {
                uint32_t fl = uni(flushed);
                while (P0 - fl >= kXbFlush) fl = uni(xb_flush_step(sh, out, fl, lane));
                flushed = fl;
            }

            // ---- the passes
            const unsigned long long xb_t3 = XB_CLOCK();
            (void)xb_t3;
            uint32_t sbase = 1;
            const uint64_t *hbits = (const uint64_t *)sh.bits;
            uint32_t P = P0, p = P0 + (uint32_t)lane;
            uint32_t k = 0;
            // two passes at a time while neither is marked: their bitmap words, table entries and bytes are read together, so
            // that a wavefront meets the LDS latency once per pair (the passes are a chain of three dependent LDS reads)
            while (P + 128 <= sEnd && ((special & 3u) | (special2 & 2u)) == 0) {
                const uint64_t Ha = hbits[k], Hb = hbits[k + 1];
                const uint32_t owna = xb_owner(Ha, sbase, lblo, lbhi);
                const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)owna, 63) + 1u;
                const uint32_t ownb = xb_owner(Hb, sb, lblo, lbhi);
                sbase = (uint32_t)__builtin_amdgcn_readlane((int)ownb, 63) + 1u;
                const uint2 ea = sh.table[owna], eb = sh.table[ownb];
                const uint32_t va = lds[(p + ea.x) & ea.y], vb = lds[(p + 64 + eb.x) & eb.y];
                sh.win[p & vwmask] = (uint8_t)va;
                sh.win[(p + 64) & vwmask] = (uint8_t)vb;
                XB_STAT(2, 2);
                k += 2;
                P += 128;
                p += 128;
                special >>= 2;
                special2 >>= 2;
            }
            for (; P < sEnd; k++, P += 64, p += 64, special >>= 1, special2 >>= 1) {
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
                        if (far) val = (BM && s < S ? pat : (const uint8_t *)out)[s];
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
                sh.win[p & vwmask] = (uint8_t)val;  // (beyond the stretch's end: bytes the next stretch overwrites before anything reads them)
            }
            if (lane < (int)(kXbStretch / 64)) ((uint64_t *)sh.bits)[lane] = 0ull;  // every head of the stretch has been used
            if (lane == 0) sh.special = sh.special2 = 0u;
            XB_STAT(10, XB_CLOCK() - xb_t3);
            if (error != MZD_OK) return false;
            if (nextTile) {
                T = Tn;  // (computed values only: nothing here is waiting for memory)
                t++;
            }
            return haveNext;
        };

        Plan A, B;
        if (!load_tile(T, load_recs(0), 0, outPos, 0)) {
            error = MZD_ERR_OFFSET;
        } else if (ntiles > 0) {
            plan_stretch(T, T.start, 0, A);
            A.rec = load_recs(1);
            while (step(A, B) && step(B, A)) {
            }
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
    if (error == MZD_OK) flushed = xb_flush_bytes(sh, out, flushed, outPos, lane);
    if (BM) {
        // an offset beyond the produced data (the one defect the scan cannot see): the frame ends at its first such block
        if (lane == 0 && error != MZD_OK && bk.pass == 0) atomicMin(&bk.bframes[fidx].first_bad, bi);
        return;
    }
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

#endif  // MZD_TEST_KERNELS

}  // namespace mzd
