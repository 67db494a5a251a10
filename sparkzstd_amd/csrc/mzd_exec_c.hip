// mzd_exec_c.hip -- k_exec_c: sequence execution with ONE WAVEFRONT per frame, TWO output bytes per lane and pass, and a
// pass that needs no per-pass bookkeeping from its setup.
//
// Replaces decompression/sequence_execution.go:14-63 (ExecuteSequences), ringbuffer.go:102-277 (Push / Repeat /
// RepeatBeforeIndex) and the Raw / RLE block arms framedecompressor.go:211-215,229-241, like k_exec and k_exec_b.  Same inputs
// (the 8-byte sequence records of the entropy stage, the regenerated literals), same statuses, same bytes.
//
// It is k_exec_b's method (output strictly in order; a head bitmap and a run table per stretch; matches whose source is final in
// memory staged through LDS a stretch ahead; a window ring) cut down to the instructions the method needs -- round 3 measured
// that kernel at ~610 wavefront instructions per 64 sequences, a third of them scalar, with no unit of the CU saturated: what
// bounds it is the number of instructions a wavefront has to get through, so this kernel is about that number:
//
//   * a PASS makes 128 bytes: lane j makes bytes P + j and P + 64 + j.  One 16-byte read brings both halves' head bits, the
//     owners come from v_mbcnt, a table entry is ONE dword e and the source of byte p is LDS address
//     bfi(0xfff, p + e, e) -- the low twelve bits wrap inside a 4 KiB region, the bits above pick the region (0: the window
//     ring; 0x1000: literals and staged matches);
//   * bytes a pass makes from bytes of the SAME pass (a match closer than 128 bytes) are not singled out by the setup at all:
//     the pass reads its sources, stores, and reads again until nothing changes (a fixed point of "byte = its source byte" on
//     an acyclic dependence IS the serial result; one confirming round in the common case, a second for 45 % of the passes).
//     Runs that feed themselves at a short period (offset < length, RLE-like data) would take length / offset rounds: after
//     two rounds the pass resolves what is left by pointer jumping between the lanes instead (log2 rounds);
//   * conditional stores of the setup are predicated by ADDRESS (an LDS address beyond the allocation is dropped for free),
//     not by exec masks: no scalar bookkeeping around them;
//   * a tile of 64 sequences that is one stretch (99 % of them on text-like data) takes a setup path without the
//     stretch-splitting searches.
//
// Hazards are ordered by construction as in k_exec_b: a wavefront's LDS operations execute in order; the slab is read only
// below `confirmed` (window units whose stores a wait on memory has seen complete) or after such a wait.
#pragma once

namespace mzd {

// The window ring's size WIN is a template parameter of the kernel (round 5): 4 KiB at 20 frames per CU for heterogeneous batches and
// small frames, where the frames in flight count, 8 KiB at 16 frames per CU for batches of frames of one size from 32 KiB on and for
// block mode, where the matches the ring serves count (text: 48 % of the matches lie beyond 4 KiB, 36 % beyond 8 KiB).
constexpr uint32_t kXcLit = 512;       // literals of the current stretch
constexpr uint32_t kXcStageMl = 32;    // longest match that is staged: 32 source bytes per sequence lane
// The staged matches PACKED: 8-byte slots dealt in lane order, ceil(ML / 8) per match (text: 31 staged matches of 8 bytes on average per
// stretch = 45 slots), instead of 32 bytes per lane whether it stages anything or not -- 768 bytes instead of 2 048, which is what lets
// an 8 KiB ring keep 16 frames on a CU (10.1 KB per frame).  A match that finds no slot is left to the pass, like one that is too long.
constexpr uint32_t kXcStageSlots = 96;
constexpr uint32_t kXcStage = 8 * kXcStageSlots;
constexpr uint32_t kXcStretch = 1024;  // output bytes per stretch at most (8 passes)
constexpr uint32_t kXcFlush = 512;     // 64 lanes x 8 bytes leave for the slab at a time
constexpr uint32_t kXcPass = 128;
#ifndef MZD_XC_OOR
#define MZD_XC_OOR 0x00FF0000u
#endif
constexpr uint32_t kXcOor = MZD_XC_OOR;   // an LDS address no workgroup has: stores to it are dropped (tools/ubench k_pred<2>)
// MZD_XC_NT: which of the kernel's read-once streams carry the `nt` bit (1 records, 2 literals, 4 staged match sources)
#ifndef MZD_XC_NT
#define MZD_XC_NT 0
#endif
constexpr bool kXcNtRec = (MZD_XC_NT & 1) != 0, kXcNtLit = (MZD_XC_NT & 2) != 0, kXcNtStage = (MZD_XC_NT & 4) != 0;

template <uint32_t WIN>
struct XcLds {
    uint8_t win[WIN];               // window ring at LDS offset 0 of the frame's block (a power of two, 4 KiB at least)
    uint8_t lit[kXcLit];
    uint8_t stage[kXcStage];
    uint32_t bits[kXcStretch / 32];   // heads, one bit per output byte of the stretch; a pass reads four dwords
    uint32_t table[132];              // [0] the run that continues from the stretch before, [1 + k] head k of the stretch
    uint32_t pad[4];
};
template <uint32_t WIN>
constexpr bool xc_layout_ok()
{
    return offsetof(XcLds<WIN>, stage) + kXcStage <= 2 * WIN &&  // the stage inside region 1
           offsetof(XcLds<WIN>, win) == 0 && offsetof(XcLds<WIN>, lit) == WIN && offsetof(XcLds<WIN>, bits) % 128 == 0 && sizeof(XcLds<WIN>) % 16 == 0 &&
           kXcStretch + kXcPass + kXcFlush <= WIN;  // a window unit is issued before the ring wraps onto it
}
static_assert(xc_layout_ok<4096>() && xc_layout_ok<8192>(), "ring = region 0, literals and stage inside region 1, bitmap on a 128-byte boundary");

#ifdef MZD_XC_STATS
// tools/xc_stats.py: 0 tiles, 1 stretches, 2 passes, 3 extra fixed-point rounds, 4 passes resolved by pointer jumping,
// 5 passes with a byte read from memory, 6 staged matches, 7 matches, 8 cycles setup, 9 cycles plan + flush, 10 cycles passes,
// 11 cycles total, 12 frames, 13 stretches on the general setup path, 14 cycles in blocks without sequences, 15 such blocks
__device__ unsigned long long g_xc_stats[16];
#define XC_STAT(i, n) (xcst[i] += (unsigned long long)(n))
#define XC_CLOCK() __builtin_readcyclecounter()
#else
#define XC_STAT(i, n) do { } while (0)
#define XC_CLOCK() 0ull
#endif

// bytes [flushed, upto) of the frame leave the window for the slab, byte by byte (block ends, unaligned remainders)
template <uint32_t WIN>
__device__ __noinline__ uint32_t xc_flush_bytes(const uint8_t *win, uint8_t *out, uint32_t flushed, uint32_t upto, int lane)
{
    for (uint32_t x = flushed + (uint32_t)lane; x < upto; x += 64) out[x] = win[x & (WIN - 1)];
    return upto;
}
// the 512-byte unit at `flushed` (or the bytes up to the next unit boundary) leaves for the slab; -> the new `flushed`
template <uint32_t WIN>
__device__ __forceinline__ uint32_t xc_flush_step(const uint8_t *win, uint8_t *out, uint32_t flushed, int lane)
{
    if ((flushed & (kXcFlush - 1)) == 0) {
        const uint32_t x = flushed + 8u * (uint32_t)lane;
        const uint64_t v = *(const uint64_t *)&win[x & (WIN - 1)];
        ((U64U *)(out + x))->v = v;
        return flushed + kXcFlush;
    }
    return xc_flush_bytes<WIN>(win, out, flushed, (flushed + kXcFlush) & ~(kXcFlush - 1), lane);
}
// after a bulk write straight to the slab (Raw / RLE blocks, literal-only blocks): the window ring takes the last bytes of the
// frame back from memory so that the next block's window matches find them
template <uint32_t WIN>
__device__ __noinline__ void xc_reload_window(uint8_t *win, const uint8_t *out, uint32_t outPos, int lane)
{
    constexpr uint32_t kXcWin = WIN;
    xb_wait_vm();  // the bulk stores are in memory (same CU: visible to the loads below)
    const uint32_t lo = outPos > kXcWin ? outPos - kXcWin : 0u;
    const uint32_t lo4 = (lo + 3u) & ~3u;
    const uint32_t hi4 = outPos & ~3u;
    for (uint32_t x = lo4 + 4u * (uint32_t)lane; x < hi4; x += 256) *(uint32_t *)&win[x & (kXcWin - 1)] = ((const U32U *)(out + x))->v;
    for (uint32_t x = max(lo4, hi4) + (uint32_t)lane; x < outPos; x += 64) win[x & (kXcWin - 1)] = out[x];
}

// A SMALL block without sequences (Raw, RLE, literals only) joins the frame through the ring like a run of literals: n bytes from
// `src` (null: the byte `fill`) are written behind outPos, whole window units leave for the slab as they fill up (`store` false: the
// bytes are in the slab already -- literals the Huffman stage put in place -- and everything before them has been flushed); -> the
// new `flushed`.  The other way -- flush, bulk copy to the slab, wait, reload the ring from the slab -- is three trips to memory,
// 10 900 cycles a block whatever its size, and real data has frames made of hundreds of 1 KiB blocks (the reference's corpus:
// 22 % of its largest frame's time, 16 % of the stage's over all frames; tools/xc_stats.py).  (src may be read up to 7 bytes beyond
// n: the input blob, the literal scratch and the output blob all have that slack.)
#ifndef MZD_XC_RING_BLOCK
#define MZD_XC_RING_BLOCK 2048
#endif
constexpr uint32_t kXcRingBlock = MZD_XC_RING_BLOCK;  // blocks without sequences up to this size go through the ring
template <uint32_t WIN>
__device__ __noinline__ uint32_t xc_ring_append(uint8_t *win, uint8_t *out, const uint8_t *src, uint32_t fill, uint32_t n, uint32_t outPos,
                                                uint32_t flushed, bool store, int lane)
{
    // (n <= kXcRingBlock: four window units at most, all of the block's loads in flight together -- one trip to memory per block)
    constexpr int kUnits = kXcRingBlock / kXcFlush;
    const uint32_t o = 8u * (uint32_t)lane;
    uint64_t v[kUnits];
#pragma unroll
    for (int k = 0; k < kUnits; k++) {
        v[k] = fill * 0x0101010101010101ull;
        if (src && (uint32_t)k * kXcFlush + o < n) v[k] = ld64u(src + (uint32_t)k * kXcFlush + o);
    }
#pragma unroll
    for (int k = 0; k < kUnits; k++) {
        if ((uint32_t)k * kXcFlush >= n) break;
        const uint32_t c = min(kXcFlush, n - (uint32_t)k * kXcFlush);
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            if (o + j < c) win[(outPos + o + j) & (WIN - 1)] = (uint8_t)(v[k] >> (8 * j));
        outPos += c;
        if (store)
            while (outPos - flushed >= kXcFlush) flushed = xc_flush_step<WIN>(win, out, flushed, lane);
    }
    return store ? flushed : outPos;
}

__device__ __forceinline__ void xc_lds_write_b64(uint32_t addr, uint32_t lo, uint32_t hi)
{
    const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
#ifdef MZD_XC_NO_OOR  /* debugging: the predicated stores under exec masks instead */
__device__ __forceinline__ void xc_lds_write_b32(uint32_t addr, uint32_t v) { if (addr < kXcOor) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void xc_lds_or_b32(uint32_t addr, uint32_t v) { if (addr < kXcOor) asm volatile("ds_or_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
#else
__device__ __forceinline__ void xc_lds_write_b32(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void xc_lds_or_b32(uint32_t addr, uint32_t v) { asm volatile("ds_or_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
#endif
__device__ __forceinline__ uint32_t xc_bfi(uint32_t mask, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}

// What is left of a pass after two fixed-point rounds: bytes that derive from bytes of the same pass through long chains (a
// run that feeds itself at a short period).  Pointer jumping over the 128 bytes of the pass, two per lane: an element is DONE
// (its value final: its source lay before the pass, or has been taken from a done element) or points at the element of the
// pass it copies.  Every round halves the chains: seven rounds at most.  va / vb: what lane j holds for bytes j and 64 + j.
__device__ __noinline__ uint2 xc_resolve_in_pass(uint32_t va, uint32_t vb, int ra, int rb, uint32_t lane)
{
    // state of an element in one dword: value | done << 8 | source element << 16
    uint32_t sa = (va & 0xFF) | (ra < 0 ? 0x100u : 0u) | ((uint32_t)(ra < 0 ? 0 : ra) << 16);
    uint32_t sb = (vb & 0xFF) | (rb < 0 ? 0x100u : 0u) | ((uint32_t)(rb < 0 ? 0 : rb) << 16);
    for (int round = 0; round < 8; round++) {  // (seven halvings end any chain of 128; the bound is there because a hang is not a failure mode)
        const bool da = (sa >> 8) & 1, db = (sb >> 8) & 1;
        if (!wave_any(!da || !db)) break;
        const uint32_t qa = sa >> 16, qb = sb >> 16;  // source elements (0..127)
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qa & 63) << 2), (int)sa);
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qa & 63) << 2), (int)sb);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qb & 63) << 2), (int)sa);
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((qb & 63) << 2), (int)sb);
        const uint32_t ta = qa & 64 ? a1 : a0, tb = qb & 64 ? b1 : b0;  // the source element's state
        if (!da) sa = ta;  // its value and done flag if it is done, else its own source: the chain halves
        if (!db) sb = tb;
    }
    return make_uint2(sa & 0xFF, sb & 0xFF);
}

// BM (block mode, mzd_exec_blk.hip): as k_exec_b<true> -- the wavefront's job is ONE SEGMENT of a frame (a few consecutive blocks
// with sequences, from a block flagged kBjHead to the next; a Raw / RLE / literal-only block is a job of its own), `out_blob` is
// the plane of this pass, and whatever lies before the segment's start S is read from the pass's pattern `bk.pat`: the ring is
// preloaded with it, staged and far reads below S go to it.
template <bool BM, uint32_t WIN = 4096>
__global__ __launch_bounds__(64, WIN == 4096 ? 5 : 4) void k_exec_c(const uint8_t *__restrict__ in, uint8_t *out_blob, const DFrame *__restrict__ frames,
                                                  const DBlock *__restrict__ blocks, const BlockSum *__restrict__ sums,
                                                  const uint64_t *__restrict__ recs, const uint8_t *__restrict__ litbuf,
                                                  int32_t *frame_status, uint64_t *frame_out_len,
                                                  const uint32_t *__restrict__ order, uint32_t first, XbBlk bk, int32_t *frame_hist)
{
    constexpr uint32_t kXcWin = WIN;
    constexpr int kXcNear = (int)WIN - (int)kXcPass;  // a window match this close to its pass is served by the ring
    constexpr uint32_t kXcFarMark = 2 * WIN;          // table entry of a window match that is neither in the ring nor staged: | lane of its sequence
    typedef XcLds<WIN> XcLdsW;
    __shared__ __attribute__((aligned(128))) XcLdsW sh;
    const uint8_t *const lds = (const uint8_t *)&sh;
    const int lane = threadIdx.x;
    uint32_t fidx, bi0 = 0;
    BJob jb{};
    if (BM) {
        uint32_t job = blockIdx.x;
        if (bk.np) {  // every pass in one launch
            job = blockIdx.x / bk.np;
            bk.pass = blockIdx.x - job * bk.np;
            bk.pat += (size_t)bk.pass * bk.pstride;
            if (bk.pass) out_blob = bk.planes + (size_t)(bk.pass - 1) * bk.plane_stride;
        }
        if (job >= bk.heads[0]) return;  // (the job list k_blk_scan made: heads[0] jobs, heads[1 + j] = job j's first block)
        const uint32_t g = bk.heads[1 + job];
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
    // the segment's first byte (block mode); a chunk of a frame starts behind the window bytes its slab begins with (DFrame::start)
    const uint32_t S = BM ? jb.start : (fr.continues ? (uint32_t)fr.start : 0u);
    // (block mode: what lies before a chunk's FIRST job is not a pattern but those bytes themselves, the same in every pass)
    const uint8_t *const pat = BM && bi0 == 0 && fr.continues ? (const uint8_t *)out : bk.pat;

    int error = BM ? (int)MZD_OK : fr.plan_status;
    uint32_t outPos = S;         // bytes of this frame produced so far (frames of 4 GiB and more take k_exec)
    uint32_t flushed = S;        // [0, flushed) has left for the slab (the youngest units may still be in flight)
    uint32_t confirmed = S;      // [0, confirmed) has ARRIVED in the slab: a wait on memory came after its stores
    int H0 = 1, H1 = 4, H2 = 8;  // framedecompressor.go:48,59
    if (BM) {
        H0 = jb.H0;
        H1 = jb.H1;
        H2 = jb.H2;
    } else if (fr.continues) {
        H0 = fr.hist[0];
        H1 = fr.hist[1];
        H2 = fr.hist[2];
    }
    // LDS addresses of the areas the predicated stores go to (the kernel's only shared object: its offset is what the
    // compiler assigned, normally 0; the region arithmetic below is relative to it)
    const uint32_t ldsBase = (uint32_t)(uintptr_t)&sh;
    const uint32_t tabA = ldsBase + (uint32_t)offsetof(XcLdsW, table), bitsA = ldsBase + (uint32_t)offsetof(XcLdsW, bits);
    for (int i = lane; i < 132; i += 64) sh.table[i] = 0u;
    if (lane < (int)(kXcStretch / 32)) sh.bits[lane] = 0u;
    if (BM && S > 0 && !(jb.flags & kBjDirect)) xc_reload_window<WIN>(sh.win, pat, S, lane);  // the ring's view of the frame before the segment
    if (!BM && S > 0) xc_reload_window<WIN>(sh.win, out, S, lane);                              // ... before the chunk
    uint32_t bi = bi0;
    // constants of the passes, in VGPRs (a vector instruction with a literal or scalar operand issues at half rate)
    uint32_t vwmask = kXcWin - 1;
    uint32_t lblo = lane < 32 ? 1u << lane : 0u, lbhi = lane < 32 ? 0u : 1u << (lane - 32);
    uint32_t clrA = lane < (int)(kXcStretch / 64) ? bitsA + 8u * (uint32_t)lane : kXcOor;  // where the lane clears the bitmap after a stretch
    uint32_t vlane = (uint32_t)lane;
    asm volatile("" : "+v"(vwmask), "+v"(lblo), "+v"(lbhi), "+v"(clrA), "+v"(vlane));
#ifdef MZD_XC_STATS
    unsigned long long xcst[16] = {0};
    const unsigned long long xc_t0 = XC_CLOCK();
#endif

    for (; bi < fr.n_blocks && error == MZD_OK; bi++) {
        // (block mode: the job ends where the next one starts, or where the frame ended)
        if (BM && bi > bi0 && (bk.jobs[fr.first_block + bi].flags & (kBjHead | kBjSkip))) break;
        const unsigned long long xc_ta = XC_CLOCK();
        (void)xc_ta;
        const DBlock b = blocks[fr.first_block + bi];
        if (b.type != MZD_BLOCK_COMPRESSED) {
            // Raw (framedecompressor.go:211-215) / RLE (:229-241): straight copy / fill
            if ((uint64_t)outPos + b.size > fr.out_capacity) {
                error = MZD_ERR_DST_FULL;
                break;
            }
            if (!BM && b.size <= kXcRingBlock) {
                flushed = xc_ring_append<WIN>(sh.win, out, b.type == MZD_BLOCK_RAW ? in + b.src_off : nullptr,
                                              b.type == MZD_BLOCK_RAW ? 0u : (uint32_t)in[b.src_off], b.size, outPos, flushed, true, lane);
                outPos += b.size;
                XC_STAT(14, XC_CLOCK() - xc_ta);
                XC_STAT(15, 1);
                continue;
            }
            flushed = xc_flush_bytes<WIN>(sh.win, out, flushed, outPos, lane);  // (same ring: the first 4 KiB of the block)
            if (b.type == MZD_BLOCK_RAW) xb_bulk_copy(out + outPos, in + b.src_off, b.size, lane);
            else xb_bulk_fill(out + outPos, in[b.src_off], b.size, lane);
            outPos += b.size;
            if (!BM) xc_reload_window<WIN>(sh.win, out, outPos, lane);
            flushed = confirmed = outPos;
            XC_STAT(14, XC_CLOCK() - xc_ta);
            XC_STAT(15, 1);
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
            if (!BM && b.lit_regen <= kXcRingBlock) {
                if (b.pad[0]) {  // in place: the ring takes them from the slab, nothing is stored again
                    flushed = xc_flush_bytes<WIN>(sh.win, out, flushed, outPos, lane);
                    flushed = xc_ring_append<WIN>(sh.win, out, out + outPos, 0u, b.lit_regen, outPos, flushed, false, lane);
                } else {
                    flushed = xc_ring_append<WIN>(sh.win, out, litRle ? nullptr : lits, litRle ? (uint32_t)lits[0] : 0u, b.lit_regen, outPos, flushed,
                                                  true, lane);
                }
                outPos += b.lit_regen;
                XC_STAT(14, XC_CLOCK() - xc_ta);
                XC_STAT(15, 1);
                continue;
            }
            flushed = xc_flush_bytes<WIN>(sh.win, out, flushed, outPos, lane);
            if (!b.pad[0]) {
                if (litRle) xb_bulk_fill(out + outPos, lits[0], b.lit_regen, lane);
                else xb_bulk_copy(out + outPos, lits, b.lit_regen, lane);
            }
            outPos += b.lit_regen;
            if (!BM) xc_reload_window<WIN>(sh.win, out, outPos, lane);
            flushed = confirmed = outPos;
            XC_STAT(14, XC_CLOCK() - xc_ta);
            XC_STAT(15, 1);
            continue;
        }
        if (litRle) {  // RLE literals (literals.go:390-396): every literal of every stretch is this byte
            const uint64_t v = lits[0] * 0x0101010101010101ull;
            *(uint64_t *)&sh.lit[8 * lane] = v;
        }

        // ---- tiles of 64 sequences; the literals after the last sequence (sequence_execution.go:55-59) ride along as one
        // more sequence without a match.  A software pipeline over stretches, as in k_exec_b: the loads of the next stretch
        // (staged matches, literals, the records of the tile after) are issued before the passes of the current one.
        const uint32_t rest = b.lit_regen - litTotal;
        const uint32_t nps = b.n_seq + (rest ? 1u : 0u);
        const uint32_t ntiles = (nps + 63) >> 6;
        const uint64_t *brec = recs + b.rec_off;

        struct Tile {
            uint32_t LL, ML, lstart, mstart, lsrc;
            int off;
            uint32_t start, E, lits, litRun;  // wave-uniform: the tile's first byte, the byte after its last, its literals, literals before it
        };
        struct Plan {
            uint32_t P, sEnd, la, lb;
            uint64_t stg;      // lanes whose match goes through the stage
            uint64_t stg2;     // ... those of them longer than 16 bytes
            uint64_t farwin;   // lanes with a window match that is neither staged nor within the ring's reach
            U128U sv, sv2;     // the staged source bytes
            uint64_t lv;       // the stretch's literals, 8 per lane
            uint64_t rec;      // the records of the tile AFTER the stretch's tile
        };
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        auto load_recs = [&](uint32_t tile) -> uint64_t {
            const uint32_t si = min(tile * 64 + (uint32_t)lane, b.n_seq - 1);  // (clamped: no exec mask around the load; n_seq > 0)
            return ld64_once<kXcNtRec>(brec + si);
        };
        // tile t from its records; false: an offset beyond the produced data (ringbuffer.go:206-214)
        auto load_tile = [&](Tile &T, uint64_t rec, uint32_t t, uint32_t tileStart, uint32_t litRun) -> bool {
            uint32_t LL = (uint32_t)rec & kRecLlMask;
            uint32_t ML = (uint32_t)(rec >> kRecMlShift) & kRecMlMask;
            uint32_t offf = (uint32_t)(rec >> kRecOffShift) & kRecOffMask;
            if (t + 1 == ntiles) {  // the block's last tile: lanes beyond the sequences; the trailing literals
                const uint32_t si = t * 64 + (uint32_t)lane;
                if (si >= b.n_seq) {
                    LL = si == b.n_seq ? rest : 0u;
                    ML = 0;
                    offf = 0;
                }
            }
            int off = (int)offf;
            if (wave_any((offf & kRecOffSymbolic) != 0)) {
                if (offf & kRecOffSymbolic) {
                    const uint32_t u = offf & (kRecOffSymbolic - 1);
                    off = sel3(u & 3, H0, H1, H2) - (int)(u >> 2);
                }
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
            XC_STAT(0, 1);
            XC_STAT(7, __popcll(wave_ballot(ML > 0)));
            return !wave_any(ML > 0 && (off <= 0 || (uint32_t)off > T.mstart));
        };
        // the stretch of tile T that starts at P with the literal cursor at la: its extent, and its loads on their way
        auto plan_stretch = [&](const Tile &T, uint32_t P, uint32_t la, Plan &N) {
            uint32_t sEnd = T.E, lb = T.litRun + T.lits;
            if (T.E - P > kXcStretch || lb - la > kXcLit) {
                // the tile does not fit one stretch from here: the stretch ends after 1024 bytes, or where its 512th literal does
                sEnd = min(T.E, P + kXcStretch);
                {
                    // literal cursor at the stretch's end: the last sequence that starts at or before it (the lanes' starts ascend)
                    const int k = 63 - __builtin_clzll(wave_ballot(T.lstart <= sEnd));
                    const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)T.lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)T.lsrc, k);
                    const uint32_t kn = (uint32_t)__builtin_amdgcn_readlane((int)T.LL, k);
                    lb = ks + min(kn, sEnd - kl);
                }
                if (lb - la > kXcLit) {
                    const int k = 63 - __builtin_clzll(wave_ballot(T.lsrc <= la + kXcLit));  // the run that holds literal la + 512
                    const uint32_t kl = (uint32_t)__builtin_amdgcn_readlane((int)T.lstart, k), ks = (uint32_t)__builtin_amdgcn_readlane((int)T.lsrc, k);
                    sEnd = kl + (la + kXcLit - ks);
                    lb = la + kXcLit;
                }
            }
            N.P = uni(P);
            N.sEnd = uni(sEnd);
            N.la = uni(la);
            N.lb = uni(lb);
            // matches whose source is final in the slab: 32 source bytes into the stage, all of the stretch's loads in flight
            // together (a match not farther back than the ring reaches is served by the ring whatever pass it falls into)
            // (a match that CONTINUES from the stretch before is classified -- and staged -- again: every lane stores its stage slot)
            const bool mHere = T.ML > 0 && T.mstart < N.sEnd && T.mstart + T.ML > N.P;
            const uint32_t q0 = T.mstart - (uint32_t)T.off;
            const bool farm = mHere && T.off > kXcNear;
            // (block mode: a source that straddles the segment's start is left to the pass)
            bool stg = farm && T.ML <= kXcStageMl && q0 + T.ML <= confirmed && (!BM || q0 >= S || q0 + T.ML <= S);
            // (slots in lane order: the staged matches are the lanes whose slots end inside the stage -- a prefix of the candidates)
            if (wave_incl_scan_dpp(stg ? (T.ML + 7u) >> 3 : 0u) > kXcStageSlots) stg = false;
            const uint8_t *const rb = BM && q0 < S ? pat : (const uint8_t *)out;
            N.stg = wave_ballot(stg);
            N.stg2 = wave_ballot(stg && T.ML > 16);
            N.farwin = wave_ballot(farm && !stg);
            // (under exec masks: the address unit spends time on every ACTIVE lane of a scattered load, whatever it reads -- with
            // all 64 lanes loading, idle ones from one hot line, TA_TA_BUSY went from 46 % to 81 % of the kernel's time)
            N.sv = N.sv2 = U128U{0, 0, 0, 0};
#ifndef MZD_ABL_XC_NOSTAGE
            if (stg) N.sv = ld128u_once<kXcNtStage>(rb + q0);
            if (stg && T.ML > 16) N.sv2 = ld128u_once<kXcNtStage>(rb + q0 + 16);
#endif
            // the stretch's literals: [la, lb) of the block's literals (lb <= lit_regen)
            N.lv = 0;
            const uint32_t li = N.la + 8u * (uint32_t)lane;
            if (!litRle && li < N.lb) N.lv = ld64u_once<kXcNtLit>(lits + li);
        };

        Tile T;
        uint32_t t = 0;
        // One step of the pipeline: finish the setup of stretch C (its loads were issued a step ago), plan stretch N and issue
        // its loads, run C's passes.  -> false when C was the block's last stretch (or the block failed).
        auto step = [&](Plan &C, Plan &N) -> bool {
            // ---- the current stretch: everything a pass looks up goes to LDS
            const unsigned long long xc_t2 = XC_CLOCK();
            (void)xc_t2;
            const uint32_t P0 = C.P, sEnd = C.sEnd, sLen = sEnd - P0, la = C.la;
            const bool stg = (C.stg >> lane) & 1;
            const uint32_t n8 = stg ? (T.ML + 7u) >> 3 : 0u;
            const uint32_t stgA = (uint32_t)offsetof(XcLdsW, stage) + 8u * (wave_incl_scan_dpp(n8) - n8);  // where this lane's staged bytes go
            // the heads of the stretch: table entries and bitmap bits.  (A lambda instantiated on both paths below: a `bool` that
            // merges from two branches travels through a VGPR as 0 / 1 and is compared again)
            auto write_heads = [&](bool litIn, bool mIn) {
                // table index of the sequence's heads: 1 + the heads of the lanes below (+ its own literal head)
                const uint64_t litMask = wave_ballot(litIn), mMask = wave_ballot(mIn);
                const uint32_t hb = __builtin_amdgcn_mbcnt_hi((uint32_t)(litMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)litMask, 1u)) +
                                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mMask, 0u));
                // the entries: literal byte p lies at lit + (lsrc - la) + (p - lstart); a staged match byte at stage + 32 lane +
                // (p - mstart); a window match byte at ring position p - off
                const uint32_t eL = ((T.lsrc - la - T.lstart) & (kXcWin - 1)) | kXcWin;
                uint32_t eM = (uint32_t)(-T.off) & (kXcWin - 1);
                if (T.off > kXcNear) eM = kXcFarMark | (uint32_t)lane;
                if (stg) eM = ((stgA - T.mstart) & (kXcWin - 1)) | kXcWin;
                const uint32_t hbm = hb + (litIn ? 1u : 0u);
                xc_lds_write_b32(litIn ? tabA + 4u * hb : kXcOor, eL);
                xc_lds_write_b32(mIn ? tabA + 4u * hbm : kXcOor, eM);
                const uint32_t xl = T.lstart - P0, xm = T.mstart - P0;
                xc_lds_or_b32(litIn ? bitsA + ((xl >> 3) & 0x7Cu) : kXcOor, 1u << (xl & 31));
                xc_lds_or_b32(mIn ? bitsA + ((xm >> 3) & 0x7Cu) : kXcOor, 1u << (xm & 31));
            };
            if (P0 == T.start && sEnd == T.E) {
                // the whole tile is this stretch: every run starts in it, none continues from a stretch before
                write_heads(T.LL > 0, T.ML > 0);
            } else {
                const bool contL = T.lstart < P0 && P0 < T.mstart;     // a run that continues from the stretch before (one lane at most)
                const bool contM = T.mstart < P0 && P0 < T.mstart + T.ML;
                if (contL) sh.table[0] = ((T.lsrc - la - T.lstart) & (kXcWin - 1)) | kXcWin;
                if (contM) {
                    uint32_t e = (uint32_t)(-T.off) & (kXcWin - 1);
                    if (stg) e = ((stgA - T.mstart) & (kXcWin - 1)) | kXcWin;
                    else
                    if (T.off > kXcNear) e = kXcFarMark | (uint32_t)lane;
                    sh.table[0] = e;
                }
                // the sequence's literal run / match starts in this stretch
                write_heads(T.LL > 0 && T.lstart - P0 < sLen, T.ML > 0 && T.mstart - P0 < sLen);
                XC_STAT(13, 1);
            }
#ifndef MZD_ABL_XC_NOWAIT  /* ablations: timing experiments only, wrong results */
            xb_wait_vm();  // the staged bytes, the literals and the next tile's records are here; so is every window unit issued before
#endif
            confirmed = uni(flushed);
            // (every lane stores its 16 bytes: what a lane without a staged match leaves in its slot is never looked at)
            {
                uint8_t *const sp = (uint8_t *)&sh + stgA;
                if (stg) *(uint2 *)sp = make_uint2(C.sv.x, C.sv.y);
                if (stg && T.ML > 8) *(uint2 *)(sp + 8) = make_uint2(C.sv.z, C.sv.w);
                if (C.stg2) {
                    if (stg && T.ML > 16) *(uint2 *)(sp + 16) = make_uint2(C.sv2.x, C.sv2.y);
                    if (stg && T.ML > 24) *(uint2 *)(sp + 24) = make_uint2(C.sv2.z, C.sv2.w);
                }
            }
            if (!litRle) *(uint64_t *)&sh.lit[8 * lane] = C.lv;
            const bool farwin = C.farwin != 0;
            XC_STAT(1, 1);
            XC_STAT(6, __popcll(C.stg));
            XC_STAT(8, XC_CLOCK() - xc_t2);

            // ---- the NEXT stretch (of this tile, or the first of the next tile): planned, its loads issued
            const unsigned long long xc_t4 = XC_CLOCK();
            (void)xc_t4;
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
#ifdef MZD_ABL_XC_NOFLUSH
                while (P0 - fl >= kXcFlush) fl += kXcFlush;
#else
                while (P0 - fl >= kXcFlush) fl = uni(xc_flush_step<WIN>(sh.win, out, fl, lane));
#endif
                flushed = fl;
            }
            XC_STAT(9, XC_CLOCK() - xc_t4);

            // ---- the passes: 128 bytes each, lane j makes bytes P + j and P + 64 + j
            const unsigned long long xc_t3 = XC_CLOCK();
            (void)xc_t3;
            uint32_t sbase = 1;
            const uint4 *hbits = (const uint4 *)sh.bits;
            uint32_t pa = P0 + (uint32_t)lane;
            uint32_t k = 0;
#ifdef MZD_ABL_XC_NOPASS
            for (uint32_t P = sEnd; P < sEnd; k++, P += kXcPass, pa += kXcPass) {
#else
            for (uint32_t P = P0; P < sEnd; k++, P += kXcPass, pa += kXcPass) {
#endif
                const uint4 H = hbits[k];
                const uint32_t pb = pa + 64u;
                const uint32_t wa = pa & vwmask, wb = pb & vwmask;
                uint32_t ea, eb, aa, ab, va, vb, na, nb, rounds;
                XC_STAT(2, 1);
                if (!farwin) {
                    // The pass, hand-scheduled (the compiler's version of the same statement: 66 instructions, this: 46).  Owners from
                    // v_mbcnt (heads below the lane + the lane's own head bit), the two table entries, the two source bytes, the
                    // stores; then the fixed point: read the sources again, done when no lane sees a change (at most four rounds
                    // here; what still moves then is a long chain inside the pass: the resolver below).  gfx950 wants two wait
                    // states between a VALU write of an SGPR / VCC and a VALU read of it and one between a VALU write of a VGPR and
                    // a v_readlane of it: the order below keeps them (nothing inside an asm statement is padded by the compiler).
                    uint32_t ca, cb, t, u, sa;  // (t / u also hold the table addresses, ca / cb the bytes of the confirming reads)
                    unsigned long long m;
                    asm volatile(
                        "v_mbcnt_lo_u32_b32 %[ca], %[h0], %[sbase]\n\t"
                        "v_and_b32 %[t], %[h0], %[lblo]\n\t"
                        "v_mbcnt_lo_u32_b32 %[cb], %[h2], 0\n\t"
                        "v_and_b32 %[u], %[h2], %[lblo]\n\t"
                        "v_mbcnt_hi_u32_b32 %[ca], %[h1], %[ca]\n\t"
                        "v_and_or_b32 %[t], %[h1], %[lbhi], %[t]\n\t"
                        "v_mbcnt_hi_u32_b32 %[cb], %[h3], %[cb]\n\t"
                        "v_and_or_b32 %[u], %[h3], %[lbhi], %[u]\n\t"
                        "v_cmp_eq_u32 vcc, 0, %[t]\n\t"
                        "v_cmp_eq_u32_e64 %[m], 0, %[u]\n\t"
                        "s_nop 0\n\t"
                        "v_subbrev_co_u32 %[ca], vcc, 0, %[ca], vcc\n\t"          /* owner of byte a */
                        "v_subbrev_co_u32_e64 %[cb], %[m], 0, %[cb], %[m]\n\t"    /* heads of half b at or below the lane, - 1 */
                        "v_lshl_add_u32 %[t], %[ca], 2, %[tab]\n\t"
                        "v_readlane_b32 %[sa], %[ca], 63\n\t"
                        "ds_read_b32 %[ea], %[t]\n\t"
                        "s_nop 0\n\t"
                        "v_add3_u32 %[cb], %[cb], %[sa], 1\n\t"                  /* owner of byte b */
                        "v_lshl_add_u32 %[u], %[cb], 2, %[tab]\n\t"
                        "v_readlane_b32 %[sbase], %[cb], 63\n\t"
                        "ds_read_b32 %[eb], %[u]\n\t"
                        "s_add_u32 %[sbase], %[sbase], 1\n\t"
                        "s_waitcnt lgkmcnt(1)\n\t"
                        "v_add_u32 %[t], %[ea], %[pa]\n\t"
                        "v_bfi_b32 %[aa], %[mask], %[t], %[ea]\n\t"
                        "ds_read_u8 %[va], %[aa]\n\t"
                        "s_waitcnt lgkmcnt(1)\n\t"
                        "v_add_u32 %[u], %[eb], %[pb]\n\t"
                        "v_bfi_b32 %[ab], %[mask], %[u], %[eb]\n\t"
                        "ds_read_u8 %[vb], %[ab]\n\t"
                        /* while the bytes are on their way: does a byte of this pass derive from the pass itself?  A ring entry is
                           4096 - offset: its sum with the byte's index in the pass carries into bit 12 exactly when offset <= index */
                        "v_add_u32 %[t], %[ea], %[lane]\n\t"
                        "v_add3_u32 %[u], %[eb], %[lane], 64\n\t"
                        "v_xor_b32 %[t], %[t], %[ea]\n\t"
                        "v_xor_b32 %[u], %[u], %[eb]\n\t"
                        "v_or_b32 %[t], %[t], %[u]\n\t"
                        "v_and_b32 %[t], %[winbit], %[t]\n\t"
                        "v_cmp_ne_u32 vcc, 0, %[t]\n\t"
                        "s_mov_b32 %[rounds], 0\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b8 %[wa], %[va]\n\t"   /* (beyond the stretch's end: bytes the next stretch overwrites before anything reads them) */
                        "ds_write_b8 %[wb], %[vb]\n\t"
                        "s_cbranch_vccz L_xc_done_%=\n"
                        /* to the fixed point: read the sources again; done when no lane sees a change */
                        "L_xc_round_%=:\n\t"
                        "ds_read_u8 %[ca], %[aa]\n\t"
                        "ds_read_u8 %[cb], %[ab]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_cmp_ne_u32 vcc, %[ca], %[va]\n\t"
                        "v_cmp_ne_u32_e64 %[m], %[cb], %[vb]\n\t"
                        "s_or_b64 vcc, vcc, %[m]\n\t"
                        "s_cbranch_vccz L_xc_done_%=\n\t"
                        "v_mov_b32 %[va], %[ca]\n\t"
                        "v_mov_b32 %[vb], %[cb]\n\t"
                        "ds_write_b8 %[wa], %[ca]\n\t"
                        "ds_write_b8 %[wb], %[cb]\n\t"
                        "s_add_u32 %[rounds], %[rounds], 1\n\t"
                        "s_cmp_lt_u32 %[rounds], 4\n\t"
                        "s_cbranch_scc1 L_xc_round_%=\n"
                        "L_xc_done_%=:\n\t"
                        : [ca] "=&v"(ca), [cb] "=&v"(cb), [t] "=&v"(t), [u] "=&v"(u), [ea] "=&v"(ea),
                          [eb] "=&v"(eb), [aa] "=&v"(aa), [ab] "=&v"(ab), [va] "=&v"(va), [vb] "=&v"(vb),
                          [sa] "=&s"(sa), [m] "=&s"(m), [rounds] "=&s"(rounds), [sbase] "+s"(sbase)
                        : [h0] "v"(H.x), [h1] "v"(H.y), [h2] "v"(H.z), [h3] "v"(H.w), [lblo] "v"(lblo), [lbhi] "v"(lbhi), [tab] "s"(tabA),
                          [pa] "v"(pa), [pb] "v"(pb), [mask] "v"(vwmask), [wa] "v"(wa), [wb] "v"(wb), [lane] "v"(vlane), [winbit] "s"(kXcWin)
                        : "memory", "vcc", "scc");
                    XC_STAT(3, rounds);
                    if (rounds >= 4) {
                        // still moving after five rounds: chains inside the pass (a run feeding itself at a short period).  What a
                        // byte copies: the element of the pass its window source falls on (its ring slot is one of the pass's)
                        XC_STAT(4, 1);
                        const uint32_t da = (wa - aa) & vwmask, db = (wb - ab) & vwmask;  // distance back to the source, for ring sources
                        // (distance 0: a byte that is its own source -- entry 0 -- is final as it stands)
                        const int ra = ea < kXcWin && da != 0 && da <= (uint32_t)lane ? (int)((uint32_t)lane - da) : -1;
                        const int rb = eb < kXcWin && db != 0 && db <= 64u + (uint32_t)lane ? (int)(64u + (uint32_t)lane - db) : -1;
                        // bytes whose source lies before the pass are final as read; the others start from their pointers
                        const uint2 r = xc_resolve_in_pass(va, vb, ra, rb, (uint32_t)lane);
                        sh.win[wa] = (uint8_t)r.x;
                        sh.win[wb] = (uint8_t)r.y;
                    }
                    continue;
                }
                // ---- a stretch with window matches that are neither staged nor within the ring's reach (long far matches; a
                // source not yet confirmed when the stretch was planned): the same pass in C++, the far bytes read from the slab
                {
                    const uint32_t ca = __builtin_amdgcn_mbcnt_hi(H.y, __builtin_amdgcn_mbcnt_lo(H.x, sbase));
                    const uint32_t owna = ca - (((H.x & lblo) | (H.y & lbhi)) ? 0u : 1u);
                    const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)owna, 63) + 1u;
                    const uint32_t cb = __builtin_amdgcn_mbcnt_hi(H.w, __builtin_amdgcn_mbcnt_lo(H.z, sb));
                    const uint32_t ownb = cb - (((H.z & lblo) | (H.w & lbhi)) ? 0u : 1u);
                    sbase = (uint32_t)__builtin_amdgcn_readlane((int)ownb, 63) + 1u;
                    ea = sh.table[owna];
                    eb = sh.table[ownb];
                    aa = xc_bfi(vwmask, pa + ea, ea);
                    ab = xc_bfi(vwmask, pb + eb, eb);
                    va = lds[aa];
                    vb = lds[ab];
                    const bool fa = (ea & kXcFarMark) != 0 && pa < sEnd, fb = (eb & kXcFarMark) != 0 && pb < sEnd;
                    if (wave_any(fa || fb)) {
                        XC_STAT(5, 1);
                        xb_wait_vm();  // every window unit issued so far has arrived in the slab
                        const uint32_t offa = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ea & 63) << 2), T.off);
                        const uint32_t offb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((eb & 63) << 2), T.off);
                        // (a byte this far back has left the ring for the slab: off > kXcNear and the units up to P0 - 511 are out)
                        if (fa) va = (BM && pa - offa < S ? pat : (const uint8_t *)out)[pa - offa];
                        if (fb) vb = (BM && pb - offb < S ? pat : (const uint8_t *)out)[pb - offb];
                    }
                    sh.win[wa] = (uint8_t)va;
                    sh.win[wb] = (uint8_t)vb;
                    // the other bytes of the pass to their fixed point (the far ones keep what memory gave them)
                    for (uint32_t round = 0; round < 130; round++) {  // (a chain inside a pass is at most 128 long)
                        na = fa ? va : lds[aa];
                        nb = fb ? vb : lds[ab];
                        if (!wave_any(na != va || nb != vb)) break;
                        va = na;
                        vb = nb;
                        sh.win[wa] = (uint8_t)va;
                        sh.win[wb] = (uint8_t)vb;
                        XC_STAT(3, 1);
                    }
                }
            }
            xc_lds_write_b32(clrA, 0u);  // every head of the stretch has been used: the bitmap is clear for the next one
            xc_lds_write_b32(clrA + 4u, 0u);  // (kXcOor + 4 is as far out of range as kXcOor)
            XC_STAT(10, XC_CLOCK() - xc_t3);
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
    if (error == MZD_OK) flushed = xc_flush_bytes<WIN>(sh.win, out, flushed, outPos, lane);
    if (BM) {
        // an offset beyond the produced data (the one defect the scan cannot see): the frame ends at its first such block
        if (lane == 0 && error != MZD_OK && bk.pass == 0) atomicMin(&bk.bframes[fidx].first_bad, bi);
        return;
    }
#ifdef MZD_XC_STATS
    xcst[11] = XC_CLOCK() - xc_t0;
    xcst[12] = 1;
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_xc_stats[i], xcst[i]);
#endif
    if (lane == 0) {
        int e = error;
        if (e == MZD_OK && fr.content_size != MZD_UNKNOWN_SIZE && outPos != fr.content_size) e = MZD_ERR_DST_FULL;
        frame_status[fidx] = e;
        frame_out_len[fidx] = outPos;
        if (frame_hist) {  // (batches with a chunk of a frame in them: what the next chunk starts with)
            frame_hist[3 * fidx] = H0;
            frame_hist[3 * fidx + 1] = H1;
            frame_hist[3 * fidx + 2] = H2;
        }
    }
}

// ---- batches of nothing but Raw / RLE blocks (framedecompressor.go:211-215,229-241; BASELINE configs[1]): the pass IS a copy.
// k_exec gives such a frame one workgroup that walks its blocks -- 4 096 workgroups of two wavefronts for the config's 512 MiB,
// 0.61 of the copy ceiling measured beside it.  Here the grid is (frame, 16 KiB chunk): workgroup (f, c) does chunk c of every
// block of frame f, 256 lanes x 16 bytes per instruction, four loads in flight per lane before the stores; chunk 0 reports the
// frame.  A block is at most 128 KiB: eight chunks.
constexpr uint32_t kCopyChunk = 16384, kCopyChunksPerBlock = kBlockMax / kCopyChunk;
#ifndef MZD_COPY_NT
#define MZD_COPY_NT 1
#endif
// the copy's loads and stores with the nt bit (both streams are touched once): 0.169 -> 0.164 ms for BASELINE config 2, 0.80 -> 0.85 of
// the copy ceiling measured beside it (profiles/r5_copy_nt.txt)
constexpr bool kCopyNt = MZD_COPY_NT != 0;
__global__ __launch_bounds__(256) void k_copy_blocks(const uint8_t *__restrict__ in, uint8_t *out_blob, const DFrame *__restrict__ frames,
                                                     const DBlock *__restrict__ blocks, int32_t *frame_status, uint64_t *frame_out_len,
                                                     const uint32_t *__restrict__ order, uint32_t first, uint32_t n_frames)
{
    const uint32_t tid = threadIdx.x;
    // (launched with a workgroup per chunk; the loop is there for a smaller grid)
    for (uint32_t w = blockIdx.x; w < n_frames * kCopyChunksPerBlock; w += gridDim.x) {
    const uint32_t fi = w / kCopyChunksPerBlock, c = w % kCopyChunksPerBlock;
    const uint32_t fidx = order ? order[first + fi] : first + fi;
    const DFrame fr = frames[fidx];
    uint8_t *out = out_blob + fr.out_offset;
    int error = fr.plan_status;
    uint64_t outPos = fr.continues ? fr.start : 0;
    for (uint32_t bi = 0; bi < fr.n_blocks && error == MZD_OK; bi++) {
        const DBlock b = blocks[fr.first_block + bi];
        if (b.type == MZD_BLOCK_COMPRESSED) {  // (not in such a batch; a frame that has one is not this kernel's)
            error = MZD_ERR_UNSUPPORTED;
            break;
        }
        if (outPos + b.size > fr.out_capacity) {
            error = MZD_ERR_DST_FULL;
            break;
        }
        const uint32_t lo = c * kCopyChunk;
        if (lo < b.size) {
            const uint32_t n = min(kCopyChunk, b.size - lo);
            uint8_t *dst = out + outPos + lo;
            // (slabs are 256-byte aligned and a frame's blocks before the last are whole multiples of nothing in particular: the
            // destination may be misaligned, so 16-byte accesses go through the packed type -- the hardware takes them)
            const uint32_t n16 = n >> 4;
            if (b.type == MZD_BLOCK_RAW) {
                const uint8_t *src = in + b.src_off + lo;
                uint32_t i = tid;
                for (; i + 768 < n16; i += 1024) {
                    const U128U v0 = ld128u_once<kCopyNt>(src + 16 * i), v1 = ld128u_once<kCopyNt>(src + 16 * (i + 256));
                    const U128U v2 = ld128u_once<kCopyNt>(src + 16 * (i + 512)), v3 = ld128u_once<kCopyNt>(src + 16 * (i + 768));
                    st128u_once<kCopyNt>(dst + 16 * i, v0);
                    st128u_once<kCopyNt>(dst + 16 * (i + 256), v1);
                    st128u_once<kCopyNt>(dst + 16 * (i + 512), v2);
                    st128u_once<kCopyNt>(dst + 16 * (i + 768), v3);
                }
                for (; i < n16; i += 256) *(U128U *)(dst + 16 * i) = *(const U128U *)(src + 16 * i);
                for (uint32_t j = (n16 << 4) + tid; j < n; j += 256) dst[j] = src[j];
            } else {
                const uint32_t v = in[b.src_off] * 0x01010101u;
                const U128U f{v, v, v, v};
                for (uint32_t i = tid; i < n16; i += 256) st128u_once<kCopyNt>(dst + 16 * i, f);
                for (uint32_t j = (n16 << 4) + tid; j < n; j += 256) dst[j] = (uint8_t)v;
            }
        }
        outPos += b.size;
    }
    if (c == 0 && tid == 0) {
        int e = error;
        if (e == MZD_OK && fr.content_size != MZD_UNKNOWN_SIZE && outPos != fr.content_size) e = MZD_ERR_DST_FULL;
        frame_status[fidx] = e;
        frame_out_len[fidx] = outPos;
    }
    }
}

}  // namespace mzd
