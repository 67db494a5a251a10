// mzd_exec_b.hip -- k_exec_b: sequence execution with ONE WAVEFRONT per frame and ONE LANE PER OUTPUT BYTE.
//
// Replaces decompression/sequence_execution.go:14-63 (ExecuteSequences), ringbuffer.go:102-277 (Push / Repeat /
// RepeatBeforeIndex) and the Raw / RLE block arms framedecompressor.go:211-215,229-241, like k_exec.  Same inputs
// (the 8-byte sequence records of the entropy stage, the regenerated literals) and the same statuses; another shape:
//
//   k_exec    a lane per SEQUENCE, 8 KiB of the block in LDS, a per-byte validity bitmap and a dataflow loop over
//             the pending matches.  Its LDS pipe is what fills (every copy is a byte-misaligned LDS access: a cycle
//             per active lane), and its 8.5 KiB per frame keep it from sharing a CU with the sequence stage.
//   k_exec_b  the output is produced strictly IN ORDER, 64 bytes per pass, lane j making byte P + j:
//               1. which sequence owns the byte: every sequence marks the first byte of its literal run and of its
//                  match in a small LDS "head map" (one byte per output position, written once per 64-sequence
//                  tile); a pass reads its 64 map bytes and a wave-wide max-scan (DPP) hands every lane the last
//                  head at or below it;
//               2. where the byte comes from: two ds_bpermute fetch the owner's displacement (match: -offset,
//                  literal: literal cursor - output position), source = position + displacement;
//               3. the byte itself: a literal (from a 512-byte LDS ring of the block's literals), a NEAR match byte
//                  (the last 2 KiB of output live in an LDS ring), a FAR match byte (global memory: everything
//                  older left for the frame's slab in 512-byte units), or a byte this very pass produces
//                  (offset < 64: resolved between the lanes by pointer jumping, at most six rounds);
//               4. one aligned byte store into the window ring.
//             No byte-misaligned LDS access, no bitmap, no atomics, no barrier: all LDS instructions are aligned
//             whole-wavefront ones (~16 LDS-pipe cycles per 64 bytes), and a frame needs 3 KiB of LDS, so a CU holds
//             32 of them -- or a few beside a sequence-stage workgroup that owns the rest of the LDS.
//
// Hazards are ordered by construction: a wavefront's LDS operations execute in order (a pass's reads precede its
// stores), global stores of the window units are waited for (vmcnt) before the NEXT unit is issued, and a far read
// can only touch units at least two units old.
#pragma once

namespace mzd {

constexpr uint32_t kXbWin = 2048;      // near window (ring, position & (kXbWin - 1))
constexpr uint32_t kXbMap = 512;       // head map (ring, position & (kXbMap - 1)); heads are written at most this far ahead
constexpr uint32_t kXbLit = 512;       // literal ring: two units
constexpr uint32_t kXbLitUnit = 256;   // 64 lanes x 4 bytes
constexpr uint32_t kXbFlush = 512;     // 64 lanes x 8 bytes leave for global memory at a time
static_assert(kXbWin >= 2 * kXbFlush + 64 + 64 + 256, "a far read must never meet the unit whose stores are still in flight");

constexpr uint32_t kXbStage = 1024;    // far matches of the current tile: 16 source bytes per sequence lane
#ifdef MZD_XB_STATS
// tools/xb_stats.py: 0 tiles, 1 fast tiles, 2 fast passes, 3 general passes, 4 passes with a byte made by the pass itself,
// 5 fast passes that went to memory themselves, 6 staged matches, 7 matches, 8 cycles in fast-tile setup, 9 cycles in fast
// passes, 10 cycles in general tiles, 11 cycles total, 12 frames
__device__ unsigned long long g_xb_stats[16];
#define XB_STAT(i, n) (xbst[i] += (unsigned long long)(n))
#define XB_CLOCK() __builtin_readcyclecounter()
#else
#define XB_STAT(i, n) do { } while (0)
#define XB_CLOCK() 0ull
#endif
struct XbLds {
    uint8_t win[kXbWin];      // 0x000
    uint8_t lit[kXbLit];      // 0x800
    uint8_t map[kXbMap];      // 0xa00
    uint8_t stage[kXbStage];  // 0xc00
};
static_assert(offsetof(XbLds, lit) == 0x800 && offsetof(XbLds, map) == 0xa00 && offsetof(XbLds, stage) == 0xc00 && kXbWin == 0x800 &&
              kXbLit == 0x200 && kXbMap == 0x200 && kXbStage == 0x400, "xb_pass addresses the rings with immediate masks and offsets");
// head map entry: (sequence lane << 2) | code; 0 = no head at this byte
enum { kXbLitHead = 1, kXbMatchHead = 2, kXbStagedHead = 3 };

// wave64 inclusive max-scan on the DPP path (values are unsigned, 0 = nothing): row_shr 1/2/4/8, row_bcast 15/31
__device__ __forceinline__ uint32_t wave_incl_max_dpp(uint32_t v)
{
    v = max(v, dpp_shr<0x111, 0xf, 0xf>(v));
    v = max(v, dpp_shr<0x112, 0xf, 0xf>(v));
    v = max(v, dpp_shr<0x114, 0xf, 0xe>(v));
    v = max(v, dpp_shr<0x118, 0xf, 0xc>(v));
    v = max(v, dpp_shr<0x142, 0xa, 0xf>(v));
    v = max(v, dpp_shr<0x143, 0xc, 0xf>(v));
    return v;
}

__device__ __forceinline__ void xb_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// bytes [flushed, upto) of the frame leave the window for the slab, byte by byte (block ends, unaligned remainders)
__device__ __forceinline__ void xb_flush_bytes(XbLds &sh, uint8_t *out, uint32_t &flushed, uint32_t upto, int lane)
{
    for (uint32_t x = flushed + (uint32_t)lane; x < upto; x += 64) out[x] = sh.win[x & (kXbWin - 1)];
    flushed = upto;
}

// one step of the steady-state flush: the 512-byte unit at `flushed` (or the bytes up to the next unit boundary)
__device__ __forceinline__ void xb_flush_step(XbLds &sh, uint8_t *out, uint32_t &flushed, int lane, bool wait = true)
{
    if (wait) xb_wait_vm();  // the previous unit has arrived: every unit but the one issued below is final in memory
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
// bytes of the frame back from memory so that the next block's near matches find them
__device__ __forceinline__ void xb_reload_window(XbLds &sh, const uint8_t *out, uint32_t outPos, uint32_t &validFrom,
                                                 uint32_t &flushed, int lane)
{
    xb_wait_vm();  // the bulk stores are in memory (same CU: visible to the loads below)
    const uint32_t lo = outPos > kXbWin ? outPos - kXbWin : 0u;
    const uint32_t lo4 = (lo + 3u) & ~3u;
    const uint32_t hi4 = outPos & ~3u;
    for (uint32_t x = lo4 + 4u * (uint32_t)lane; x < hi4; x += 256) *(uint32_t *)&sh.win[x & (kXbWin - 1)] = ((const U32U *)(out + x))->v;
    for (uint32_t x = max(lo4, hi4) + (uint32_t)lane; x < outPos; x += 64) sh.win[x & (kXbWin - 1)] = out[x];
    validFrom = min(lo4, outPos);
    flushed = outPos;
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

// one 256-byte unit of the block's literals, a dword per lane (zero beyond the regenerated size)
__device__ __forceinline__ uint32_t xb_lit_unit(const uint8_t *lits, uint32_t unit_off, uint32_t lit_regen, int lane)
{
    const uint32_t i = unit_off + 4u * (uint32_t)lane;
    return i < lit_regen ? ((const U32U *)(lits + i))->v : 0u;
}


// lanes whose source byte is produced by this very pass (offset <= lane): pointer jumping between the lanes -- a lane either
// takes its source lane's byte or, while that one is still waiting itself, its source lane.  At most six rounds.
__device__ __forceinline__ uint32_t xb_resolve_in_pass(uint32_t val, uint32_t srcl, bool dep)
{
    uint32_t done = dep ? 0u : 1u;
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

// One pass of the steady state, hand-written: 64 output bytes [P, P + 64) (the lanes of `act`; all of them except in a
// tile's last pass).  Preconditions (the caller's fast-tile test): the XbLds block sits at LDS address 0; the heads of
// every sequence that starts inside the pass are in the map; the window ring holds [P - kXbWin, P); the literal ring
// holds every literal the pass consumes; everything below P - kXbWin has been issued to the slab.
// A byte's source by the code of the head that owns it:
//   1 literal        literal ring,  index p + dlv of the owner
//   2 match          window ring, position p + offx (= -offset) -- or, older than the window, the slab (rare: the tile
//                    setup stages far matches; what is left are long ones and sources not yet confirmed in memory)
//   3 staged match   stage, p + offx (= 16 * owner lane - match start)
// The compiler's version of the same statement (the general loop in k_exec_b) spends ~55 scalar and ~12 branch
// instructions per pass on exec-mask bookkeeping; this one has 9 scalar ones and one branch.
// gfx950 wait states kept by hand (the hazard recogniser does not look into inline asm): 2 between a VALU write of a
// VGPR and a DPP read, 1 before a v_readlane of it, 2 between a VALU write of an SGPR / VCC and a VALU read of it.
//   r (out): source position relative to P for window-ring match lanes (>= 0: produced by this pass), -1 for all others
__device__ __forceinline__ void xb_pass(uint32_t P, uint64_t act, uint32_t lane, int offx, int dlv, uint32_t &carry,
                                        const uint8_t *out, uint32_t litMask, uint32_t &val, int &r, uint64_t &dep)
{
    uint32_t p, m, e, t, g, x, y, q, a, b, c, l, f;
    uint64_t far, sn;
    asm volatile(
        "v_add_u32 %[p], %[P], %[lane]\n\t"
        "v_mov_b32 %[e], 0\n\t"
        "v_and_b32 %[m], 0x1ff, %[p]\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "ds_read_u8 %[e], %[m] offset:0xa00\n\t"        // head map
        "s_mov_b64 exec, -1\n\t"
        "v_mov_b32 %[t], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // inclusive max-scan: the last head at or below every byte
        "v_max_u32_dpp %[e], %[e], %[e] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %[e], %[e], %[e] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %[e], %[e], %[e] row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %[e], %[e], %[e] row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_max_u32_dpp %[e], %[e], %[e] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mov_b32_dpp %[t], %[e] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max3_u32 %[g], %[e], %[t], %[carry]\n\t"    // (sequence lane << 2) | code
        "v_and_b32 %[c], 3, %[g]\n\t"
        "v_readlane_b32 %[carry], %[g], 63\n\t"
        "ds_bpermute_b32 %[x], %[g], %[offx]\n\t"       // the owner's match displacement (lane = address bits 7:2)
        "ds_bpermute_b32 %[y], %[g], %[dlv]\n\t"        // the owner's literal cursor - output position
        "v_cmp_lt_u32 vcc, 1, %[c]\n\t"                 // match byte (codes 2, 3)
        "v_cmp_eq_u32 %[sn], 2, %[c]\n\t"               // ... from the window ring
        "s_and_b64 vcc, vcc, %[act]\n\t"
        "s_and_b64 %[sn], %[sn], %[act]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32 %[q], %[y], %[x], vcc\n\t"
        "v_add_u32 %[q], %[p], %[q]\n\t"                // code 2: source position in the frame; 1: index in the block's literals; 3: stage offset
        "v_add_u32 %[r], %[lane], %[x]\n\t"
        "v_cndmask_b32 %[r], -1, %[r], %[sn]\n\t"
        "v_and_b32 %[a], 0x7ff, %[q]\n\t"               // window ring
        "v_and_or_b32 %[b], %[q], %[litmask], %[c800]\n\t"  // literal ring
        "v_cndmask_b32 %[a], %[b], %[a], vcc\n\t"
        "v_cmp_eq_u32 vcc, 3, %[c]\n\t"
        "v_and_or_b32 %[b], %[q], %[c3ff], %[cc00]\n\t"  // stage
        "v_cmp_le_i32 %[dep], 0, %[r]\n\t"
        "v_cndmask_b32 %[a], %[a], %[b], vcc\n\t"
        "v_cmp_gt_i32 %[far], %[negw], %[r]\n\t"        // older than the window
        "ds_read_u8 %[l], %[a]\n\t"
        "s_cmp_eq_u64 %[far], 0\n\t"
        "s_cbranch_scc1 L_xb_nofar_%=\n\t"
        "s_waitcnt vmcnt(0)\n\t"                        // every window unit issued so far has arrived in the slab
        "s_mov_b64 exec, %[far]\n\t"
        "global_load_ubyte %[f], %[q], %[outb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_waitcnt vmcnt(0)\n"
        "L_xb_nofar_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32 %[val], %[l], %[f], %[far]\n\t"
        : [p] "=&v"(p), [m] "=&v"(m), [e] "=&v"(e), [t] "=&v"(t), [g] "=&v"(g), [x] "=&v"(x), [y] "=&v"(y), [q] "=&v"(q),
          [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [l] "=&v"(l), [f] "=&v"(f), [far] "=&s"(far), [dep] "=&s"(dep), [sn] "=&s"(sn),
          [val] "=&v"(val), [r] "=&v"(r), [carry] "+s"(carry)
        : [P] "s"(P), [act] "s"(act), [lane] "v"(lane), [offx] "v"(offx), [dlv] "v"(dlv), [outb] "s"(out),
          [litmask] "s"(litMask), [c800] "v"(0x800u), [c3ff] "s"(0x3ffu), [cc00] "v"(0xc00u), [negw] "s"(-(int)kXbWin)
        : "memory", "vcc", "scc");
}

__global__ __launch_bounds__(64, 8) void k_exec_b(const uint8_t *__restrict__ in, uint8_t *out_blob, const DFrame *__restrict__ frames,
                                                  const DBlock *__restrict__ blocks, const BlockSum *__restrict__ sums,
                                                  const uint64_t *__restrict__ recs, const uint8_t *__restrict__ litbuf,
                                                  int32_t *frame_status, uint64_t *frame_out_len)
{
    __shared__ __attribute__((aligned(16))) XbLds sh;
    const int lane = threadIdx.x;
    const DFrame fr = frames[blockIdx.x];
    uint8_t *out = out_blob + fr.out_offset;

    int error = fr.plan_status;
    uint32_t outPos = 0;         // bytes of this frame produced so far (frames of 4 GiB and more take k_exec)
    uint32_t flushed = 0;        // [0, flushed) has left for the slab (the last unit may still be in flight)
    uint32_t validFrom = 0;      // the window ring holds [max(validFrom, outPos - kXbWin), outPos)
    uint32_t confirmed = 0;      // [0, confirmed) is known to have ARRIVED in the slab (a wait on memory came after its stores)
    int H0 = 1, H1 = 4, H2 = 8;  // framedecompressor.go:48,59
    for (uint32_t i = 4u * (uint32_t)lane; i < kXbMap; i += 256) *(uint32_t *)&sh.map[i] = 0u;
    const bool fastOK = (uint32_t)(uintptr_t)&sh == 0u;  // xb_pass addresses the rings with immediate offsets
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
            xb_reload_window(sh, out, outPos, validFrom, flushed, lane);
            confirmed = outPos;
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
            xb_reload_window(sh, out, outPos, validFrom, flushed, lane);
            confirmed = outPos;
            continue;
        }

        // ---- the literal ring: units of 256 bytes, [litLo, litLo + 512) resident, the next unit on its way in `pend`
        uint32_t litLo = 0, litC = 0;  // litC: literal cursor (literals consumed by the passes so far)
        uint32_t pend = 0;
        const uint32_t litMask = litRle ? 0u : kXbLit - 1;
        const uint32_t litEnd = litRle ? 0u : b.lit_regen;  // refills stop here
        if (litRle) {
            sh.lit[0] = lits[0];
        } else {
            const uint32_t u0 = xb_lit_unit(lits, 0, b.lit_regen, lane), u1 = xb_lit_unit(lits, kXbLitUnit, b.lit_regen, lane);
            pend = xb_lit_unit(lits, 2 * kXbLitUnit, b.lit_regen, lane);
            *(uint32_t *)&sh.lit[4 * lane] = u0;
            *(uint32_t *)&sh.lit[kXbLitUnit + 4 * lane] = u1;
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
        for (uint32_t t = 0; t < ntiles && error == MZD_OK; t++) {
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
            const bool bad = isSeq && ML > 0 && (off <= 0 || (uint32_t)off > mstart);  // ringbuffer.go:206-214
            if (wave_any(bad)) {
                error = MZD_ERR_OFFSET;
                break;
            }
            const int offv = -off;                                         // match byte p comes from p + offv
            const int dlv = (int)(litRun + sLL - LL) - (int)lstart;        // literal byte p is literal p + dlv of the block
            const uint32_t E = tileStart + tileOut;

            uint32_t P = tileStart, hlim = tileStart, carry = 0;
            XB_STAT(0, 1);
            XB_STAT(7, __popcll(wave_ballot(isSeq && ML > 0)));
            const unsigned long long xb_t1 = XB_CLOCK();
            (void)xb_t1;
            // ---- the steady state: a tile whose passes need nothing but the pass itself (xb_pass)
            if (fastOK && tileStart >= validFrom + kXbWin && tileOut <= 1024 && tileLits <= kXbLitUnit) {
                // matches whose source is final in the slab (the window units of earlier tiles, confirmed by the last wait on
                // memory): 16 source bytes per sequence into the stage, ALL of the tile's loads in flight together -- a pass
                // then finds the byte in LDS instead of waiting for memory itself
                const uint32_t q0 = mstart - (uint32_t)off;
                const bool stg = isSeq && ML > 0 && ML <= 16 && q0 + ML <= confirmed;
                U128U sv{0, 0, 0, 0};
                if (stg) sv = *(const U128U *)(out + q0);
                while (litRun - litLo >= kXbLitUnit && litLo + kXbLit < litEnd) {
                    *(uint32_t *)&sh.lit[(litLo & (kXbLit - 1)) + 4 * lane] = pend;
                    litLo += kXbLitUnit;
                    pend = xb_lit_unit(lits, litLo + kXbLit, b.lit_regen, lane);
                }
                xb_wait_vm();  // the staged bytes are here, and so is every window unit issued by earlier tiles
                confirmed = flushed;
                *(uint4 *)&sh.stage[16 * lane] = make_uint4(sv.x, sv.y, sv.z, sv.w);
                while (tileStart - flushed >= kXbFlush) xb_flush_step(sh, out, flushed, lane, false);
                const int offx = stg ? (int)(16u * (uint32_t)lane) - (int)mstart : offv;
                XB_STAT(1, 1);
                XB_STAT(6, __popcll(wave_ballot(stg)));
                const unsigned long long xb_t2 = XB_CLOCK();
                (void)xb_t2;
                XB_STAT(8, xb_t2 - xb_t1);
                const uint32_t mcode = ((uint32_t)lane << 2) | (stg ? (uint32_t)kXbStagedHead : (uint32_t)kXbMatchHead);
                for (uint32_t lo = tileStart; lo < E; lo += kXbMap) {
                    // the heads of this stretch of the tile (the map reaches kXbMap bytes); the stretch's passes only read them
                    if (LL > 0 && lstart - lo < kXbMap) sh.map[lstart & (kXbMap - 1)] = (uint8_t)((lane << 2) | kXbLitHead);
                    if (ML > 0 && mstart - lo < kXbMap) sh.map[mstart & (kXbMap - 1)] = (uint8_t)mcode;
                    const uint32_t stretchEnd = min(E, lo + kXbMap);
                    for (; P < stretchEnd; P += 64) {
                        const uint32_t n = stretchEnd - P;
                        const uint64_t act = n >= 64 ? ~0ull : (1ull << n) - 1;
                        uint32_t val;
                        int r;
                        uint64_t dep;
                        xb_pass(P, act, (uint32_t)lane, offx, dlv, carry, out, litMask, val, r, dep);
                        XB_STAT(2, 1);
                        XB_STAT(4, dep != 0);
                        XB_STAT(5, wave_any(r < -(int)kXbWin));
                        if (dep) val = xb_resolve_in_pass(val, r >= 0 ? (uint32_t)r : (uint32_t)lane, r >= 0);
                        if ((act >> lane) & 1) sh.win[(P + (uint32_t)lane) & (kXbWin - 1)] = (uint8_t)val;
                    }
                    *(uint64_t *)&sh.map[8 * lane] = 0ull;  // every head of the stretch has been used
                }
                P = E;
                litC = litRun + tileLits;
                XB_STAT(9, XB_CLOCK() - xb_t2);
            }
#ifdef MZD_XB_STATS
            const unsigned long long xb_t3 = XB_CLOCK();
            const bool xb_general = P < E;
#endif
            while (P < E) {
                // heads of the tile's sequences, as far ahead as the map reaches
                if (P + 64 > hlim && hlim < E) {
                    const uint32_t lo = hlim, span = P + kXbMap - hlim;
                    if (LL > 0 && lstart - lo < span) sh.map[lstart & (kXbMap - 1)] = (uint8_t)((lane << 2) | kXbLitHead);
                    if (ML > 0 && mstart - lo < span) sh.map[mstart & (kXbMap - 1)] = (uint8_t)((lane << 2) | kXbMatchHead);
                    hlim = P + kXbMap;
                }
                // literal ring: the unit below the cursor's is dead
                while (litC - litLo >= kXbLitUnit && litLo + kXbLit < litEnd) {
                    *(uint32_t *)&sh.lit[(litLo & (kXbLit - 1)) + 4 * lane] = pend;
                    litLo += kXbLitUnit;
                    pend = xb_lit_unit(lits, litLo + kXbLit, b.lit_regen, lane);
                }
                const uint32_t p = P + (uint32_t)lane;
                const bool act = p < E;
                uint32_t e = 0;
                if (act) {
                    e = sh.map[p & (kXbMap - 1)];
                    sh.map[p & (kXbMap - 1)] = 0;
                }
                uint32_t g = max(wave_incl_max_dpp(e), carry);  // the last head at or below this byte: (sequence lane << 2) | kind
                carry = (uint32_t)__builtin_amdgcn_readlane((int)g, 63);
                const bool isM = (g & 2u) != 0;
                const int X = __builtin_amdgcn_ds_bpermute((int)(g & 0xFCu), offv);
                const int Y = __builtin_amdgcn_ds_bpermute((int)(g & 0xFCu), dlv);
                const uint32_t q = p + (uint32_t)(isM ? X : Y);  // match: frame-relative source position; literal: index in the block's literals
                const uint32_t vlo = max(validFrom, P > kXbWin ? P - kXbWin : 0u);
                const bool mAct = act && isM;
                const bool dep = mAct && q >= P;   // produced by this very pass (offset <= lane)
                const bool far = mAct && q < vlo;  // older than the window: final in the slab
                uint32_t val = 0;
                if (far) val = out[q];
                if (act && !dep && !far) {
                    const uint32_t a = isM ? (uint32_t)offsetof(XbLds, win) + (q & (kXbWin - 1)) : (uint32_t)offsetof(XbLds, lit) + (q & litMask);
                    val = ((const uint8_t *)&sh)[a];
                }
                const uint64_t litm = wave_ballot(act && !isM);
                if (litm) litC = (uint32_t)__builtin_amdgcn_readlane((int)q, 63 - __builtin_clzll(litm)) + 1u;
                if (wave_any(dep)) val = xb_resolve_in_pass(val, dep ? q - P : (uint32_t)lane, dep);
                XB_STAT(3, 1);
                if (act) sh.win[p & (kXbWin - 1)] = (uint8_t)val;
                P = min(P + 64, E);
                if (P - flushed >= kXbFlush) xb_flush_step(sh, out, flushed, lane);
            }
#ifdef MZD_XB_STATS
            if (xb_general) xbst[10] += XB_CLOCK() - xb_t3;
#endif
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
        frame_status[blockIdx.x] = e;
        frame_out_len[blockIdx.x] = outPos;
    }
}

}  // namespace mzd
