// mzd_huf_w.hip -- k_huf_w: Huffman literal streams, a WAVEFRONT per stream, with nothing but whole lines between the CU and memory.
//
// Replaces structure/huffman.go:221-264 (DecodeStream) and the 1- / 4-stream dispatch literals.go:290-371, like k_huf and
// k_huf_seg: same symbols, same end conditions, same statuses.
//
// k_huf_seg's method -- Huffman codes self-synchronise, so the R data bits of a stream are cut into 64 segments, a lane each,
// every lane starts an approach run ahead of its segment, and the chain of code boundaries must close (see k_huf_seg in
// mzd_kernels.hip) -- with the memory side turned around.  There every lane copied ITS OWN 128 bytes of the stream into a private
// LDS strip (eight 16-byte loads per lane, strips 48 bytes apart: every one of them a 64-line gather, and 2.7 times the stream's
// bytes) and stored ITS OWN symbols with 16-byte stores at whatever byte its share of the output began (another 64-line scatter
// per instruction, partial lines at both ends: k_huf_seg's round was 13 k cycles of strip fill, 25 k of counting and 25 k of
// write-out, and the CU's address unit was busy for the kernel's whole duration).  Here, per round of 64 segments:
//
//   load      the round's bytes ONCE, 16 bytes per lane and instruction, aligned and contiguous (three or four instructions), into
//             one flat LDS area F; every lane's bit window then reads aligned dwords of F (what lies below the stream's first
//             byte is zero there: reversebitstream.go:23-27);
//   decode    approach run, then the segment -- four symbols between two looks at the remaining bits, the symbols KEPT in registers
//             (W[i], four to a dword; the loop is unrolled over i, so the index is static).  A lane decodes on to the end of its
//             group of four and counts the symbols that START below the segment's end: no second, one-symbol-at-a-time loop;
//   validate  the chain of boundaries must close (k_huf_seg's rule; a lane that is off takes its neighbour's exit and decodes again);
//   compact   an exclusive scan of the counts gives every lane its byte offset; the lanes funnel-shift their dwords into place in
//             a staging area that takes F's place (aligned dword stores; the two partial dwords at a lane's ends byte by byte);
//   store     the staging area leaves in aligned 16-byte pieces, 1 KiB per instruction; the bytes before the first and after
//             the last whole piece of the round leave one by one (sixteen lanes, one line).
//
// A segment that makes more symbols than a lane has registers for (codes much shorter than the stream's average) halves the
// segment size for that round and runs it again: no second code path.  The segment size itself is chosen per stream so that its
// rounds are equally long (a stream of 26 000 bits: one round of 416-bit segments, not a round of 384 and a round of four lanes).
#pragma once

namespace mzd {

#ifndef MZD_HW_NW
#define MZD_HW_NW 26
#endif
#ifndef MZD_HW_MAXSEG
#define MZD_HW_MAXSEG 448
#endif
#ifndef MZD_HW_APPROACH
#define MZD_HW_APPROACH 128
#endif
constexpr int kHwNW = MZD_HW_NW;            // symbol dwords a lane keeps per segment: 4 * kHwNW symbols
constexpr int kHwMaxSeg = MZD_HW_MAXSEG;    // largest segment, in bits (a multiple of 32)
constexpr int kHwApproach = MZD_HW_APPROACH;
constexpr int kHwSlackLo = 32;              // bytes of F below the round's last bit: three symbols decoded past a segment's end (33 bits), the 64-bit window, two dwords of refill
constexpr int kHwInBytes = (64 * kHwMaxSeg / 8 + kHwSlackLo + 15 + 4 + 4 + 15) / 16 * 16;  // + alignment of the first piece, the window's upper dword, what a round's start is not known by when its bytes are requested
constexpr int kHwStageBytes = 64 * 4 * kHwNW + 16 + 16;                               // every lane's symbols + the round's offset in its 16-byte piece
constexpr int kHwPieces = (kHwInBytes + 1023) / 1024;  // 16-byte pieces per lane
constexpr int kHwWaveBytes = ((kHwInBytes > kHwStageBytes ? kHwInBytes : kHwStageBytes) + 63) / 64 * 64;
static_assert(kHwMaxSeg % 32 == 0 && kHwMaxSeg >= 64 && kHwInBytes % 16 == 0 && kHwInBytes <= 4096, "k_huf_w geometry");

#ifdef MZD_HUF_W_STATS
// 0 rounds, 1 validation rounds, 2 lanes decoded again, 3 rounds run again with half the segment; wavefront cycles: 8 load, 9 decode,
// 10 validation, 11 compaction, 12 store, 13 whole stream
__device__ unsigned long long g_huf_w_stats[16];
#define HW_CLK() __builtin_readcyclecounter()
#define HW_ADD(i, v) do { if (lane == 0) atomicAdd(&g_huf_w_stats[i], (unsigned long long)(v)); } while (0)
#else
#define HW_CLK() 0ull
#define HW_ADD(i, v) do { } while (0)
#endif

// Bit window of one lane over the flat area F (LDS byte f <-> stream byte xF + f; the stream is read from its last byte down, a
// byte from its top bit down: the bits of dword F[w] follow those of F[w + 1]).  Three dwords in registers, A = F[w + 1], B = F[w],
// C = F[w - 1], and k = bits of A already consumed, 1..32 (32: A is spent; never 0, so that 32 - k is a funnel-shift amount): the
// next 64 bits are T = {alignbit(A, B, 32 - k), alignbit(B, C, 32 - k)} -- rebuilt once per GROUP of four symbols, which consume 44
// bits at most and shift T themselves.  Instructions are what this kernel pays for (round 6's first form kept the k_huf_seg
// window, two 64-bit shifts per symbol and a refill under a branch per two: 21 VALU instructions per symbol; this: ~10).
struct HwWin {
    const uint32_t *F;
    uint32_t A, B, C, D;  // D = F[w - 2]: read a group ahead, so that an advance by one dword -- two groups in three -- waits for nothing
    int w, k;
    __device__ __forceinline__ void seek(int r, int bit)  // r = F byte that holds the bit, bit = its index from the byte's top (0..7)
    {
        const int kk = 8 * (3 - (r & 3)) + bit;  // bits of the byte's dword above the bit
        w = (r >> 2) - (kk ? 1 : 0);
        k = kk ? kk : 32;
        A = F[w + 1];
        B = F[w];
        C = F[w - 1];
        D = F[w - 2];
    }
    __device__ __forceinline__ uint64_t window() const
    {
        const uint32_t s = 32u - (uint32_t)k;
        return ((uint64_t)__builtin_amdgcn_alignbit(A, B, s) << 32) | __builtin_amdgcn_alignbit(B, C, s);
    }
    // c bits (<= 44 when MaxBits is 11, <= 32 when WIDE) have been consumed: the window moves down by 0, 1 (or, not WIDE, 2) dwords
    template <bool WIDE>
    __device__ __forceinline__ void advance(int c)
    {
        const int kn = k + c - 1, adv = kn >> 5;
        k = (kn & 31) + 1;
        w -= adv;
        uint32_t a1 = adv ? B : A, b1 = adv ? C : B, c1 = adv ? D : C;
        // (of the NEW w; read again when the window did not move: one load either way, and no lane waits for it before the next
        // group's end.  A load under `if (adv > 1)` becomes a select between an LDS and a scratch address and a flat load: measured)
        const uint32_t e2 = F[w - 2];
        if (!WIDE) {  // (two dwords: rare -- 33 bits and more in four symbols)
            const uint32_t e1 = F[w - 1];
            a1 = adv > 1 ? C : a1;
            b1 = adv > 1 ? D : b1;
            c1 = adv > 1 ? e1 : c1;
        }
        A = a1;
        B = b1;
        C = c1;
        D = e2;
    }
};
// one symbol off the top of T: the cell {symbol, nbits << 8}
__device__ __forceinline__ uint32_t hw_sym(const uint16_t *tbl, uint64_t &T, uint32_t idx_shift)
{
    const uint32_t e = tbl[(uint32_t)(T >> 32) >> idx_shift];
    T <<= e >> 8;
    return e;
}

// The round's bytes into F: stream bytes [xF, xF + nF) in whole aligned 16-byte pieces, lane j piece j + 64 c.  The four loads and
// their wait are ONE asm statement: C++ loads under their conditions are paired with their uses by the compiler (four trips to
// memory in a row, measured), and loads left in flight across asm statements may have their registers moved before they land.
// A lane without a piece (beyond nF, or wholly below the stream) loads the round's last piece and drops it.  Bytes below the
// start of the stream read as zero (reversebitstream.go:23-27).
__device__ __forceinline__ void hw_load_round(uint8_t *wl, const uint8_t *s, int xF, int nF, int lane)
{
    static_assert(kHwPieces == 4, "hw_load_round names four pieces");
    const int np = (nF + 15) >> 4;  // pieces
    const uint8_t *a[kHwPieces];
    bool live[kHwPieces];
#pragma unroll
    for (int c = 0; c < kHwPieces; c++) {
        const int pc = lane + 64 * c;
        live[c] = pc < np && xF + 16 * pc > -16;
        a[c] = s + xF + 16 * (live[c] ? pc : np - 1);  // (the last piece holds the round's first bit: inside the stream)
    }
    u32x4 q0, q1, q2, q3;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\t"
                 "global_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
    const u32x4 q[kHwPieces] = {q0, q1, q2, q3};
#pragma unroll
    for (int c = 0; c < kHwPieces; c++) {
        const int pc = lane + 64 * c, x = xF + 16 * pc;
        if (pc < np) {
            u32x4 v = live[c] ? q[c] : u32x4{0, 0, 0, 0};
            if (x < 0 && x > -16) {
                const int z = -x;
                uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                if (z >= 8) { lo = 0; hi = (hi >> (8 * (z - 8))) << (8 * (z - 8)); }
                else lo = (lo >> (8 * z)) << (8 * z);
                v = u32x4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
            }
            *(u32x4 *)(wl + 16 * pc) = v;
        }
    }
}

__device__ __forceinline__ uint32_t hw_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// WIDE: MaxBits <= 8 -- four symbols between two refills (31 + 4 * 8 <= 64); otherwise two (31 + 2 * 11 <= 64)
template <bool WIDE>
__device__ __forceinline__ void huf_w_stream(const uint8_t *__restrict__ in, const HufTask &t, const uint16_t *tbl, uint8_t *wl, uint8_t *obase,
                                             BlockSum *sums, uint32_t stream_idx, int lane)
{
    // (the task is the wavefront's: its fields, and everything computed from them, live in SGPRs)
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](uint64_t v) { return (uint64_t)uni((uint32_t)v) | ((uint64_t)uni((uint32_t)(v >> 32)) << 32); };
    const uint8_t *s = in + uni64(t.in_off);
    const int len = (int)uni(t.in_size), mb = (int)uni(t.max_bits);
    const uint32_t want = uni(t.out_size);
    const uint64_t out_off = uni64(t.out_off);
    // padding: zero bits above the marker and the marker itself (huffman.go:227-238)
    const uint32_t last = len > 0 ? uni(s[len - 1]) : 0u;
    int status = last == 0 ? MZD_ERR_BAD_PADDING : MZD_OK;
    const int a0 = last ? (int)__builtin_clz(last) - 24 + 1 : 8;
    const int R = 8 * len - a0;  // data bits
    // the stream's segment size: rounds of equal length, as few as kHwMaxSeg allows -- and short enough that a lane's expected
    // symbols (B * want / R) stay 15 % below what its registers hold
    int Bs = kHwMaxSeg;
    {
        const int cap = (int)(((uint64_t)(4 * kHwNW) * 85 / 100) * (uint64_t)(R > 0 ? R : 1) / (want ? want : 1u)) & ~31;  // bits that make that many symbols
        Bs = min(Bs, max(cap, 32));
        const int rounds = (R + 64 * Bs - 1) / (64 * Bs);
        if (rounds > 0) Bs = min(Bs, (((R + rounds - 1) / rounds + 63) / 64 + 31) & ~31);
        // (an ODD number of dwords: lanes that move in step read F dwords Bs / 32 apart -- an even stride halves the banks they hit)
        if (!(Bs & 32) && Bs > 32 && rounds > 0) {
            const int hard = (int)(((uint64_t)(4 * kHwNW) * 96 / 100) * (uint64_t)(R > 0 ? R : 1) / (want ? want : 1u));  // 4 % below the registers
            if ((int64_t)rounds * 64 * (Bs - 32) >= R) Bs -= 32;                      // (no round more for it)
            else if (Bs + 32 <= kHwMaxSeg && Bs + 32 <= hard) Bs += 32;
        }
        Bs = max(Bs, 32);
    }
    HwWin d;
    d.F = (const uint32_t *)wl;
    const uint32_t ish = 32u - (uint32_t)mb;  // the next MaxBits bits of the window's upper dword
    uint32_t *const F32 = (uint32_t *)wl;
    int p0 = 0;             // exact code boundary where the round starts
    uint32_t out_done = 0;  // symbols written by earlier rounds
    const unsigned long long c_begin = HW_CLK();
    unsigned long long acc[5] = {0, 0, 0, 0, 0}, acc_n[4] = {0, 0, 0, 0};
    (void)c_begin;
    (void)acc;
    (void)acc_n;
    uint32_t W[kHwNW];
    while (status == MZD_OK && p0 < R) {
        const unsigned long long c0 = HW_CLK();
        // ---- the round's bytes into F: stream bytes [xF, xF + nF), whole aligned 16-byte pieces, all of a lane's pieces in flight
        // together.  (Touching the NEXT round's lines from here -- an LDS-DMA load per 128-byte line into a sink, nothing to wait for --
        // was measured at nothing: 0.388 ms with, 0.382 without, config 3; twenty wavefronts per CU hide one trip to memory per round.)
        const int need_top = len - 1 - ((a0 + p0) >> 3);
        const int need_lo = len - 1 - ((a0 + min(R, p0 + 64 * Bs)) >> 3) - kHwSlackLo;
        const int xF = need_lo - (int)((uintptr_t)(s + need_lo) & 15);
        const int nF = need_top + 4 - xF;
        hw_load_round(wl, s, xF, nF, lane);
        const unsigned long long c1 = HW_CLK();
        // ---- decode (and again with half the segment if a lane ran out of registers)
        int B = Bs;
        int lo, hi, tpos = 0, epos = 0;
        uint32_t cnt = 0;
        bool act;
        unsigned long long c2 = c1, c3 = c1;
        for (;;) {
            lo = p0 + lane * B;
            act = lo < R;
            hi = min(R, lo + B);
            bool run = act;
            int spos = max(lo - kHwApproach, p0);
            bool ovf = false;
            for (int guard = 0; guard < 66; guard++) {
                if (run) {
                    int pos = spos;
                    {
                        const int a = a0 + pos;
                        d.seek((len - 1 - (a >> 3)) - xF, a & 7);
                    }
                    // approach: groups of four symbols to the first code boundary at or after lo (a lane that starts exact is there
                    // already); the group that crosses lo is taken up to that boundary only -- the window advances by any number of bits
                    while (wave_any(pos < lo)) {
                        if (pos < lo) {
                            uint64_t T = d.window();
                            const int s1 = pos + (int)(hw_sym(tbl, T, ish) >> 8);
                            const int s2 = s1 + (int)(hw_sym(tbl, T, ish) >> 8);
                            const int s3 = s2 + (int)(hw_sym(tbl, T, ish) >> 8);
                            const int s4 = s3 + (int)(hw_sym(tbl, T, ish) >> 8);
                            const int np = s1 >= lo ? s1 : (s2 >= lo ? s2 : (s3 >= lo ? s3 : s4));
                            d.advance<WIDE>(np - pos);
                            pos = np;
                        }
                    }
                    tpos = pos;
                    // the segment: groups of four symbols, kept; the symbols of a lane's last group that start at or after hi are
                    // decoded and not counted
                    uint32_t n = 0;
                    bool go = pos < hi;
#pragma unroll
                    for (int i = 0; i < kHwNW; i++) {
                        if (!wave_any(go)) break;
                        if (go) {
                            uint64_t T = d.window();
                            const uint32_t e0 = hw_sym(tbl, T, ish);
                            const uint32_t e1 = hw_sym(tbl, T, ish);
                            const uint32_t e2 = hw_sym(tbl, T, ish);
                            const uint32_t e3 = hw_sym(tbl, T, ish);
                            W[i] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e3, e2, 0x0c0c0400u), __builtin_amdgcn_perm(e1, e0, 0x0c0c0400u), 0x05040100u);
                            const int s1 = pos + (int)(e0 >> 8), s2 = s1 + (int)(e1 >> 8), s3 = s2 + (int)(e2 >> 8), s4 = s3 + (int)(e3 >> 8);
                            d.advance<WIDE>(s4 - pos);
                            if (wave_any(s3 >= hi)) {  // a lane's last group: 1..3 of its symbols start below hi
                                if (s3 >= hi) {
                                    n += 1u + (s1 < hi ? 1u : 0u) + (s2 < hi ? 1u : 0u);
                                    pos = s1 >= hi ? s1 : (s2 >= hi ? s2 : s3);
                                    go = false;
                                } else {
                                    n += 4;
                                    pos = s4;
                                    go = s4 < hi;
                                }
                            } else {
                                n += 4;
                                pos = s4;
                                go = s4 < hi;
                            }
                        }
                    }
                    ovf = go;  // symbols to go and no register left
                    cnt = n;
                    epos = pos;
                }
                if (wave_any(ovf)) break;
                if (guard == 0) c2 = HW_CLK();
                // validation: the chain of boundaries must close (lanes run in lockstep here)
                const int tnext = __shfl_down(tpos, 1, 64);
                const bool nact = (bool)__shfl_down((int)act, 1, 64) && lane < 63;
                const bool bad = act && nact && epos != tnext;
                if (!wave_any(bad)) break;
                acc_n[1] += 1;
                acc_n[2] += (unsigned long long)__popcll(wave_ballot(bad));
                run = (bool)__shfl_up((int)bad, 1, 64) && lane > 0;
                spos = __shfl_up(epos, 1, 64);  // (>= lo: the lane below ended at or after its hi)
                if (run && spos >= hi) {        // nothing of the segment is this lane's: the lane below decoded through it
                    tpos = epos = spos;
                    cnt = 0;
                    run = false;
                }
            }
            c3 = HW_CLK();
            if (!wave_any(ovf) || B <= 32) break;
            B = max(32, (B >> 1) & ~31);
            acc_n[3] += 1;
        }
        acc_n[0] += 1;
        const uint32_t incl = wave_incl_scan_u32(act ? cnt : 0u, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint64_t am = wave_ballot(act);
        p0 = __builtin_amdgcn_readlane(epos, 63 - __builtin_clzll(am));  // lane 0 is active: am != 0
        if (out_done + total > want) {  // the serial loop stops at `want` symbols with bits left (literals.go:320,332,349,366)
            status = MZD_ERR_HUF_LENGTH;
            break;
        }
        // ---- compaction: the lane's cnt symbols to staging bytes [T, T + cnt), T = lead + the symbols of the lanes below; the
        // staging area takes F's place (every read of F lies before this point in the wavefront's program order)
        uint8_t *const O = obase + out_off + out_done;
        const uint32_t lead = (uint32_t)((uintptr_t)O & 15);
#ifndef MZD_HW_ABL_NOCOMPACT  /* ablations: timing only, wrong results */
        {
            const uint32_t c = act ? cnt : 0u;
            const uint32_t T = lead + (incl - c), sh = T & 3, d0 = T >> 2;
            const uint32_t nfull = (c + sh) >> 2;  // dwords m < nfull lie wholly inside [T, T + c) -- but dword 0 when sh != 0
            const uint32_t shb = 32u - 8u * sh;
            uint32_t headv = 0, tailv = 0;
#pragma unroll
            for (int m = 0; m <= kHwNW; m++) {
                const uint32_t lw = m > 0 ? W[m - 1] : 0u, hw = m < kHwNW ? W[m] : 0u;
                const uint32_t v = sh ? hw_alignbit(hw, lw, shb) : hw;
                if (m == 0) headv = v;
                if ((uint32_t)m == nfull) tailv = v;
                if ((uint32_t)m < nfull && (m > 0 || sh == 0)) F32[d0 + m] = v;
            }
            // the partial dwords at the lane's two ends, byte by byte: dword 0 from byte sh on, dword nfull up to byte (c + sh) & 3
            const uint32_t tb = (c + sh) & 3;
            if (c) {
                if (sh) {
                    const uint32_t be = nfull == 0 ? tb : 4u;  // (all of the lane's bytes in one dword: it ends at tb)
#pragma unroll
                    for (uint32_t b = 1; b < 4; b++)
                        if (b >= sh && b < be) wl[4 * d0 + b] = (uint8_t)(headv >> (8 * b));
                }
                if (tb && (nfull > 0 || sh == 0)) {  // (sh != 0 and no whole dword: the head's loop above wrote the lane's bytes)
#pragma unroll
                    for (uint32_t b = 0; b < 3; b++)
                        if (b < tb) wl[4 * (d0 + nfull) + b] = (uint8_t)(tailv >> (8 * b));
                }
            }
        }
#endif
        const unsigned long long c4 = HW_CLK();
        // ---- store: staging byte q <-> address (O - lead) + q, valid q in [lead, end)
        {
            uint8_t *const G = O - lead;
            const uint32_t end = lead + total;
            const uint32_t cfirst = (lead + 15) >> 4, cend = end >> 4;  // whole 16-byte pieces [cfirst, cend)
#if !defined(MZD_HW_ABL_NOSTORE) && !defined(MZD_HW_ABL_NOCOMPACT)
            if (cend > cfirst) {
                for (uint32_t c = cfirst + (uint32_t)lane; c < cend; c += 64) {
                    const uint4 v = *(const uint4 *)(wl + 16 * c);
                    *(uint4 *)(G + 16 * c) = v;
                }
            }
            // the bytes before the first and after the last whole piece (or all of them, if the round has no whole piece)
            const uint32_t h0 = lead, h1 = cend >= cfirst ? min(end, 16 * cfirst) : end;
            if (lead && (uint32_t)lane < 16 && h0 + (uint32_t)lane < h1) G[h0 + lane] = wl[h0 + lane];
            if (cend >= cfirst && (uint32_t)lane < 16) {
                const uint32_t q = max(16 * cend, lead) + (uint32_t)lane;  // (lead == 0 and no whole piece: everything is tail)
                if (q < end && (q >= h1 || !lead)) G[q] = wl[q];
            }
#endif
        }
        out_done += total;
        {
            const unsigned long long c5 = HW_CLK();
            (void)c5;
            acc[0] += c1 - c0;
            acc[1] += c2 - c1;
            acc[2] += c3 - c2;
            acc[3] += c4 - c3;
            acc[4] += c5 - c4;
        }
    }
    for (int i = 0; i < 5; i++) HW_ADD(8 + i, acc[i]);
    for (int i = 0; i < 4; i++) HW_ADD(i, acc_n[i]);
    HW_ADD(13, HW_CLK() - c_begin);
    // ---- status of the whole stream: what the serial loop gives (k_huf_seg's rule)
    if (status == MZD_OK) {
        const int rem = R - p0;
        if (out_done < want) status = rem < 0 ? MZD_ERR_HUF_BITS : MZD_ERR_HUF_LENGTH;
        else if (rem < 0) status = MZD_ERR_HUF_BITS;
    }
    if (status != MZD_OK && lane == 0) atomicMin(&sums[uni(t.block)].huf_err, (stream_idx << 8) | (uint32_t)status);
}

// One workgroup = the (up to) four streams of a literals section = four wavefronts that share the section's decode table in LDS
// (<= 4 KiB) + kHwWaveBytes each.
__global__ __launch_bounds__(256) void k_huf_w(const uint8_t *__restrict__ in, const HufTask *__restrict__ tasks, uint32_t n_tasks,
                                               const uint16_t *__restrict__ huf_entries, uint8_t *__restrict__ litbuf, uint8_t *out_blob,
                                               BlockSum *sums, uint32_t table_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tbl = (uint16_t *)smem;  // the section's decode table (table_bytes, a multiple of 64)
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
    uint8_t *wl = smem + table_bytes + (size_t)wave * kHwWaveBytes;
    uint8_t *const obase = __builtin_amdgcn_readfirstlane((int)t.pad) ? out_blob : litbuf;  // (see k_huf)
    if (__builtin_amdgcn_readfirstlane((int)t.max_bits) <= 8) huf_w_stream<true>(in, t, tbl, wl, obase, sums, tid & 3u, lane);
    else huf_w_stream<false>(in, t, tbl, wl, obase, sums, tid & 3u, lane);
}

}  // namespace mzd
