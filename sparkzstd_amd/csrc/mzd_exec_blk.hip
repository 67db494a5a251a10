// mzd_exec_blk.hip -- the blocks of ONE frame executed side by side (few large frames: framedecompressor.go:246-254 walks a
// frame's blocks one after the other, and so does k_exec_b's wavefront -- 2.7 us per 64 sequences, 354 MB/s for a frame
// however large the chip).
//
// A block's bytes depend on the blocks before it only through the matches that reach back over its start (ringbuffer.go:242-277
// RepeatBeforeIndex), directly or through later matches that copy those bytes again.  k_exec_b only ever COPIES bytes, so what
// it makes of a byte that lay before the block's start is that byte, whatever it was.  Block mode runs every block of every
// frame as its own wavefront job (k_exec_b<true>), NP times, each time with a different PATTERN in place of the data before
// the block's start:
//     pass 0        byte 0 of the position x (frame-relative)                 -> written to the frame's slab itself
//     pass 1 (, 2)  byte 1 (, 2) of x                                         -> to planes beside the slab
//     last pass     byte 0 of x  XOR  (1 + the top seven bits of x)           -> to the last plane
// A byte of the block that does not derive from earlier blocks is the same in all passes (and final in the slab after pass 0).
// One that does differs between pass 0 and the last pass (the XOR term is never zero), and the passes together spell the
// position it was copied from: its ORIGIN, below the block's start.  (NP = 3 covers frames below 8 MiB, NP = 4 below 2 GiB.)
// k_blk_fixup then walks the blocks of a frame in order -- a step is a parallel gather slab[x] = slab[origin(x)] over the
// block's derived bytes (their origins lie in blocks that are final), a few microseconds instead of the block's execution.
// The offset history at every block's start (framedecompressor.go:23) and the block's first output byte come from a scan
// over the blocks' summaries (k_blk_scan), which also applies the per-block checks of the serial walk in its order.
// A JOB is a segment of one or more consecutive blocks of a frame (from a block flagged kBjHead to the next), executed in order by one wavefront with the
// pattern in place of whatever lies before the SEGMENT's start; the fix-up walk has a step per segment.  Fewer, longer jobs
// trade the passes' parallelism for a shorter walk: two blocks per job when the batch has few frames.
#pragma once

namespace mzd {

// ---- scan: a wavefront per frame, 64 blocks per round (their summaries loaded side by side, then walked in order)
__global__ __launch_bounds__(64) void k_blk_scan(const DFrame *__restrict__ frames, const DBlock *__restrict__ blocks,
                                                 const BlockSum *__restrict__ sums, BJob *__restrict__ jobs, BFrame *__restrict__ bframes,
                                                 uint32_t gs, uint32_t *__restrict__ heads, uint32_t *__restrict__ walk, int32_t *frame_hist)
{
    // heads: [0] a counter (zero at launch), [1 ...] the blocks where a job starts, in no particular order
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    const DFrame fr = frames[f];
    int error = fr.plan_status;
    uint64_t outPos = 0;
    int H0 = 1, H1 = 4, H2 = 8;  // framedecompressor.go:48,59
    if (fr.continues) {  // a chunk of a frame: behind the window bytes its slab begins with, with the history of the blocks before
        outPos = fr.start;
        H0 = fr.hist[0];
        H1 = fr.hist[1];
        H2 = fr.hist[2];
    }
    uint32_t n_ok = error == MZD_OK ? fr.n_blocks : 0u;
    bool prev_direct = true;  // (the block before the frame's first: a job starts there anyway)
    uint32_t reach = 0;       // the largest offset code of the frame's blocks (BlockSum::reach; ~0 when the sequence kernel does not say)
    // (a chunk's repeat offsets may be the history it was handed: offsets no block of the chunk spells out)
    if (fr.continues) reach = (uint32_t)max(max(fr.hist[0], fr.hist[1]), fr.hist[2]);
    for (uint32_t base = 0; base < fr.n_blocks; base += 64) {
        const uint32_t bi = base + lane;
        const bool valid = bi < fr.n_blocks;
        uint32_t bo = 0, flags = 0;
        int e = MZD_OK, h0 = 0, h1 = 0, h2 = 0;
        bool hist = false;
        if (valid) {
            // (field by field: whole structs indexed per lane went through 80 bytes of scratch)
            const DBlock *const bp = blocks + fr.first_block + bi;
            if (bp->type != MZD_BLOCK_COMPRESSED) {
                bo = bp->size;
                flags = kBjDirect;
            } else {
                // the checks of k_exec_b's serial walk, in its order
                const BlockSum *const sp = sums + fr.first_block + bi;
                const uint32_t huf_err = sp->huf_err, lit_total = sp->lit_total, lit_regen = bp->lit_regen;
                e = huf_err != 0xFFFFFFFFu ? (int)(huf_err & 0xFF) : sp->status;
                if (e == MZD_OK && bp->n_seq == 0) e = bp->pad[1];  // (zero sequences in the two-byte form: the planner's verdict)
                if (e == MZD_OK && lit_total > lit_regen) e = MZD_ERR_LITERALS;  // sequence_execution.go:27-29
                bo = sp->out_total + (lit_regen - min(lit_total, lit_regen));
                if (e == MZD_OK && bo > kBlockMax) e = MZD_ERR_CORRUPT_SIZES;
                if (bp->n_seq == 0) flags = kBjDirect;
                else {
                    reach = max(reach, sp->reach);
                    hist = true;
                    h0 = sp->hist[0];
                    h1 = sp->hist[1];
                    h2 = sp->hist[2];
                }
            }
        }
        // ---- the round's 64 blocks at once: where each starts (a prefix sum), which is the first that fails a check (a ballot),
        // and the offset history at each one's start -- the blocks' history steps are maps "slot k of the history before, minus
        // d" or constants, and such maps compose: a scan over the lanes (one 1 GiB frame: 8 192 blocks walked one by one were
        // 1.7 ms of a 20 ms pass)
        const uint32_t n = min(64u, fr.n_blocks - base);
        const uint32_t incl = wave_incl_scan_dpp(valid ? bo : 0u);
        uint32_t myStart = (uint32_t)outPos + incl - bo;
        int m0 = H0, m1 = H1, m2 = H2;
        if (error == MZD_OK) {
            int ee = e;
            if (valid && ee == MZD_OK && outPos + incl > fr.out_capacity) ee = MZD_ERR_DST_FULL;
            const uint64_t bad = wave_ballot(valid && ee != MZD_OK);
            const uint32_t nok = bad ? (uint32_t)__builtin_ctzll(bad) : n;  // blocks of this round that pass
            // history maps: conc[i] ? value v[i] : (slot k[i] of the history at the block's start) - v[i]
            bool c0 = false, c1 = false, c2 = false;
            uint32_t k0 = 0, k1 = 1, k2 = 2, v0 = 0, v1 = 0, v2 = 0;  // (a block without sequences: the identity)
            auto dec = [](int h, bool &c, uint32_t &k, uint32_t &v) {
                if (h > 0) { c = true; k = 0; v = (uint32_t)h; }
                else { const uint32_t u = (uint32_t)(-h - 1); c = false; k = min(u & 3u, 2u); v = u >> 2; }  // (resolve_hist / sel3)
            };
            if (hist) { dec(h0, c0, k0, v0); dec(h1, c1, k1, v1); dec(h2, c2, k2, v2); }
            // inclusive scan: lane L holds the map of blocks 0..L of the round
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const bool ac0 = (bool)__shfl_up((int)c0, sft, 64), ac1 = (bool)__shfl_up((int)c1, sft, 64), ac2 = (bool)__shfl_up((int)c2, sft, 64);
                const uint32_t ak0 = (uint32_t)__shfl_up((int)k0, sft, 64), ak1 = (uint32_t)__shfl_up((int)k1, sft, 64), ak2 = (uint32_t)__shfl_up((int)k2, sft, 64);
                const uint32_t av0 = (uint32_t)__shfl_up((int)v0, sft, 64), av1 = (uint32_t)__shfl_up((int)v1, sft, 64), av2 = (uint32_t)__shfl_up((int)v2, sft, 64);
                if ((int)lane >= sft) {
                    // (the three-way picks by masks: as `k == 0 ? a : (k == 1 ? b : c)` over values that live in a lambda's captures the
                    // compiler made them lookups in a table it kept in scratch -- 112 bytes per lane, 138 scratch accesses per round)
                    auto pick = [](uint32_t k, uint32_t x0, uint32_t x1, uint32_t x2) -> uint32_t {
                        return (x0 & (0u - (uint32_t)(k == 0))) | (x1 & (0u - (uint32_t)(k == 1))) | (x2 & (0u - (uint32_t)(k >= 2)));
                    };
                    auto comp = [&](bool &c, uint32_t &k, uint32_t &v) {  // this lane's component after the earlier lanes' map A
                        if (c) return;
                        const bool ac = pick(k, (uint32_t)ac0, (uint32_t)ac1, (uint32_t)ac2) != 0;
                        const uint32_t ak = pick(k, ak0, ak1, ak2), av = pick(k, av0, av1, av2);
                        if (ac) { c = true; k = 0; v = av - v; }
                        else { k = ak; v = av + v; }
                    };
                    comp(c0, k0, v0);
                    comp(c1, k1, v1);
                    comp(c2, k2, v2);
                }
            }
            auto apply = [&](bool c, uint32_t k, uint32_t v) -> int { return c ? (int)v : (int)((uint32_t)(k == 0 ? H0 : (k == 1 ? H1 : H2)) - v); };
            // the history at this block's start: the map of the lanes before it (lane 0: the history the round starts with)
            {
                const bool pc0 = (bool)__shfl_up((int)c0, 1, 64), pc1 = (bool)__shfl_up((int)c1, 1, 64), pc2 = (bool)__shfl_up((int)c2, 1, 64);
                const uint32_t pk0 = (uint32_t)__shfl_up((int)k0, 1, 64), pk1 = (uint32_t)__shfl_up((int)k1, 1, 64), pk2 = (uint32_t)__shfl_up((int)k2, 1, 64);
                const uint32_t pv0 = (uint32_t)__shfl_up((int)v0, 1, 64), pv1 = (uint32_t)__shfl_up((int)v1, 1, 64), pv2 = (uint32_t)__shfl_up((int)v2, 1, 64);
                if (lane > 0) {
                    m0 = apply(pc0, pk0, pv0);
                    m1 = apply(pc1, pk1, pv1);
                    m2 = apply(pc2, pk2, pv2);
                }
            }
            // what the next round starts with: the map of the round's last good block, and its end
            if (nok > 0) {
                const int last = (int)nok - 1;
                const int n0 = apply((bool)__shfl((int)c0, last, 64), (uint32_t)__shfl((int)k0, last, 64), (uint32_t)__shfl((int)v0, last, 64));
                const int n1 = apply((bool)__shfl((int)c1, last, 64), (uint32_t)__shfl((int)k1, last, 64), (uint32_t)__shfl((int)v1, last, 64));
                const int n2 = apply((bool)__shfl((int)c2, last, 64), (uint32_t)__shfl((int)k2, last, 64), (uint32_t)__shfl((int)v2, last, 64));
                H0 = n0;
                H1 = n1;
                H2 = n2;
                outPos += (uint32_t)__shfl((int)incl, last, 64);
            }
            if (nok < n) {
                error = __shfl(ee, (int)nok, 64);
                n_ok = base + nok;
            }
        }
        const bool pd_up = (bool)__shfl_up((int)(flags & kBjDirect), 1, 64);  // (all lanes)
        const bool pd = lane == 0 ? prev_direct : pd_up;
        if (valid) {
            // a job starts at every gs-th block (of the batch's numbering), at a block without sequences and behind one
            uint32_t jf = flags | (bi >= n_ok ? kBjSkip : 0u);
            if (bi == 0 || (fr.first_block + bi) % gs == 0 || (flags & kBjDirect) || pd) jf |= kBjHead;
            static_assert(sizeof(BJob) == 32, "stored as two 16-byte halves");
            uint4 *const jp = (uint4 *)(jobs + fr.first_block + bi);  // start, len, H0, H1 | H2, flags, frame, pad
            jp[0] = make_uint4(myStart, bo, (uint32_t)m0, (uint32_t)m1);
            jp[1] = make_uint4((uint32_t)m2, jf, f, 0u);
        }
        prev_direct = (bool)(__shfl((int)flags, 63, 64) & kBjDirect);
        {
            // this round's job starts go on the list (skipped blocks start nothing)
            const bool head = valid && (bi == 0 || (fr.first_block + bi) % gs == 0 || (flags & kBjDirect) || pd) && bi < n_ok;
            const uint64_t hm = wave_ballot(head);
            if (hm) {
                uint32_t at = 0;
                if (lane == 0) at = atomicAdd(&heads[0], (uint32_t)__popcll(hm));
                at = (uint32_t)__shfl((int)at, 0, 64);
                if (head) heads[1 + at + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u))] = fr.first_block + bi;
            }
        }
    }
    // how far back the frame's matches go: an origin lies less than the largest offset below its segment's start, so with offsets
    // up to 8 MiB the low 23 bits of the position say which one it is and the pass that spells the bits above is not run
    reach = wave_max_u32(reach);
    if (lane == 0) {
        // the frames the fix-up walk has anything to do for (a frame of one block is final after pass 0): walk[0] counts them
        if (fr.n_blocks > 1) walk[1 + atomicAdd(&walk[0], 1u)] = f;
        BFrame bf;
        bf.status = error;
        bf.out_len = (uint32_t)outPos;
        bf.n_ok = n_ok;
        bf.first_bad = 0xFFFFFFFFu;
        bf.cnt = 0;
        bf.bail = 0;
        bf.high = reach > (1u << 23) ? 1u : 0u;
        bf.pad = 0;
        bframes[f] = bf;
        if (frame_hist) {  // the history behind the frame's last good block: what the next chunk starts with
            frame_hist[3 * f] = H0;
            frame_hist[3 * f + 1] = H1;
            frame_hist[3 * f + 2] = H2;
        }
    }
}

// ---- the patterns: pat[p * stride + x] for x < n
__global__ void k_blk_pattern(uint8_t *pat, uint64_t stride, uint32_t n, uint32_t np)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // four positions each
    const uint32_t x = 4 * i;
    if (x >= n) return;
    for (uint32_t p = 0; p < np; p++) {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t y = x + j;
            const uint32_t v = p < 2 ? (y >> (8 * p)) & 0xFFu : (p == 2 ? (y & 0xFFu) ^ (((y >> 16) & 0x7Fu) + 1u) : (y >> 23) & 0xFFu);
            w |= v << (8 * j);
        }
        *(uint32_t *)(pat + p * stride + x) = w;
    }
}

// ---- fix-up: G workgroups per frame walk its blocks in order; a block's derived bytes are gathered from their origins.
// A step's chain is: see the step before finished -> gather -> store -> publish.  Everything that does not depend on the
// blocks before (the block's own bytes in the planes, which of them are derived, their origins) is loaded BEFORE the wait, and
// the gathers of a 16-byte chunk are issued together (a byte that is not derived reads itself): one memory latency per step
// and thread, not one per byte.
// What one workgroup writes and another gathers a step later crosses the XCDs' L2 caches.  Fences of agent scope would do
// (write back / invalidate the whole L2, per workgroup and step: 160 us a step with 1 024 workgroups); instead the few
// accesses concerned go to memory themselves -- the gathers are agent-scope loads (sc1), the chunk stores write through
// (sc0 sc1); a step is
// published after a wait for its stores.
constexpr int kFixK = 2;  // chunks per thread and round
constexpr uint32_t kFixMaxG = 256;  // fix-up workgroups per frame at most (the stride of `done`)

__device__ __forceinline__ void fix_store16(uint8_t *p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// The origin of a derived byte from what the passes left at its position: byte 0 and byte 1 of the origin, the XOR term that
// marks it as derived (1 + bits 16..22) and, for a frame whose matches may reach back 8 MiB or more, bits 23..30.  Without
// those: the origin is the position below the segment's start S, within 8 MiB of it, that has these low 23 bits.
template <int NP>
__device__ __forceinline__ uint32_t fix_origin(uint32_t b0, uint32_t b1, uint32_t dj, uint32_t bh, uint32_t S, bool high)
{
    const uint32_t low = b0 | (b1 << 8) | ((dj - 1u) << 16);
    if (NP == 4 && high) return low | (bh << 23);
    if (NP == 3) return low;  // (frames below 8 MiB)
    return (S - 1u) - (((S - 1u) - low) & 0x7FFFFFu);
}

template <int NP>
struct FixChunks {
    U128U a[kFixK], e[kFixK], b1[kFixK], b2[kFixK];
    uint32_t x[kFixK];
    bool need[kFixK];
};

template <int NP>
__device__ __forceinline__ void fix_load(FixChunks<NP> &C, const uint8_t *p0, const uint8_t *p1, const uint8_t *pH, const uint8_t *pE, uint32_t S,
                                         uint32_t n, uint32_t c0, uint32_t cstep, bool high)
{
    const uint32_t nfull = n >> 4;  // whole chunks (the last bytes of a block go one by one)
#pragma unroll
    for (int k = 0; k < kFixK; k++) {
        const uint32_t c = c0 + (uint32_t)k * cstep;
        C.x[k] = S + 16 * c;
        C.need[k] = c < nfull;
        C.a[k] = C.e[k] = U128U{0, 0, 0, 0};
        if (C.need[k]) {
            C.a[k] = *(const U128U *)(p0 + C.x[k]);
            C.e[k] = *(const U128U *)(pE + C.x[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < kFixK; k++) {
        C.need[k] = ((C.a[k].x ^ C.e[k].x) | (C.a[k].y ^ C.e[k].y) | (C.a[k].z ^ C.e[k].z) | (C.a[k].w ^ C.e[k].w)) != 0;
        C.b1[k] = C.b2[k] = U128U{0, 0, 0, 0};
        if (C.need[k]) {
            C.b1[k] = *(const U128U *)(p1 + C.x[k]);
            if (NP == 4 && high) C.b2[k] = *(const U128U *)(pH + C.x[k]);
        }
    }
}

template <int NP>
__device__ __forceinline__ void fix_gather(const FixChunks<NP> &C, uint8_t *p0, uint32_t S, bool high)
{
#pragma unroll
    for (int k = 0; k < kFixK; k++) {
        if (!C.need[k]) continue;
        const uint32_t aw[4] = {C.a[k].x, C.a[k].y, C.a[k].z, C.a[k].w}, ew[4] = {C.e[k].x, C.e[k].y, C.e[k].z, C.e[k].w};
        const uint32_t w1[4] = {C.b1[k].x, C.b1[k].y, C.b1[k].z, C.b1[k].w}, w2[4] = {C.b2[k].x, C.b2[k].y, C.b2[k].z, C.b2[k].w};
        // A derived byte is fetched from its origin.  (A load per byte, derived or not: the walk was 73 ms for 64 frames of
        // 128 MiB whose blocks do reach back; one load for four bytes whose origins follow each other, byte loads otherwise:
        // 58 ms at first, 43.6 ms with sixteen workgroups per frame; grouped by distance as below: 27 ms.  One 16-byte load per
        // distance, four per chunk behind one wait: slower.)
        // per aligned dword of the chunk: the derived bytes that are the same DISTANCE from their origins (a match copied them
        // together) come with one four-byte load at (dword position - distance); a dword touches two matches at most in the
        // common case, so two such loads, and single bytes for what is left
        uint32_t o[4];
        uint32_t ldA[4], ldB[4], mA[4], mB[4], b8[16];
        bool hasA[4], hasB[4], get[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t dlt[4], keep = 0, okq = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int sh = 8 * i;
                const uint32_t aj = (aw[q] >> sh) & 0xFF, dj = aj ^ ((ew[q] >> sh) & 0xFF);
                uint32_t org = fix_origin<NP>(aj, (w1[q] >> sh) & 0xFF, dj, (w2[q] >> sh) & 0xFF, S, high);
                const bool ok = dj != 0 && org < S && org >= (uint32_t)i;  // (beyond the job's start: only in a failed job; reads as 0)
                dlt[i] = org - (uint32_t)i;
                okq |= ok ? 1u << i : 0u;
                if (dj == 0) keep |= aj << sh;                  // not derived: the byte itself
                get[4 * q + i] = dj != 0 && org < S && !ok;     // (an origin in the frame's first three bytes: a byte load)
                b8[4 * q + i] = org;
            }
            o[q] = keep;
            // group A: the bytes at the distance of the first derived byte; group B: of the first one not in A
            uint32_t rem = okq;
            hasA[q] = rem != 0;
            const uint32_t iA = rem ? (uint32_t)__builtin_ctz(rem) : 0u;
            const uint32_t dA = iA == 0 ? dlt[0] : (iA == 1 ? dlt[1] : (iA == 2 ? dlt[2] : dlt[3]));
            uint32_t ma = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) ma |= ((rem >> i) & 1u) && dlt[i] == dA ? 1u << i : 0u;
            rem &= ~ma;
            hasB[q] = rem != 0;
            const uint32_t iB = rem ? (uint32_t)__builtin_ctz(rem) : 0u;
            const uint32_t dB = iB == 0 ? dlt[0] : (iB == 1 ? dlt[1] : (iB == 2 ? dlt[2] : dlt[3]));
            uint32_t mb = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) mb |= ((rem >> i) & 1u) && dlt[i] == dB ? 1u << i : 0u;
            rem &= ~mb;
#pragma unroll
            for (int i = 0; i < 4; i++) get[4 * q + i] = get[4 * q + i] || ((rem >> i) & 1u);
            ldA[q] = dA;
            ldB[q] = dB;
            mA[q] = ((ma & 1u) * 0xFFu) | ((ma & 2u) * 0x7F80u) | ((ma & 4u) * 0x3FC000u) | ((ma & 8u) * 0x1FE00000u);  // bit i -> byte i
            mB[q] = ((mb & 1u) * 0xFFu) | ((mb & 2u) * 0x7F80u) | ((mb & 4u) * 0x3FC000u) | ((mb & 8u) * 0x1FE00000u);
        }
        // the single bytes first (the compiler's own loads), the dwords behind them: one wait for all of them
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (get[j]) b8[j] = __hip_atomic_load(p0 + b8[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (Four-byte agent-scope loads at BYTE-aligned addresses: the hardware takes them, C++ has no name for them -- an atomic
        // load through a misaligned uint32_t pointer is undefined behaviour for the compiler -- so the eight of a chunk are written
        // out in ONE asm statement, each under its own lane mask (a dword without a group loads nothing: unconditional loads from
        // the chunk's own position cost the walk of one 1 GiB frame 20 %), with the wait inside the statement: a load left in
        // flight across statements lands in a register the compiler believes it may already use for something else.)
        uint32_t vA[4] = {0, 0, 0, 0}, vB[4] = {0, 0, 0, 0};
        {
            const uint64_t mA0 = wave_ballot(hasA[0]), mA1 = wave_ballot(hasA[1]), mA2 = wave_ballot(hasA[2]), mA3 = wave_ballot(hasA[3]);
            const uint64_t mB0 = wave_ballot(hasB[0]), mB1 = wave_ballot(hasB[1]), mB2 = wave_ballot(hasB[2]), mB3 = wave_ballot(hasB[3]);
            unsigned long long save;
            asm volatile("s_mov_b64 %[save], exec\n\t"
                         "s_mov_b64 exec, %[mA0]\n\tglobal_load_dword %[a0], %[pa0], off sc1\n\t"
                         "s_mov_b64 exec, %[mB0]\n\tglobal_load_dword %[b0], %[pb0], off sc1\n\t"
                         "s_mov_b64 exec, %[mA1]\n\tglobal_load_dword %[a1], %[pa1], off sc1\n\t"
                         "s_mov_b64 exec, %[mB1]\n\tglobal_load_dword %[b1], %[pb1], off sc1\n\t"
                         "s_mov_b64 exec, %[mA2]\n\tglobal_load_dword %[a2], %[pa2], off sc1\n\t"
                         "s_mov_b64 exec, %[mB2]\n\tglobal_load_dword %[b2], %[pb2], off sc1\n\t"
                         "s_mov_b64 exec, %[mA3]\n\tglobal_load_dword %[a3], %[pa3], off sc1\n\t"
                         "s_mov_b64 exec, %[mB3]\n\tglobal_load_dword %[b3], %[pb3], off sc1\n\t"
                         "s_mov_b64 exec, %[save]\n\t"
                         "s_waitcnt vmcnt(0)"
                         : [a0] "+v"(vA[0]), [a1] "+v"(vA[1]), [a2] "+v"(vA[2]), [a3] "+v"(vA[3]), [b0] "+v"(vB[0]), [b1] "+v"(vB[1]),
                           [b2] "+v"(vB[2]), [b3] "+v"(vB[3]), [save] "=&s"(save)
                         : [pa0] "v"(p0 + ldA[0]), [pa1] "v"(p0 + ldA[1]), [pa2] "v"(p0 + ldA[2]), [pa3] "v"(p0 + ldA[3]),
                           [pb0] "v"(p0 + ldB[0]), [pb1] "v"(p0 + ldB[1]), [pb2] "v"(p0 + ldB[2]), [pb3] "v"(p0 + ldB[3]),
                           [mA0] "s"(mA0), [mA1] "s"(mA1), [mA2] "s"(mA2), [mA3] "s"(mA3), [mB0] "s"(mB0), [mB1] "s"(mB1), [mB2] "s"(mB2),
                           [mB3] "s"(mB3)
                         : "memory");
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            o[q] |= (vA[q] & mA[q]) | (vB[q] & mB[q]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (get[4 * q + i]) o[q] |= (b8[4 * q + i] & 0xFF) << (8 * i);
        }
        fix_store16(p0 + C.x[k], u32x4{o[0], o[1], o[2], o[3]});
    }
}

// RESCUE = false: the walk proper.  RESCUE = true: ONE workgroup per frame finishes the walk of a frame whose workgroups gave up
// waiting for each other (`bail`: siblings that were not resident -- the launch is sized so that they are, but nothing promises it
// when another stream or process shares the GPU).  A step's gathers are all-or-nothing per workgroup (a workgroup gives up only
// at the wait in front of them) and `done[g]` says how many steps workgroup g finished, so the rescue redoes exactly the chunk
// sets that were not gathered, step by step, without waiting for anybody: the frame is decoded instead of failing with
// MZD_ERR_DEVICE.  (A chunk that WAS gathered must not be looked at again: which bytes are derived is read from the slab's own
// pass-0 bytes.)
template <int NP, bool RESCUE>
__global__ __launch_bounds__(256) void k_blk_fixup(uint8_t *out_blob, const uint8_t *pl1, const uint8_t *pl2, const uint8_t *pl3,
                                                   const DFrame *__restrict__ frames, const BJob *__restrict__ jobs, BFrame *bframes, uint32_t G,
                                                   uint32_t spread, uint32_t *done, uint32_t test_bail_step, const uint32_t *__restrict__ walk)
{
    // XCD placement, for speed only (workgroup b runs on XCD b % 8 -- observed, not promised).  Many frames (`spread` = 0): with
    // several workgroups per frame the launch has eight times the workgroups and the ones on a frame's XCD do its work, so that a
    // step's hand-off stays inside one L2.  Few frames (`spread` = 1): the launch is the frames' workgroups and a frame's sit on
    // all XCDs -- one XCD cannot move a large frame's planes fast enough (mzd_api.hip)
    uint32_t w = blockIdx.x;
    if (!RESCUE && G > 1 && !spread) {
        w = blockIdx.x >> 3;
        if ((blockIdx.x & 7) != ((w / G) & 7)) return;  // (w / G: the frame's slot in the walk list)
    }
    // (the launch is for the frames of more than one block, listed by k_blk_scan: in a batch of 2 000 single-block frames and two
    // large ones, two frames are walked, each by as many workgroups as a batch of two would give it)
    const uint32_t slot = RESCUE ? w : w / G, g0 = RESCUE ? 0u : w % G, tid = threadIdx.x;
    if (slot >= walk[0]) return;
    const uint32_t f = walk[1 + slot];
    const DFrame fr = frames[f];
    BFrame *bf = &bframes[f];
    if (RESCUE && !bf->bail) return;
    const uint32_t nb = min(bf->n_ok, bf->first_bad);  // blocks [0, nb) executed without a defect
    uint8_t *p0 = out_blob + fr.out_offset;
    const uint8_t *p1 = pl1 + fr.out_offset;
    const uint8_t *pE = pl2 + fr.out_offset;                    // the plane of the XOR term
    const uint8_t *pH = NP == 4 ? pl3 + fr.out_offset : nullptr;  // the plane of the high position bits, written for frames with `high`
    const bool high = NP == 4 && bf->high != 0;
    const uint32_t cstep = G * 256;
    uint32_t *fdone = done + (size_t)slot * kFixMaxG;
    __shared__ uint32_t go;
    // the frame's jobs after its first (which derives nothing: its blocks follow each other inside one job), in order.
    // (Loading the NEXT job's extent and plane bytes while this job's step runs -- a software pipeline over the jobs -- made the
    // walk slower, 8.1 -> 8.9 ms for 4 095 steps: a step is the hand-off between the workgroups, not the loads before it.)
    uint32_t steps = 0;  // jobs this workgroup is done with
    for (uint32_t bi = 1; bi < nb;) {
        // (this block's entry and the four after it in one go: the job's extent should not cost a latency per block)
        BJob c[5];
#pragma unroll
        for (int q = 0; q < 5; q++) c[q] = jobs[fr.first_block + min(bi + (uint32_t)q, fr.n_blocks - 1)];
        const BJob jb = c[0];
        if (!(jb.flags & kBjHead)) {  // (a block of the first job)
            bi++;
            continue;
        }
        // the job: blocks [bi, e) -- up to the next head, and not beyond the blocks that executed (k_exec_b<true> stops at a
        // skipped block and at the first block that failed: the blocks of the job before that one ran and are fixed up, as the
        // serial walk would have left them)
        uint32_t e = bi + 1;
        BJob je = c[0];
#pragma unroll
        for (int q = 1; q < 5; q++)
            if (e == bi + (uint32_t)q && e < nb && !(c[q].flags & (kBjHead | kBjSkip))) {
                je = c[q];
                e++;
            }
        if (e == bi + 5)  // (jobs of more than five blocks: experiments only)
            while (e < nb && !(jobs[fr.first_block + e].flags & (kBjHead | kBjSkip))) {
                je = jobs[fr.first_block + e];
                e++;
            }
        bi = e;
        if (!(jb.flags & kBjDirect)) {
            const uint32_t S = jb.start, n = je.start + je.len - jb.start;
            const uint32_t nchunks = n >> 4;
            for (uint32_t g = g0; g < (RESCUE ? G : g0 + 1); g++) {
                if (RESCUE && fdone[g] > steps) continue;  // workgroup g gathered its chunks of this step
                FixChunks<NP> C;
                fix_load<NP>(C, p0, p1, pH, pE, S, n, g * 256 + tid, cstep, high);  // (nothing here was written by this kernel)
                if (!RESCUE && G > 1 && steps > 0) {
                    // every workgroup of the frame is done with the jobs before this one (a bounded wait: all of them are
                    // resident -- the launch is sized for that -- but a hang is not an acceptable failure mode)
                    if (tid == 0) {
                        uint32_t it = 0, ok = 1;
                        const uint32_t target = G * steps;
                        while (__hip_atomic_load(&bf->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                            __builtin_amdgcn_s_sleep(4);
                            if (++it > 4000000u || __hip_atomic_load(&bf->bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                ok = 0;
                                break;
                            }
                        }
                        if (test_bail_step && steps >= test_bail_step && g == 1) ok = 0;  // (mzd_debug_force_fixup_bail: the tests' way into the rescue)
                        go = ok;
                    }
                    __syncthreads();
                    if (!go) {
                        if (tid == 0) __hip_atomic_store(&bf->bail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
                fix_gather<NP>(C, p0, S, high);
                for (uint32_t c0 = g * 256 + tid + kFixK * cstep; c0 < nchunks; c0 += kFixK * cstep) {
                    fix_load<NP>(C, p0, p1, pH, pE, S, n, c0, cstep, high);
                    fix_gather<NP>(C, p0, S, high);
                }
                if (g == 0 && tid < (n & 15)) {  // the job's last bytes
                    const uint32_t x = S + (n & ~15u) + tid;
                    const uint32_t aj = p0[x], dj = aj ^ pE[x];
                    if (dj) {
                        const uint32_t org = fix_origin<NP>(aj, p1[x], dj, NP == 4 && high ? pH[x] : 0u, S, high);
                        *(volatile uint8_t *)(p0 + x) = org < S ? *(const volatile uint8_t *)(p0 + org) : (uint8_t)0;
                    }
                }
            }
        }
        steps++;
        if (RESCUE) {
            xb_wait_vm();
            __syncthreads();  // (one workgroup; its stores write through and its gathers are agent-scope loads)
        } else if (G > 1) {
            xb_wait_vm();  // this thread's stores of the step have reached memory
            __syncthreads();
            if (tid == 0) {
                fdone[g0] = steps;
                __hip_atomic_fetch_add(&bf->cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            __syncthreads();  // (one workgroup: its own stores are visible to it)
        }
    }
    if (RESCUE && tid == 0) bf->bail = 0u;  // the frame is whole: k_blk_final reports what the walk found
}

// ---- status and length of every frame, as the serial walk reports them
__global__ void k_blk_final(const DFrame *__restrict__ frames, const BJob *__restrict__ jobs, const BFrame *__restrict__ bframes,
                            int32_t *frame_status, uint64_t *frame_out_len, uint32_t n_frames)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const DFrame fr = frames[f];
    const BFrame bf = bframes[f];
    int e = bf.status;
    uint32_t len = bf.out_len;
    if (bf.first_bad < bf.n_ok) {
        const BJob jb = jobs[fr.first_block + bf.first_bad];
        e = MZD_ERR_OFFSET;
        len = jb.start + jb.len;  // (the length k_exec reports for a block that failed on an offset)
    }
    if (bf.bail) e = MZD_ERR_DEVICE;
    if (e == MZD_OK && fr.content_size != MZD_UNKNOWN_SIZE && len != fr.content_size) e = MZD_ERR_DST_FULL;
    frame_status[f] = e;
    frame_out_len[f] = len;
}

}  // namespace mzd
