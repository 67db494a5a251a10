// mzd_seq_q4.hip -- k_seq_q4: the FSE sequence decode (structure/sequences.go:126-206) with FOUR LANES PER CHAIN.
//
// Why.  What bounds the sequence decode is the step latency of the chain wavefront: LDS capacity fixes the
// number of chains a CU can hold (their tables), a lone wavefront issues one instruction every ~5-6 cycles
// whatever its kind, so the number of instructions of a step IS the step (k_seq_pipe: ~66 -> ~460 cycles).
// A third of them is the same small computation done three times, once per FSE state (cell address, cell
// read, next / code split, highbit, extra-bit count, state field, new state).  Here the three states of a
// chain sit in three adjacent LANES, so that computation is issued once, and the sums that couple the
// states (bits of the extra fields, running sum of the state fields) are five DPP adds inside the quad:
//   lane 4c + 0: literal-length state   lane 4c + 1: match-length state   lane 4c + 2: offset state
//   lane 4c + 3: the chain's bit window: it alone reads the bitstream ring (a byte-misaligned 8-byte LDS
//                read costs the LDS pipe a cycle per ACTIVE lane) and hands the window to stage B.
// Its own table cell is a constant (next = 1, code 0): zero state bits, zero extra bits -- which is also what
// row_shr:1 brings into lane 4c + 0 from the quad below, so the running sums need no masks.
// The window is no longer kept in registers and refilled: every step reads the 8 bytes at its cursor
// straight from the ring (the read is issued at the end of the step before, behind that step's work).
// 14 chains per chain wavefront, FOUR chain wavefronts per workgroup (56 chains at most; LDS holds 54),
// stage B / C / P as in k_seq_pipe (one lane per chain): 7 wavefronts per workgroup.
//
// Results are k_seq_pipe's, record for record: same cells (next:10 | c6:6), same escapes, same general
// step for the rare cases (window too short, stream end, last sequence, escape), same stage C.
#pragma once

namespace mzd {

constexpr int kQ4ChainsPerWave = 14;  // quads 14 and 15 of a chain wavefront are parked
constexpr int kQ4ChainWaves = 4;
#ifndef MZD_Q4_BWAVES
#define MZD_Q4_BWAVES 3
#endif
constexpr int kQ4BWaves = MZD_Q4_BWAVES;  // stage-B wavefronts: they take the batches of four steps in turn
constexpr int kQ4Threads = 64 * (kQ4ChainWaves + 3 + kQ4BWaves);  // + stage B, C1, C2, P
constexpr int kQ4Cols = 57;  // queue columns: one per chain (<= 56) + the column parked quads write to (56)

struct Q4Shared {
    uint32_t head1[4];                 // steps produced by each chain wavefront (0xFFFFFFFF: has no chains)
    uint32_t tailB[2];                 // steps of ITS batches each stage-B wavefront has consumed (even / odd batches of four)
    uint32_t head2[2];                 // steps each stage-B wavefront has handed to stage C1
    uint32_t head3, tail2, pad0[2];    // steps stage C1 has finished (records complete in q2) / stage C2 has stored
    uint32_t progress[64];             // per chain: byte offset of the window (from in - MZD_IN_PAD), published per batch
    int32_t stC[64];                   // final status of stage C1 (offsets)
    int32_t stC2[64];                  // final status of stage C2 (sizes)
    int32_t stA[64];                   // final status of stage A
    uint32_t ring_low[64];             // per chain: lowest offset wave P has put in the ring
    uint64_t q1w[kPipeDepth][kQ4Cols];      // mode 0: the 8 bytes at the cursor; mode 1: LL:17 | ML:18 | offset value:29
    uint16_t q1c[kPipeDepth][kQ4Cols][4];   // mode 0: LL cell, ML cell, OF cell, bits consumed of q1w (0..7); mode 1: [3] = 0x8000
    uint64_t q2[kPipeDepth][kQ4Cols];       // B -> C1: LL:17 | ML:18 | offset value:29; C1 -> C2: the finished record, in place
    uint8_t ring[kQ4ChainWaves * kQ4ChainsPerWave][kPipeRing + 8];
    uint16_t dummy[8];                 // dummy[1] = 1: the cell of parked lanes and of every quad's fourth lane
};
constexpr int kQ4FixedLds = (512 + (int)sizeof(Q4Shared) + 15) & ~15;
constexpr int kQ4MaxChains = (160 * 1024 - kQ4FixedLds) / (kSeqCellsPerChain * 2);
constexpr int kQ4Chains = 56;        // chains per workgroup: every quad of the four chain wavefronts; all of a CU's LDS with full-size tables
constexpr int kQ4ChainsBeside = 54;  // ... when the Huffman kernel runs beside this one: ~5 KiB of LDS stay free for its workgroups
static_assert(kQ4MaxChains >= kQ4Chains && kQ4Chains <= kQ4Cols - 1 && kQ4Chains <= kQ4ChainWaves * kQ4ChainsPerWave &&
                  offsetof(Q4Shared, ring) % 8 == 0 && offsetof(Q4Shared, q1w) % 8 == 0, "k_seq_q4 LDS layout");

#ifdef MZD_Q4_STATS
__device__ unsigned long long g_q4_stats[8];  // chain wavefronts, steps, cycles of stage A, queue-full polls, ring polls, general steps, cycles in them, chains in general steps by reason (last sequence | escape cell << 20 | bits << 40)
#endif
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v)  // quad_perm, all rows and banks, out-of-range lanes read 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

// What stage B hands stage C1 in a record's offset field: the repeat case already decided from the two things it depends
// on (offset value, literal length == 0; sequence_execution.go:65-114) -- 1..4 = the history cases of q4_history_step,
// >= 5 = a new offset of (code - 4).  Stage C1 is ONE wavefront walking every chain's history, an instruction every ~6
// cycles: whatever does not need the history is computed a stage earlier.
__device__ __forceinline__ uint32_t q4_offset_code(uint32_t ofv, uint32_t ll) { return ofv + ((ofv > 3 || ll == 0) ? 1u : 0u); }
constexpr uint32_t kQ4SymBase = (kRecOffSymbolic << 3) - 8u;  // (symbolic | (-off - 1)) << 3 = kQ4SymBase - (off << 3)

// One sequence of the repeat-offset history for every chain of the wavefront (lane = chain).  `hi` = high dword of the
// record as stage B left it (ML bits 15..17 | offset code << 3); h0..h2 = the history, kept SHIFTED LEFT BY 3 like the
// record's field (symbolic values of a block whose history is unknown are negative: -8, -16, -24); returns the finished
// high dword.  maxv / minv collect the two error conditions (code too large; zero offset).  Hand-scheduled: the
// compiler's version of the same selects puts every compare in an SGPR pair right before its use, and gfx950 wants two
// wait states between a VALU write of a mask and the VALU read of it (an s_nop per select); here the five compares on
// the code are issued together and none of the 26 instructions waits.
__device__ __forceinline__ uint32_t q4_history_step(uint32_t hi, uint32_t &h0, uint32_t &h1, uint32_t &h2, uint32_t &maxv,
                                                    uint32_t &minv)
{
    uint32_t rh, idx, nw, t, off;
    unsigned long long m0, m1, m2, m3, m4;
    asm volatile(
        "v_lshrrev_b32 %[idx], 3, %[hi]\n\t"
        "v_and_b32 %[nw], -8, %[hi]\n\t"
        "v_cmp_lt_i32 vcc, 0, %[h0]\n\t"
        "v_min_u32 %[idx], 5, %[idx]\n\t"              // 1..4: history cases, 5: new offset
        "v_add_u32 %[nw], -32, %[nw]\n\t"              // new offset = code - 4
        "v_cndmask_b32_e64 %[t], 32, 8, vcc\n\t"
        "v_cmp_eq_u32_e64 %[m0], 4, %[idx]\n\t"
        "v_cmp_eq_u32_e64 %[m1], 3, %[idx]\n\t"
        "v_cmp_eq_u32_e64 %[m2], 2, %[idx]\n\t"
        "v_cmp_gt_u32_e64 %[m3], 2, %[idx]\n\t"
        "v_cmp_le_u32_e64 %[m4], 3, %[idx]\n\t"
        "v_sub_u32 %[t], %[h0], %[t]\n\t"              // hist_dec(h0): h0 - 1, or h0 - 4 when symbolic
        "v_cndmask_b32_e64 %[off], %[nw], %[t], %[m0]\n\t"
        "v_cndmask_b32_e64 %[off], %[off], %[h2], %[m1]\n\t"
        "v_cndmask_b32_e64 %[off], %[off], %[h1], %[m2]\n\t"
        "v_cndmask_b32_e64 %[off], %[off], %[h0], %[m3]\n\t"
        "v_cmp_lt_i32 vcc, 0, %[off]\n\t"
        "v_sub_u32 %[t], %[symbase], %[off]\n\t"
        "v_cndmask_b32_e64 %[h2], %[h2], %[h1], %[m4]\n\t"
        "v_cndmask_b32_e64 %[h1], %[h0], %[h1], %[m3]\n\t"
        "v_cndmask_b32_e64 %[h0], %[off], %[h0], %[m3]\n\t"
        "v_cndmask_b32_e32 %[t], %[t], %[off], vcc\n\t"
        "v_max_u32 %[maxv], %[maxv], %[hi]\n\t"
        "v_min_u32 %[minv], %[minv], %[off]\n\t"
        "v_bfi_b32 %[rh], 7, %[hi], %[t]"
        : [rh] "=&v"(rh), [idx] "=&v"(idx), [nw] "=&v"(nw), [t] "=&v"(t), [off] "=&v"(off), [m0] "=&s"(m0), [m1] "=&s"(m1),
          [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [maxv] "+v"(maxv),
          [minv] "+v"(minv)
        : [hi] "v"(hi), [symbase] "s"(kQ4SymBase)
        : "vcc");
    return rh;
}

__global__ __launch_bounds__(kQ4Threads) void k_seq_q4(const uint8_t *__restrict__ in, const SeqTask *__restrict__ tasks,
                                                       uint32_t n_tasks, const uint32_t *__restrict__ fse_entries,
                                                       uint64_t *__restrict__ recs, TileBase *__restrict__ tiles,
                                                       BlockSum *sums, uint32_t nch, uint64_t in_base, uint32_t cells_ll,
                                                       uint32_t cells_ml, uint32_t cells_of)
{
    // cells_ll / cells_ml / cells_of: the largest LL / ML / OF table of the batch (powers of two).  A chain's slot in LDS is
    // exactly that wide -- 1280 cells when the tables have the format's largest accuracy logs (9 / 9 / 8), 160 when they
    // are the predefined ones -- so batches of small tables leave room for two or three workgroups per CU.
    const uint32_t off_ml = cells_ll, off_of = cells_ll + cells_ml, slot_cells = cells_ll + cells_ml + cells_of;
#ifdef MZD_SHIFT_Q4  /* experiment: the whole instruction stream four bytes later (code-placement sensitivity of hand-written streams) */
    asm volatile("s_nop 0");
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef MZD_Q4_PROF
    const long long prof_w0 = wall_clock64(), prof_c0 = clock64();  // 100 MHz / shader clock
#endif
    uint32_t *CTc = (uint32_t *)smem;  // [2][64] by c6: base(24) | extra(8)   (predefined.go:5-20,36-50)
    Q4Shared *shs = (Q4Shared *)(smem + 512);
    uint16_t *cells = (uint16_t *)(smem + kQ4FixedLds);
    const int lane = threadIdx.x & 63;
    // Which stage wavefronts 4..8 run (logical ids 4 = B even batches, 5 = B odd, 6 = C1, 7 = C2, 8 = P).  Wavefront w of
    // a workgroup runs on SIMD w % 4 beside chain wavefront w % 4, and the four chain wavefronts advance in lockstep (they
    // share the queue), so the busiest SIMD sets everybody's step.  Measured per step: P beside chain wavefront 0 and C2 as
    // the third wavefront of that SIMD, B / B / C1 beside the others: 307 cycles; in the order of the stages: 316.
#ifndef MZD_Q4_ROLES
#define MZD_Q4_ROLES {8, 4, 5, 6, 7, 9, 10}
#endif
    const int wave = threadIdx.x >> 6;  // 0..3: chain wavefronts; 4..8: the stages, see above
    constexpr int kRoles[7] = MZD_Q4_ROLES;  // (9, 10: the third and fourth stage-B wavefronts, when there are such; four: no better than three)
    const int lw = wave < kQ4ChainWaves ? wave : kRoles[wave - kQ4ChainWaves];
    const bool chainw = wave < kQ4ChainWaves;
    // the chain this lane works for: chain wavefronts 4 lanes per chain, the others one lane per chain
    const uint32_t quad = (uint32_t)lane >> 2, role = (uint32_t)lane & 3u;
    // (quads 14 and 15 of a chain wavefront have no chain: they are parked and write their queue entries to column 55,
    // which no workgroup uses: nch <= 54)
    const uint32_t ch = chainw ? (quad < (uint32_t)kQ4ChainsPerWave ? (uint32_t)wave * kQ4ChainsPerWave + quad : (uint32_t)kQ4Cols - 1u) : (uint32_t)lane;
    const uint32_t tid = blockIdx.x * nch + ch;
    const bool has = ch < nch && tid < n_tasks;
    // in_base == ~0 (round 6): the window is the WORKGROUP's -- from its first chain's bitstream on: the chains of a workgroup are
    // consecutive blocks of a list in frame order, a few MB apart at most, wherever in a blob of any size they lie (what used to send
    // a frame whose bitstreams span more than 4 GiB to the older kernel)
    if (in_base == ~0ull) in_base = tasks[min(blockIdx.x * nch, n_tasks - 1)].in_off;
    SeqTask t;
    if (has) {
        t = tasks[tid];
        t.in_off -= in_base;  // the launch's window of the blob: bitstreams are addressed with 32-bit offsets from it
    } else {
        t.n_seq = 0; t.in_size = 0; t.ll_off = t.of_off = t.ml_off = 0; t.ll_log = t.of_log = t.ml_log = 0;
        t.in_off = 0; t.rec_off = 0; t.tile_off = 0; t.block = 0; t.hist_known = 0;
    }
    in += in_base;
    if (lw == 8) {
        CTc[lane] = 0;
        CTc[64 + lane] = 0;
        shs->progress[lane] = (uint32_t)t.in_off + MZD_IN_PAD + t.in_size;
        shs->ring_low[lane] = (has && t.n_seq > 0) ? 0xFFFFFFFFu : 0u;  // nothing in the ring yet / nothing needed
        shs->stC[lane] = MZD_OK;
        shs->stC2[lane] = MZD_OK;
        shs->stA[lane] = MZD_OK;
        if (lane < 8) shs->dummy[lane] = 1;
        if (lane == 0) { shs->tailB[0] = 0; shs->tailB[1] = 0; shs->head2[0] = 0; shs->head2[1] = 0; shs->head3 = 0; shs->tail2 = 0; }
        // a chain wavefront without chains never produces anything: nobody waits for it
        if (lane < 4) shs->head1[lane] = (uint32_t)lane * kQ4ChainsPerWave < min(nch, n_tasks - blockIdx.x * nch) ? 0u : 0xFFFFFFFFu;
        __builtin_amdgcn_s_waitcnt(0);  // the zero fill above before the scattered fill below (same wavefront: LDS is in order)
        if (lane < 36 && seq_code6(0, lane) != kPipeEscape) CTc[seq_code6(0, lane)] = c_ll_base[lane] | ((uint32_t)c_ll_extra[lane] << 24);
        if (lane < 53 && seq_code6(1, lane) != kPipeEscape) CTc[64 + seq_code6(1, lane)] = c_ml_base[lane] | ((uint32_t)c_ml_extra[lane] << 24);
    }
    // ---- stage the three tables of every chain of this workgroup
    {
        uint32_t *desc = (uint32_t *)&shs->q1w[0][0];  // [chain][4]: ll_off, ml_off, of_off, logs
        if (lw == 4) {
            desc[4 * lane + 0] = t.ll_off;
            desc[4 * lane + 1] = t.ml_off;
            desc[4 * lane + 2] = t.of_off;
            desc[4 * lane + 3] = has ? ((uint32_t)t.ll_log | ((uint32_t)t.ml_log << 8) | ((uint32_t)t.of_log << 16)) : 0x00FFFFFFu;
        }
        __syncthreads();
        // A wavefront per chain (round robin), its lanes over the cells the chain's three tables REALLY have: a chain with
        // predefined or small tables costs what it has, not the batch's largest slot (real data: 40 us of staging per
        // workgroup for chains of a few hundred steps), and nothing divides by the slot size.
        const uint32_t nchw = min(nch, n_tasks - blockIdx.x * nch);
        constexpr int UNR = 8;
        for (uint32_t c = (uint32_t)wave; c < nchw; c += kQ4Threads / 64) {
            const uint32_t lg3 = desc[4 * c + 3];
            const uint32_t lgL = lg3 & 0xFF, lgM = (lg3 >> 8) & 0xFF, lgO = (lg3 >> 16) & 0xFF;
            const uint32_t nL = lgL <= 9 ? 1u << lgL : 0u, nM = lgM <= 9 ? 1u << lgM : 0u, nO = lgO <= 9 ? 1u << lgO : 0u;
            const uint32_t oL = desc[4 * c + 0], oM = desc[4 * c + 1], oO = desc[4 * c + 2];
            const uint32_t tot = nL + nM + nO, base = c * slot_cells;
            for (uint32_t i0 = (uint32_t)lane; i0 < tot; i0 += 64 * UNR) {
                uint32_t e[UNR], n[UNR], c6k[UNR], dst[UNR];
                bool ok[UNR];
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    const uint32_t i = i0 + 64u * u;
                    const uint32_t kind = i >= nL + nM ? 2u : (i >= nL ? 1u : 0u);
                    const uint32_t j = i - (kind == 2 ? nL + nM : (kind == 1 ? nL : 0u));
                    n[u] = kind == 2 ? nO : (kind == 1 ? nM : nL);
                    ok[u] = i < tot;
                    c6k[u] = kind;
                    dst[u] = base + (kind == 2 ? off_of : (kind == 1 ? off_ml : 0u)) + j;
                    e[u] = ok[u] ? fse_entries[(kind == 2 ? oO : (kind == 1 ? oM : oL)) + j] : 0u;  // baseline(16) | nbits(8) | symbol(8)
                }
#pragma unroll
                for (int u = 0; u < UNR; u++) {
                    const uint32_t baseline = e[u] & 0xFFFF, nb = (e[u] >> 16) & 0xFF, sym = e[u] >> 24;
                    const uint32_t c6 = c6k[u] == 2 ? sym : seq_code6((int)c6k[u], sym);
                    if (ok[u]) cells[dst[u]] = c6 == kPipeEscape ? (uint16_t)0 : (uint16_t)(((baseline + n[u]) >> (nb & 31)) | (c6 << 10));
                }
            }
        }
    }
    __syncthreads();

#ifdef MZD_Q4_PROF
    const long long prof_w1 = wall_clock64();
#endif
    // trip count of every stage: the longest chain of the workgroup (the q2 descriptors are gone: ask every lane of B)
    uint32_t nmax;
    {
        uint32_t *nm = (uint32_t *)&shs->q2[0][0];
        if (lw == 4) {
            const uint32_t m = wave_max_u32(has ? t.n_seq : 0u);
            if (lane == 0) nm[0] = m;
        }
        __syncthreads();
        nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nm[0]);
        __syncthreads();
    }

    if (chainw) {
        // ================= stage A: four lanes per chain =================
        const uint8_t *inb = in - MZD_IN_PAD;
        const int alL = t.ll_log, alM = t.ml_log, alO = t.of_log;
        const uint32_t sizeL = 1u << alL, sizeM = 1u << alM, sizeO = 1u << alO;
        const uint32_t slot = ch * slot_cells;
        const bool spare = role == 3;
        const uint32_t dummy_addr = 512u + (uint32_t)offsetof(Q4Shared, dummy);  // byte address of dummy[0]; dummy[1] is the cell
        // ---- per chain (its fourth lane): padding, initial states in the order LL, OF, ML (sequences.go:133-159)
        uint64_t C = 0, D = 0;
        uint32_t off = 0;  // general step only: offset of D's bytes
        int k = 0, rem = 0;
        int status = MZD_OK;
        auto refill = [&]() {
            const int nb = k >> 3, sh = nb * 8;
            C = (C << sh) | ((D >> 1) >> (63 - sh));
            off -= (uint32_t)nb;
            k &= 7;
            asm volatile("" ::"v"((uint32_t)C), "v"((uint32_t)(C >> 32)) : "memory");  // see SeqBits::refill
            D = ld64u(inb + off);
        };
        auto peek = [&](int n) -> uint32_t { return (uint32_t)(((C << k) >> 1) >> (63 - n)); };
        // the same window fed from the chain's LDS ring (general step: wave P keeps the bytes around the cursor there; a
        // global load per refill was most of a general step's ~1500 cycles, and the other chain wavefronts wait for it)
        auto ring64 = [&](uint32_t u) -> uint64_t { return ((const U64U *)(shs->ring[ch] + (u & (uint32_t)(kPipeRing - 1))))->v; };
        auto refill_ring = [&]() {
            const int nb = k >> 3, sh = nb * 8;
            C = (C << sh) | ((D >> 1) >> (63 - sh));
            off -= (uint32_t)nb;
            k &= 7;
            D = ring64(off);
        };
        uint32_t sL = 0, sM = 0, sO = 0;
        bool live = has && t.n_seq > 0;
        if (live && spare) {
            SeqBits br;
            rem = br.init(in + t.in_off, (int)t.in_size);
            C = br.C; D = br.D; k = br.k; off = (uint32_t)(br.pd - inb);
            if (rem < 0) {
                status = MZD_ERR_BAD_PADDING;  // sequences.go:141-143
            } else {
                sL = peek(alL); k += alL;
                sO = peek(alO); k += alO;
                refill();
                sM = peek(alM); k += alM;
                rem -= alL + alO + alM;
                if (rem < 0) status = MZD_ERR_SEQ_BITS;
            }
        }
        // the hot loop's view of the cursor: woff = offset of the 8 bytes that hold the next bit (bit 63 - wk of them)
        uint32_t woff = off + 8u;
        uint32_t wk = (uint32_t)k;
        uint32_t rem1 = (uint32_t)rem + 1u;
        uint32_t last_i = t.n_seq - 1;
        live = live && (bool)dpp_quad<QP(3, 3, 3, 3)>((uint32_t)(status == MZD_OK));
        // ---- hand the chain's state to its four lanes
        // (within a chain's four lanes: DPP quad permutes, one instruction each; a __shfl is an LDS round trip, and ten of
        // them were two thirds of a general step)
        auto from_spare = [&](uint32_t v) -> uint32_t { return dpp_quad<QP(3, 3, 3, 3)>(v); };
        uint32_t st;  // this lane's FSE state, pre-biased by the table size
        uint32_t cb, shr, Kc, nbK;
        {
            const uint32_t l = from_spare(sL) + sizeL, m = from_spare(sM) + sizeM, o = from_spare(sO) + sizeO;
            st = role == 0 ? l : (role == 1 ? m : (role == 2 ? o : 1u));
            const uint32_t base = (uint32_t)kQ4FixedLds;
            cb = role == 0 ? base + 2u * (slot - sizeL)
                           : (role == 1 ? base + 2u * (slot + off_ml - sizeM) : (role == 2 ? base + 2u * (slot + off_of - sizeO) : dummy_addr));
            shr = role == 2 ? 10u : 12u;
            Kc = role == 0 ? 3u : (role == 1 ? 7u : 0u);
            nbK = role == 0 ? 31u - (uint32_t)alL : (role == 1 ? 31u - (uint32_t)alM : (role == 2 ? 31u - (uint32_t)alO : 31u));
            woff = from_spare(woff);
            wk = from_spare(wk);
            rem1 = from_spare(rem1);
        }
        // steps before the chain's last sequence, TIMES 64: a term of the step's bit limit that must not bind before it is 0 (the last
        // sequence takes the general step: no state update, sequences.go:178).  Counted in plain steps it bound as soon as it fell
        // below a step's ~20 bits: the last 20-57 steps of EVERY chain took the general step (1 290 cycles for the whole wavefront,
        // 119 of the 133 general steps per wavefront: 4 % of the kernel; `-DMZD_Q4_STATS_REASONS`)
        uint32_t left = last_i << 6;
        uint32_t W0 = (woff + 127u) & ~127u;  // the hot loop counts the cursor in bits below this offset (see Q4_STEP)
        auto park = [&]() {  // the whole quad: constant cell, cursor 0 (the readable front slack; the ring check is always true for it)
            st = 1; cb = dummy_addr; nbK = 31; woff = 0; wk = 0; W0 = 0; rem1 = 0x7FFFFFFFu; left = 0x7FFFFFC0u; live = false;
        };
        if (!live) park();
        const uint32_t ringl128 = 512u + (uint32_t)offsetof(Q4Shared, ring) + ch * (kPipeRing + 8) + (uint32_t)kPipeRing;

        auto full_cell = [&](int kind, uint32_t x, uint32_t idx, uint32_t toff, uint32_t size, uint32_t &next, uint32_t &ct) {
            next = x & 1023;
            ct = CTc[kind * 64 + (x >> 10)];
            if (next == 0) {  // escape: the symbol is only in the host cell
#ifdef MZD_ABL_Q4_NOESC  /* ablation: timing experiment only, wrong results */
                const uint32_t e = 0x2D020000u + toff * 0u + idx * 0u;
#else
                const uint32_t e = fse_entries[toff + idx];
#endif
                const uint32_t sym = e >> 24;
                next = ((e & 0xFFFF) + size) >> ((e >> 16) & 0xFF);
                ct = kind == 0 ? (c_ll_base[min(sym, 35u)] | ((uint32_t)c_ll_extra[min(sym, 35u)] << 24))
                               : (c_ml_base[min(sym, 52u)] | ((uint32_t)c_ml_extra[min(sym, 52u)] << 24));
            }
        };
        const uint16_t *cL = cells + slot - sizeL;
        const uint16_t *cM = cells + slot + off_ml - sizeM;
        const uint16_t *cO = cells + slot + off_of - sizeO;
        // General step of sequence `idx` for the chains in `mine` (their queue entries of this step are rewritten as
        // mode 1), run by the chain's fourth lane on the chain's three states, as k_seq_pipe's.
        auto general_step = [&](uint32_t idx, bool mine) {
            uint32_t gL = dpp_quad<QP(0, 0, 0, 0)>(st), gM = dpp_quad<QP(1, 1, 1, 1)>(st), gO = dpp_quad<QP(2, 2, 2, 2)>(st);
            bool parkq = false;
            if (mine && spare) {
                const bool lastseq = idx == last_i;
                // the window as the general step keeps it: C = the 8 bytes at the cursor, D = the 8 below.  The step reads
                // down to 22 bytes below the cursor (two refills of up to 7 bytes); a batch's ring check only promises
                // 40 - 28: tell wave P where the cursor is and wait until its ring reaches 24 below (it keeps 88)
                __hip_atomic_store(&shs->progress[ch], woff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(&shs->ring_low[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > woff - 24u)
                    __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                off = woff - 8u;
                k = (int)wk;
                C = ring64(woff);
                D = ring64(off);
                const uint32_t xl = cL[gL], xm = cM[gM], xo = cO[gO];
#ifdef MZD_Q4_STATS_REASONS  /* (an atomic per chain: distorts the timing of the build) */
                atomicAdd(&g_q4_stats[7], lastseq ? 1ull : (((xl & 1023) == 0 || (xm & 1023) == 0) ? (1ull << 20) : (1ull << 40)));
#endif
                uint32_t nl = 1, nm = 1, cl = 0, cm = 0;
                full_cell(0, xl, gL - sizeL, t.ll_off, sizeL, nl, cl);
                full_cell(1, xm, gM - sizeM, t.ml_off, sizeM, nm, cm);
                const uint32_t no = xo & 1023, exO = xo >> 10;
                uint32_t nbL = (uint32_t)(alL - 31) + (uint32_t)__builtin_clz(nl | 1);
                uint32_t nbM = (uint32_t)(alM - 31) + (uint32_t)__builtin_clz(nm | 1);
                uint32_t nbO = (uint32_t)(alO - 31) + (uint32_t)__builtin_clz(no | 1);
                if (lastseq) { nbL = 0; nbM = 0; nbO = 0; }  // sequences.go:178
                const uint32_t exL = cl >> 24, exM = cm >> 24;
                const int total = (int)(exO + exM + exL + nbL + nbM + nbO);
                int remi = (int)(rem1 - 1u);
                bool ok = true;
                if (total > remi) {  // the cursor would pass the start of the stream
                    status = MZD_ERR_SEQ_BITS;
                    ok = false;
                }
                if (ok) {
                    const uint32_t ofx = peek((int)exO); k += (int)exO; refill_ring();
                    const uint32_t mlx = peek((int)exM); k += (int)exM;
                    const uint32_t llx = peek((int)exL); k += (int)exL; refill_ring();
                    const uint32_t aL = peek((int)nbL); k += (int)nbL;
                    const uint32_t aM = peek((int)nbM); k += (int)nbM;
                    const uint32_t aO = peek((int)nbO); k += (int)nbO;
                    remi -= total;
                    gL = (nl << nbL) + aL; gM = (nm << nbM) + aM; gO = (no << nbO) + aO;  // fse.go:282-290
                    const uint32_t ofv = min((1u << exO) + ofx, kRecOffSymbolic);  // exO <= 31: no wrap
                    const uint32_t llv = (cl & 0xFFFFFF) + llx;
                    shs->q1w[idx % kPipeDepth][ch] = (uint64_t)llv | ((uint64_t)((cm & 0xFFFFFF) + mlx) << kRecMlShift) |
                                                     ((uint64_t)q4_offset_code(ofv, llv) << kRecOffShift);
                    shs->q1c[idx % kPipeDepth][ch][3] = 0x8000u;
                    woff = off + 8u;
                    wk = (uint32_t)k;
                    rem1 = (uint32_t)remi + 1u;
                }
                if (lastseq || !ok) {
                    if (ok && remi != 0) status = MZD_ERR_SEQ_BITS;  // sequences.go:197-204
                    parkq = true;
                }
            }
            // back to the chain's four lanes (the cursor normalised: wk < 8)
            const bool minech = (bool)from_spare((uint32_t)(mine && spare));
            const uint32_t nL = from_spare(gL), nM = from_spare(gM), nO = from_spare(gO);
            const uint32_t nwoff = from_spare(woff), nwk = from_spare(wk), nrem1 = from_spare(rem1);
            if (minech) {
                st = role == 0 ? nL : (role == 1 ? nM : (role == 2 ? nO : 1u));
                woff = nwoff - (nwk >> 3);
                wk = nwk & 7u;
                rem1 = nrem1;
            }
            if ((bool)from_spare((uint32_t)parkq)) park();
        };

        // ---- the hot loop.  The cursor is normalised at the END of a step (wk < 8 on entry).  Runs steps until a chain
        // needs the general step (returns the mask of its lanes; its step is NOT done, everybody's queue entry IS
        // written, head1 not yet moved) or nmax is reached.
        uint32_t i = 0;
        {
            const uint32_t nb = wk >> 3;
            woff -= nb;
            wk &= 7;
        }
        const uint64_t sparemask = 0x8888888888888888ull;
        const uint32_t vzero = 0;
        const uint32_t qca = (ch * 4u + role) * 2u, qwa = ch * 8u, chan4 = ch * 4u;  // ch <= 55
        const uint32_t heada = 512u + (uint32_t)offsetof(Q4Shared, head1) + 4u * (uint32_t)wave;
        uint32_t tail0 = 0, tail1 = 0;  // what this wavefront last saw of the two stage-B wavefronts' progress
        uint32_t polls = 0;
#ifdef MZD_Q4_STATS
        const long long stats_t0 = clock64();
        uint32_t n_general = 0;
        long long general_cycles = 0;
#endif
        // Steps [0, ncom) come before every chain's last sequence: "steps before the last" (times 64) cannot be the smallest term of a
        // step's limit (64 - k is at most 64) and the loop variant that runs them leaves it out.
        const uint32_t min_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(live ? last_i : 0xFFFFFFFFu));
        const uint32_t ncom = min(min_last & ~3u, nmax & ~3u);
        while (i < nmax) {
            uint64_t smask = 0;
            {
                // The step, hand-scheduled (~41 instructions; k_seq_pipe's: ~66; a lone wavefront issues one instruction
                // per ~6 cycles whatever its kind, so the count IS the step).  The loop body is the step EIGHT times, one
                // instance per queue slot, as in k_seq_pipe: slot addresses are immediates, queue space / the ring / the
                // cursor for wave P are dealt with once per batch of four steps, the step counter moves, head1 is
                // published and the end checked at the end of a batch (a step that leaves the loop adds its own position
                // in the batch).  A step requests the NEXT step's cell as soon as it has the new state (speculatively: a
                // chain that does not "go" reads with a meaningless state; LDS reads outside the allocation return zero)
                // and does its bookkeeping behind that read.  The window is read and handed to stage B by the fourth lanes only
                // (exec switched twice per step: letting all four lanes of a chain do it -- same data, same queue address,
                // the other three reading an aligned address -- saves two instructions and was 40 % SLOWER: the LDS pipe
                // is too full for 64-lane 8-byte accesses).  DPP reads of a VGPR keep
                // two instructions' distance from the VALU write of it (the hardware does not interlock that; nothing
                // inside an asm statement is padded by the compiler).  Temporaries are fixed registers v66..v89 / s86.
                static_assert(kPipeDepth == 8 && kPipeBatch == 4 && kPipeRing == 128, "the unrolled loop assumes 2 batches of 4 slots, a 128-byte ring");
                uint32_t stb = st;  // the states alternate between two register sets
                // (wave-uniform by construction; said explicitly, or the "s" operands below are refused)
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                tail0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail0);
                tail1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail1);
                polls = (uint32_t)__builtin_amdgcn_readfirstlane((int)polls);
                if (live) left = (last_i - i) << 6;
#ifdef MZD_ABL_Q4_NOW1  /* ablations: timing experiments only, wrong results */
#define MZD_Q4_W1 "s_nop 0\n\t"
#else
#define MZD_Q4_W1 "s_waitcnt lgkmcnt(3)\n\t"
#endif
#ifdef MZD_ABL_Q4_NOW2
#define MZD_Q4_W2 "s_nop 0\n\t"
#else
#define MZD_Q4_W2 "s_waitcnt lgkmcnt(2)\n\t"
#endif
#ifdef MZD_ABL_Q4_ALIGNED  /* ablations: timing experiments only, wrong results */
#define MZD_Q4_RMASK "120"
#else
#define MZD_Q4_RMASK "127"
#endif
#ifdef MZD_ABL_Q4_NOQW
#define MZD_Q4_QWR(X) "s_nop 0\n\t"
#else
#define MZD_Q4_QWR(X) X
#endif
// (v70 = the window's byte offset, W0 - (c >> 3): the batch checks are its only users, Q4_OFF computes it)
#define Q4_OFF                                                                                              \
    "v_lshrrev_b32 v70, 3, %[c]\n\t"                                                                       \
    "v_sub_u32 v70, %[w0], v70\n\t"
#define Q4_RINGCHK(TAG)                                                                                     \
    "ds_write_b32 %[chan4], v70 offset:%[o_prog]\n"                                                        \
    "L_q4_ring" TAG "_%=:\n\t"                                                                              \
    "ds_read_b32 v66, %[chan4] offset:%[o_rlow]\n\t"                                                       \
    "v_add_u32 v67, -40, v70\n\t"                                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
    "v_cmp_gt_u32 vcc, v66, v67\n\t"                                                                      \
    "s_cbranch_vccz L_q4_go" TAG "_%=\n\t"                                                                  \
    "s_add_u32 %[polls], %[polls], 0x10000\n\t"                                                             \
    "s_sleep 1\n\t"                                                                                         \
    "s_branch L_q4_ring" TAG "_%=\n"
// TL / OTL: the progress of the stage-B wavefront that owns the four slots this batch goes to
#define Q4_CHECK(TAG, TL, OTL)                                                                              \
    "L_q4_top" TAG "_%=:\n\t"                                                                               \
    "s_sub_u32 s86, %[i], " TL "\n\t"                                                                       \
    "s_cmp_lt_u32 s86, 5\n\t" /* the batch that used these slots before (i - 8) has been consumed */        \
    "s_cbranch_scc1 L_q4_spc" TAG "_%=\n"                                                                   \
    "L_q4_poll" TAG "_%=:\n\t"                                                                              \
    "ds_read_b32 v66, %[vzero] offset:" OTL "\n\t"                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
    "v_readfirstlane_b32 " TL ", v66\n\t"                                                                  \
    "s_sub_u32 s86, %[i], " TL "\n\t"                                                                       \
    "s_cmp_lt_u32 s86, 5\n\t"                                                                               \
    "s_cbranch_scc1 L_q4_spc" TAG "_%=\n\t"                                                                 \
    "s_add_u32 %[polls], %[polls], 1\n\t"                                                                   \
    "s_sleep 1\n\t"                                                                                         \
    "s_branch L_q4_poll" TAG "_%=\n"                                                                        \
    "L_q4_spc" TAG "_%=:\n\t"                                                                               \
    /* fast path: ring_low as read during the previous step (v89; it only ever decreases) */               \
    Q4_OFF                                                                                                  \
    "v_add_u32 v67, -40, v70\n\t"                                                                          \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
    "v_cmp_gt_u32 vcc, v89, v67\n\t"                                                                      \
    "ds_write_b32 %[chan4], v70 offset:%[o_prog]\n\t"                                                       \
    "s_cbranch_vccz L_q4_go" TAG "_%=\n\t"                                                                  \
    Q4_RINGCHK(TAG)
// On entry: this step's cell is on its way into CELL (requested by the step before, or the prologue), DM gets the
// 8 bytes at the cursor (read at the end of the step before), v85 holds the limit.  DL: where the next step's window
// goes.  LIMIT: the instruction(s) that finish the next step's limit in v85 (from 64 - k).
// Round 5 (39 -> 35 instructions): the cursor is ONE running bit count c (c >> 3 bytes below W0, a multiple of 128: the ring index
// is 128 - ((c >> 3) & 127), the ring's 8 spare bytes making 128 as good as 0) instead of a byte offset and a bit count that are
// normalised every step; the shifted window X is computed by the fourth lane alone, from the registers its LDS read filled, and ONE
// DPP move hands its high dword to the quad (there were two to hand out the window itself); the cursor and the remaining-bits
// count move by `total` unconditionally -- a chain that does not "go" leaves the loop with its wavefront right after this step
// and the exit path takes the step back; the compare leaves the chains that do NOT go in vcc, so that the branch out reads it
// as it is.  The bits that remain are R0 - c: part of the limit only in the loop variant for the chains' last steps (LIMIT) --
// a chain that runs out of bits earlier (a damaged stream: its status is an error whatever it decodes) is caught when the
// loop is left.
#define Q4_STEP(DM, DL, CELL, NCELL, SA, SB, TAG, QC, QW, OUT, RLOW, LIMIT)                                 \
    "L_q4_go" TAG "_%=:\n\t"                                                                                \
    MZD_Q4_W1                                       /* the cell (behind it: the ring read, two queue writes) */ \
    /* What a step costs is what stands between a cell's arrival and the request for the next one (round 5: 20 -> 17 instructions;  \
       whatever can wait -- the total, the go test, the queue entry, the cursor -- runs behind that read), in an order that   \
       gives every DPP read its two instructions' distance from the write of its source without fillers. */                 \
    "v_lshrrev_b32 v76, %[shr], " CELL "\n\t"      /* code field */                                        \
    "v_sub_u32_e64 v76, v76, %[Kc] clamp\n\t"     /* ex */                                                \
    "v_and_b32 v77, 0x3ff, " CELL "\n\t"           /* next */                                              \
    "v_ffbh_u32 v78, v77\n\t"                                                                             \
    "v_add_u32_dpp v79, v76, v76 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                     \
    "v_min_u32 v78, 0x4000000, v78\n\t"           /* escape (next = 0): capped, the sums cannot wrap */   \
    "v_sub_u32 v78, v78, %[nbK]\n\t"              /* nb */                                                \
    "v_add_u32_dpp v81, v79, v79 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" /* o3 */            \
    "v_add_u32 v86, v81, %[k]\n\t"                /* k + o3 */                                            \
    MZD_Q4_W2                                       /* the window */                                        \
    "v_lshlrev_b64 v[74:75], v86, " DM "\n\t"    /* X = W << (k + o3): the state fields from bit 63 (the fourth lane's is the real one) */ \
    "v_add_u32_dpp v80, v78, v78 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" /* nb + nb[lane - 1] */ \
    "v_add_u32_dpp v82, v78, v80 quad_perm:[3,3,0,3] row_mask:0xf bank_mask:0xf\n\t" /* P: running sum of nb */ \
    "v_mov_b32_dpp v84, v75 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t" /* the fourth lane's X, high dword */ \
    "v_sub_u32 v83, 0, v82\n\t"                   /* -P */                                                \
    "v_bfe_u32 v84, v84, v83, v78\n\t"          /* the lane's state field */                            \
    "v_lshl_add_u32 %[s" SB "], v77, v78, v84\n\t" /* new state, in the OTHER register set */            \
    "v_lshl_add_u32 v66, %[s" SB "], 1, %[cb]\n\t"                                                         \
    "ds_read_u16 " NCELL ", v66\n\t"              /* the NEXT step's cell (the cells alternate between two registers: this step's is   \
                                                       still read below); the rest of the step runs behind the read */ \
    "v_add_u32_dpp v87, v82, v81 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t" /* total */         \
    "v_cmp_ge_u32 vcc, v87, v85\n\t"              /* NOT go: the chain needs the general step (the same in its four lanes) */ \
    "v_add_u32 %[c], %[c], v87\n\t"               /* (unconditionally: see above) */                      \
    "v_bfe_u32 v71, %[c], 3, 7\n\t"                                                                        \
    "v_cndmask_b32_e64 v88, " CELL ", %[k], %[spare]\n\t" /* queue entry: the cell, or the fourth lane's k (this step's) */ \
    "v_sub_u32 v71, %[ringl], v71\n\t"            /* ring address of the next window: 128 - ((c >> 3) & 127) */ \
    "s_mov_b64 exec, %[spare]\n\t"                  /* the fourth lanes only */                             \
    "ds_read_b64 " DL ", v71\n\t"                  /* the next step's window: as early as the cursor allows, it is needed ten instructions into that step */ \
    MZD_Q4_QWR("ds_write_b64 %[qwa], " DM " offset:" QW "\n\t") /* this step's window for stage B */       \
    "s_mov_b64 exec, -1\n\t"                                                                                \
    MZD_Q4_QWR("ds_write_b16 %[qca], v88 offset:" QC "\n\t")                                               \
    "v_and_b32 %[k], 7, %[c]\n\t"                                                                           \
    "v_sub_u32 v85, 64, %[k]\n\t"                                                                          \
    LIMIT                                           /* the next limit = min(64 - k, rem + 1[, steps before the last]) */ \
    RLOW                                                                                                    \
    "s_cbranch_vccnz " OUT "\n\t"
#define Q4_PUBLISH(OUT)                                                                                     \
    "s_add_u32 %[i], %[i], 4\n\t"                                                                           \
    "v_mov_b32 v68, %[i]\n\t"                                                                              \
    "ds_write_b32 %[heada], v68\n\t"                                                                       \
    "s_cmp_lt_u32 %[i], %[nmax]\n\t"                                                                        \
    "s_cbranch_scc0 " OUT "\n\t"
#define Q4_X1 "L_q4_x1_%="
#define Q4_X2 "L_q4_x2_%="
#define Q4_X3 "L_q4_x3_%="
#define Q4_X4 "L_q4_x4_%="
#define Q4_OUTO "L_q4_outo_%="
#define Q4_RLOW "ds_read_b32 v89, %[chan4] offset:%[o_rlow]\n\t" /* for the next batch's ring check */
#define Q4_QC(S) (512 + offsetof(Q4Shared, q1c) + (S) * kQ4Cols * 8)
#define Q4_QW(S) (512 + offsetof(Q4Shared, q1w) + (S) * kQ4Cols * 8)
#ifndef MZD_Q4_PADW
#define MZD_Q4_PADW 0  /* dwords between the 64-byte boundary and the loop's first instruction */
#endif
#define MZD_Q4_STR2(x) #x
#define MZD_Q4_STR(x) MZD_Q4_STR2(x)
#define Q4_HOT_LOOP(LIMIT, BOUND)                                                                                       \
                asm volatile(                                                                                           \
                    /* prologue = what the tail of a step before would have done.  The ring must hold the bytes at the  \
                       cursor BEFORE they are read (the very first entry: wave P may not have filled anything yet;      \
                       after a general step: the cursor has moved by more than a hot step). */                          \
                    "v_mov_b32 v89, -1\n\t"  /* no ring_low read ahead yet: the first batch check takes the slow path */ \
                    Q4_OFF                                                                                              \
                    Q4_RINGCHK("e")                                                                                     \
                    "L_q4_goe_%=:\n\t"                                                                                  \
                    "v_bfe_u32 v71, %[c], 3, 7\n\t"                                                                    \
                    "v_and_b32 %[k], 7, %[c]\n\t"                                                                       \
                    "v_sub_u32 v71, %[ringl], v71\n\t"                                                                \
                    "v_sub_u32 v85, 64, %[k]\n\t"                                                                      \
                    LIMIT                                                                                               \
                    "v_lshl_add_u32 v66, %[sa], 1, %[cb]\n\t"                                                          \
                    "ds_read_u16 v69, v66\n\t"                                                                        \
                    "ds_read_u16 v68, v66\n\t"  /* (the cells alternate between v69 and v68 like the windows: the entry step's is either) */ \
                    "s_mov_b64 exec, %[spare]\n\t"                                                                      \
                    "ds_read_b64 v[90:91], v71\n\t"                                                                  \
                    "ds_read_b64 v[92:93], v71\n\t"                                                                  \
                    "s_mov_b64 exec, -1\n\t"                                                                            \
                    "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
                    "s_and_b32 s86, %[i], 7\n\t"                                                                        \
                    "s_and_b32 %[i], %[i], -4\n\t" /* inside the loop the counter stands at the start of the batch */   \
                    "s_cmp_eq_u32 s86, 0\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_top0_%=\n\t"                                                                   \
                    "s_cmp_eq_u32 s86, 1\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_go1_%=\n\t"                                                                    \
                    "s_cmp_eq_u32 s86, 2\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_go2_%=\n\t"                                                                    \
                    "s_cmp_eq_u32 s86, 3\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_go3_%=\n\t"                                                                    \
                    "s_cmp_eq_u32 s86, 4\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_top4_%=\n\t"                                                                   \
                    "s_cmp_eq_u32 s86, 5\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_go5_%=\n\t"                                                                    \
                    "s_cmp_eq_u32 s86, 6\n\t"                                                                           \
                    "s_cbranch_scc1 L_q4_go6_%=\n\t"                                                                    \
                    "s_branch L_q4_go7_%=\n"                                                                            \
                    /* (never executed: the loop's place in its 64-byte instruction lines is fixed here, not left to what the      \
                       compiler happens to emit in front of it -- the same stream 4 bytes later has measured 3 % slower) */        \
                    ".p2align 6\n\t.fill " MZD_Q4_STR(MZD_Q4_PADW) ", 4, 0xBF800000\n"                                 \
                    Q4_CHECK("0", "%[tail0]", "%[o_tail0]")                                                             \
                    Q4_STEP("v[90:91]", "v[92:93]", "v69", "v68", "a", "b", "0", "%[qc0]", "%[qw0]", Q4_X1, "", LIMIT)          \
                    Q4_STEP("v[92:93]", "v[90:91]", "v68", "v69", "b", "a", "1", "%[qc1]", "%[qw1]", Q4_X2, "", LIMIT)          \
                    Q4_STEP("v[90:91]", "v[92:93]", "v69", "v68", "a", "b", "2", "%[qc2]", "%[qw2]", Q4_X3, "", LIMIT)          \
                    Q4_STEP("v[92:93]", "v[90:91]", "v68", "v69", "b", "a", "3", "%[qc3]", "%[qw3]", Q4_X4, Q4_RLOW, LIMIT)     \
                    Q4_PUBLISH(Q4_OUTO)                                                                                 \
                    Q4_CHECK("4", "%[tail1]", "%[o_tail1]")                                                             \
                    Q4_STEP("v[90:91]", "v[92:93]", "v69", "v68", "a", "b", "4", "%[qc4]", "%[qw4]", Q4_X1, "", LIMIT)          \
                    Q4_STEP("v[92:93]", "v[90:91]", "v68", "v69", "b", "a", "5", "%[qc5]", "%[qw5]", Q4_X2, "", LIMIT)          \
                    Q4_STEP("v[90:91]", "v[92:93]", "v69", "v68", "a", "b", "6", "%[qc6]", "%[qw6]", Q4_X3, "", LIMIT)          \
                    Q4_STEP("v[92:93]", "v[90:91]", "v68", "v69", "b", "a", "7", "%[qc7]", "%[qw7]", Q4_X4, Q4_RLOW, LIMIT)     \
                    Q4_PUBLISH(Q4_OUTO)                                                                                 \
                    "s_branch L_q4_top0_%=\n"                                                                           \
                    /* a step that leaves: the counter moves past it; after the first / third step of a batch the new   \
                       states are in set b, after the second / fourth in set a (vcc still holds the chains that did not go) */ \
                    "L_q4_x1_%=:\n\t"                                                                                   \
                    "s_add_u32 %[i], %[i], 1\n\t"                                                                       \
                    "s_branch L_q4_oute_%=\n"                                                                           \
                    "L_q4_x3_%=:\n\t"                                                                                   \
                    "s_add_u32 %[i], %[i], 3\n"                                                                         \
                    "L_q4_oute_%=:\n\t"                                                                                 \
                    "v_cndmask_b32 %[sa], %[sb], %[sa], vcc\n\t"                                                        \
                    "s_branch L_q4_done_%=\n"                                                                           \
                    "L_q4_x2_%=:\n\t"                                                                                   \
                    "s_add_u32 %[i], %[i], 2\n\t"                                                                       \
                    "s_branch L_q4_outo_%=\n"                                                                           \
                    "L_q4_x4_%=:\n\t"                                                                                   \
                    "s_add_u32 %[i], %[i], 4\n"                                                                         \
                    "L_q4_outo_%=:\n\t"                                                                                 \
                    "v_cndmask_b32 %[sa], %[sa], %[sb], vcc\n"                                                          \
                    "L_q4_done_%=:\n\t"                                                                                 \
                    "v_cndmask_b32_e64 v87, 0, v87, vcc\n\t" /* chains that did not go: the cursor back */            \
                    "v_sub_u32 %[c], %[c], v87\n\t"                                                                    \
                    "s_mov_b64 %[smask], vcc\n\t"                                                                      \
                    "s_waitcnt lgkmcnt(0)\n\t"  /* the speculative cell read and the ring read are still on their way */ \
                    : [sa] "+v"(st), [sb] "+v"(stb), [k] "+v"(wk), [c] "+v"(cur), [i] "+s"(i),     \
                      [left] "+v"(left), [tail0] "+s"(tail0), [tail1] "+s"(tail1), [polls] "+s"(polls), [smask] "=&s"(smask)               \
                    : [cb] "v"(cb), [shr] "v"(shr), [Kc] "v"(Kc), [nbK] "v"(nbK), [ringl] "v"(ringl128), [w0] "v"(W0), [r0] "v"(R0), [qca] "v"(qca), [qwa] "v"(qwa), \
                      [chan4] "v"(chan4), [heada] "v"(heada), [vzero] "v"(vzero),                                       \
                      [nmax] "s"(BOUND), [spare] "s"(sparemask),                                                       \
                      [o_tail0] "n"(512 + offsetof(Q4Shared, tailB)), [o_tail1] "n"(512 + offsetof(Q4Shared, tailB) + 4), \
                      [o_prog] "n"(512 + offsetof(Q4Shared, progress)), [o_rlow] "n"(512 + offsetof(Q4Shared, ring_low)), \
                      [qc0] "n"(Q4_QC(0)), [qc1] "n"(Q4_QC(1)), [qc2] "n"(Q4_QC(2)), [qc3] "n"(Q4_QC(3)),               \
                      [qc4] "n"(Q4_QC(4)), [qc5] "n"(Q4_QC(5)), [qc6] "n"(Q4_QC(6)), [qc7] "n"(Q4_QC(7)),               \
                      [qw0] "n"(Q4_QW(0)), [qw1] "n"(Q4_QW(1)), [qw2] "n"(Q4_QW(2)), [qw3] "n"(Q4_QW(3)),               \
                      [qw4] "n"(Q4_QW(4)), [qw5] "n"(Q4_QW(5)), [qw6] "n"(Q4_QW(6)), [qw7] "n"(Q4_QW(7))                \
                    : "memory", "vcc", "scc", "s86",                                                                    \
                      "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",                   \
                      "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84",                           \
                      "v85", "v86", "v87", "v88", "v90", "v91", "v92", "v93", "v89")
                // the asm block's cursor: bits consumed below W0 (a multiple of 128 at or above the window; 0 for a parked quad)
                uint32_t cur = ((W0 - woff) << 3) + wk;
                const uint32_t R0 = rem1 + cur;  // bits that remain + 1 = R0 - cur
                if (i < ncom) {
                    Q4_HOT_LOOP("", ncom);
                } else {
                    Q4_HOT_LOOP("v_sub_u32 v67, %[r0], %[c]\n\tv_min3_u32 v85, v85, v67, %[left]\n\tv_add_u32 %[left], -64, %[left]\n\t", nmax);
                }
                woff = W0 - (cur >> 3);
                wk = cur & 7u;
                rem1 = R0 - cur;
                // (the loop of the common steps does not look at the bits that remain: a chain that has read past the start of its
                // stream has rem1 = remaining + 1 <= 0 from here on -- as an unsigned limit that never binds; the general step of its
                // next escape, or of its last sequence at the latest, sees total > remaining and ends it with MZD_ERR_SEQ_BITS,
                // sequences.go:197-204.  Wave P stops following a cursor more than 56 bytes below the stream: nothing is read
                // outside the blob's front slack.)
#undef Q4_HOT_LOOP
#undef Q4_OFF
#undef Q4_STEP
#undef Q4_CHECK
#undef Q4_RINGCHK
#undef Q4_PUBLISH
#undef Q4_X1
#undef Q4_X2
#undef Q4_X3
#undef Q4_X4
#undef Q4_OUTO
#undef Q4_RLOW
#undef Q4_QC
#undef Q4_QW
            }
            // i has moved past the step; chains in smask have not done it yet
            if (smask) {
                const bool mine = ((smask >> lane) & 1) != 0;
#ifdef MZD_Q4_STATS
                n_general++;
                const long long g_t0 = clock64();
#endif
                general_step(i - 1, mine);
#ifdef MZD_Q4_STATS
                general_cycles += clock64() - g_t0;
#endif
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&shs->head1[wave], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#ifdef MZD_Q4_STATS
        if (lane == 0) {
            atomicAdd(&g_q4_stats[0], 1ull);
            atomicAdd(&g_q4_stats[1], (unsigned long long)nmax);
            atomicAdd(&g_q4_stats[2], (unsigned long long)(clock64() - stats_t0));
            atomicAdd(&g_q4_stats[3], (unsigned long long)(polls & 0xFFFF));
            atomicAdd(&g_q4_stats[4], (unsigned long long)(polls >> 16));
            atomicAdd(&g_q4_stats[5], (unsigned long long)n_general);
            atomicAdd(&g_q4_stats[6], (unsigned long long)general_cycles);
        }
#endif
        (void)polls;
        if (spare && has && t.n_seq > 0) shs->stA[ch] = status;
    } else if (lw == 4 || lw == 5 || lw >= 9) {
        // ================= stage B: field extraction and values, four steps at a time.  TWO wavefronts: wave 4 takes the
        // even batches (queue slots 0..3), wave 5 the odd ones (slots 4..7) -- a wavefront issues an instruction every ~6
        // cycles, and the ~55 of a stage-B step would otherwise be longer than stage A's step.
        // Round 5: THREE of them.  Stage A may run two batches ahead of the READ of a batch (the queue is two batches deep), and
        // with two wavefronts at 70 % duty the one whose turn it is was still computing its batch before most of the time: the
        // chain wavefronts polled for queue space 0.6-0.7 times per batch (302 cycles per step against 290 with stages B and C
        // compiled to no-ops).  A third wavefront does not add work, it adds slack: whose turn it is has been idle for a while.
        // Slots, and the two counters per slot parity that stages A and C1 watch, go by the BATCH (batch b: slots 4 (b & 1) ...):
        // a batch's slots are refilled only after the batch two before it has been read / stored, so each counter still only
        // ever moves forward, whichever wavefront writes it.
        const uint32_t bid = lw >= 9 ? (uint32_t)lw - 7u : (uint32_t)lw - 4u;
        const int col = min(lane, kQ4Cols - 1);  // lanes 56..63 have no column: they shadow the last one
        uint32_t head_seen = 0, tail_seen = 0;
#ifdef MZD_Q4_PROF
        long long prof_in = 0, prof_out = 0, prof_t0 = clock64();
#endif
        for (uint32_t j0 = bid * kPipeBatch; j0 < nmax; j0 += (uint32_t)kQ4BWaves * kPipeBatch) {
            const uint32_t par = (j0 / kPipeBatch) & 1u;
            const uint32_t need = min(j0 + (uint32_t)kPipeBatch, nmax);
#ifdef MZD_Q4_PROF
            const long long w0 = clock64();
#endif
            while (head_seen < need) {
                const uint32_t h0 = __hip_atomic_load(&shs->head1[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t h1 = __hip_atomic_load(&shs->head1[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t h2 = __hip_atomic_load(&shs->head1[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t h3 = __hip_atomic_load(&shs->head1[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(min(h0, h1), min(h2, h3)));
                if (head_seen < need) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_Q4_PROF
            prof_in += clock64() - w0;
#endif
            asm volatile("" ::: "memory");
            uint64_t T[kPipeBatch], Cq[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                T[u] = shs->q1w[(j0 + u) % kPipeDepth][col];
                Cq[u] = *(const uint64_t *)&shs->q1c[(j0 + u) % kPipeDepth][col][0];
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->tailB[par], need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint64_t q[kPipeBatch];
#ifdef MZD_EXP_FAST_BC  // timing experiment only (wrong results): what stage A can do when nothing holds it up
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) q[u] = T[u] ^ Cq[u];
#else
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                const uint32_t xl = (uint32_t)Cq[u] & 0xFFFF, xm = (uint32_t)(Cq[u] >> 16) & 0xFFFF, xo = (uint32_t)(Cq[u] >> 32) & 0xFFFF;
                const uint32_t kk = (uint32_t)(Cq[u] >> 48);
                const uint32_t cl = CTc[xl >> 10];
                const uint32_t cm = CTc[64 + (xm >> 10)];
                const uint32_t exO = xo >> 10;
                const uint64_t Tw = T[u] << (kk & 7u);
                const uint32_t hi = (uint32_t)(Tw >> 32);
                const uint32_t exL = cl >> 24, exM = cm >> 24;
                const uint32_t ofx = __builtin_amdgcn_ubfe(hi, 32u - exO, exO);
                const uint32_t Y = (uint32_t)((Tw << exO) >> 32);
                const uint32_t mlx = __builtin_amdgcn_ubfe(Y, 32u - exM, exM);
                const uint32_t llx = __builtin_amdgcn_ubfe(Y, 32u - exM - exL, exL);
                const uint32_t ofv = min((1u << exO) + ofx, kRecOffSymbolic);  // exO <= 31: no wrap
                const uint32_t llv = (cl & 0xFFFFFF) + llx;
                const uint64_t v = (uint64_t)llv | ((uint64_t)((cm & 0xFFFFFF) + mlx) << kRecMlShift) |
                                   ((uint64_t)q4_offset_code(ofv, llv) << kRecOffShift);
                q[u] = (kk & 0x8000u) ? T[u] : v;
            }
#endif
#ifdef MZD_Q4_PROF
            const long long w1 = clock64();
#endif
            // the four slots this wavefront writes were last used by ITS batch before (j0 - 8): free once stage C2 has stored it
            while (j0 + (uint32_t)kPipeBatch - tail_seen > (uint32_t)kPipeDepth) {
                tail_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->tail2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (j0 + (uint32_t)kPipeBatch - tail_seen > (uint32_t)kPipeDepth) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_Q4_PROF
            prof_out += clock64() - w1;
#endif
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) shs->q2[(j0 + u) % kPipeDepth][col] = q[u];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->head2[par], need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef MZD_Q4_PROF
        if (blockIdx.x == 0 && lane == 0) printf("B%u: cycles %lld wait_in %lld wait_out %lld (steps %u)\n", bid, clock64() - prof_t0, prof_in, prof_out, nmax);
#endif
    } else if (lw == 6) {
        // ================= stage C1: repeat-offset history (sequence_execution.go:65-114), record packing =================
        // Branch-free per sequence: errors are sticky (a failed block's records, sums and history are never used), the
        // history update is a chain of selects.  The steps that every chain of the workgroup still has (the first
        // `ncommon`) skip the "is this chain still active" predicate.  The finished records go back into the queue slot
        // they came from; stage C2 keeps the running sums and stores them (a wavefront issues an instruction every ~6
        // cycles: sums, tile bases and the scattered stores on top of the history would make this stage's step longer
        // than stage A's).
        const int col = min(lane, kQ4Cols - 1);
        uint32_t h0, h1, h2;  // shifted left by 3, as the offset field sits in a record's high dword
        if (t.hist_known) { h0 = 1u << 3; h1 = 4u << 3; h2 = 8u << 3; }  // framedecompressor.go:48,59
        else { h0 = (uint32_t)-1 << 3; h1 = (uint32_t)-2 << 3; h2 = (uint32_t)-3 << 3; }
        uint32_t maxv = 0, minv = 0xFFFFFFFFu;
        const uint32_t my_n = has ? t.n_seq : 0u;
        const uint32_t ncommon = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_min_u32(my_n ? my_n : 0xFFFFFFFFu)) & ~3u;
        uint32_t head_seen0 = 0, head_seen1 = 0;
#ifdef MZD_Q4_PROF
        long long prof_in = 0, prof_t0 = clock64();
#endif
        auto batch = [&](uint32_t j0, auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
            const uint32_t need = min(j0 + (uint32_t)kPipeBatch, nmax);
            uint32_t &head_seen = (j0 & kPipeBatch) ? head_seen1 : head_seen0;
#ifdef MZD_Q4_PROF
            const long long w0 = clock64();
#endif
            while (head_seen < need) {
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->head2[(j0 >> 2) & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head_seen < need) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_Q4_PROF
            prof_in += clock64() - w0;
#endif
            asm volatile("" ::: "memory");
            uint32_t qh[kPipeBatch];  // only the high dword of a record changes (offset code -> resolved offset)
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) qh[u] = ((const uint32_t *)&shs->q2[(j0 + u) % kPipeDepth][col])[1];
#ifdef MZD_EXP_FAST_BC
            maxv += qh[0] ^ qh[1] ^ qh[2] ^ qh[3];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->head3, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
#endif
            uint32_t rh[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                if (!CHECKED) {
                    rh[u] = q4_history_step(qh[u], h0, h1, h2, maxv, minv);
                } else {
                    // the same step for the chains that still have a sequence j0 + u; the others keep their history
                    const bool act = j0 + u < my_n;
                    const uint32_t hi = qh[u];
                    const uint32_t idx = act ? min(hi >> 3, 5u) : 0u;
                    uint32_t off = (hi & ~7u) - 32u;                                       // new offset
                    off = idx == 4 ? ((int)h0 > 0 ? h0 - 8u : h0 - 32u) : off;             // sequence_execution.go:65-114
                    off = idx == 3 ? h2 : off;
                    off = idx == 2 ? h1 : off;
                    off = idx <= 1 ? h0 : off;
                    h2 = idx >= 3 ? h1 : h2;
                    h1 = idx >= 2 ? h0 : h1;
                    h0 = idx >= 2 ? off : h0;
                    maxv = max(maxv, act ? hi : 0u);
                    minv = min(minv, act ? off : 0xFFFFFFFFu);
                    rh[u] = (hi & 7u) | ((int)off > 0 ? off : kQ4SymBase - off);
                }
            }
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) ((uint32_t *)&shs->q2[(j0 + u) % kPipeDepth][col])[1] = rh[u];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->head3, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        uint32_t j0 = 0;
        for (; j0 < min(ncommon, nmax); j0 += kPipeBatch) batch(j0, std::false_type{});
        for (; j0 < nmax; j0 += kPipeBatch) batch(j0, std::true_type{});
#ifdef MZD_Q4_PROF
        if (blockIdx.x == 0 && lane == 0) printf("C1: cycles %lld wait_in %lld (common steps %u)\n", clock64() - prof_t0, prof_in, ncommon);
#endif
        // errors: an offset value >= 2^28 (unsupported; its code is one more), a zero offset (ringbuffer.go:189)
        const int status = (maxv >> 3) > kRecOffSymbolic ? MZD_ERR_UNSUPPORTED : (minv == 0 ? MZD_ERR_OFFSET : MZD_OK);
        if (has && t.n_seq > 0) {
            BlockSum *bs = &sums[t.block];
            bs->hist[0] = (int)h0 >> 3;
            bs->hist[1] = (int)h1 >> 3;
            bs->hist[2] = (int)h2 >> 3;
            bs->reach = maxv >> 3;  // the largest offset code = the largest new offset + 4 (block mode: how far back a frame's matches go)
        }
        shs->stC[lane] = status;
    } else if (lw == 7) {
        // ================= stage C2: running sums, tile bases, and the finished records leave for HBM =================
        // two 16-byte stores per lane and batch instead of four 8-byte ones (every store is a scatter over the chains'
        // record streams)
        const int col = min(lane, kQ4Cols - 1);
        uint64_t *myrec = recs + t.rec_off;
        TileBase *mytile = tiles + t.tile_off;
        const uint32_t my_n = has ? t.n_seq : 0u;
        uint32_t litPos = 0, outPos = 0, err_size = 0;
        uint32_t head_seen = 0;
#ifdef MZD_Q4_PROF
        long long prof_in = 0, prof_t0 = clock64();
#endif
        for (uint32_t j0 = 0; j0 < nmax; j0 += kPipeBatch) {
            const uint32_t need = min(j0 + (uint32_t)kPipeBatch, nmax);
#ifdef MZD_Q4_PROF
            const long long w0 = clock64();
#endif
            while (head_seen < need) {
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->head3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head_seen < need) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_Q4_PROF
            prof_in += clock64() - w0;
#endif
            asm volatile("" ::: "memory");
            uint64_t rr[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) rr[u] = shs->q2[(j0 + u) % kPipeDepth][col];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->tail2, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef MZD_EXP_FAST_BC
            if ((j0 & 63) == 0 && j0 < my_n) mytile[j0 >> 6] = TileBase{litPos, outPos};
            if (j0 + (uint32_t)kPipeBatch <= my_n) {
#pragma unroll
                for (int u = 0; u < kPipeBatch; u++) {
                    const uint32_t LL = (uint32_t)rr[u] & kRecLlMask, ML = (uint32_t)(rr[u] >> kRecMlShift) & kRecMlMask;
                    litPos += LL;
                    outPos += LL + ML;
                }
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
                *(u64x2 *)(myrec + j0) = u64x2{rr[0], rr[1]};
                *(u64x2 *)(myrec + j0 + 2) = u64x2{rr[2], rr[3]};
            } else {
#pragma unroll
                for (int u = 0; u < kPipeBatch; u++)
                    if (j0 + u < my_n) {
                        const uint32_t LL = (uint32_t)rr[u] & kRecLlMask, ML = (uint32_t)(rr[u] >> kRecMlShift) & kRecMlMask;
                        litPos += LL;
                        outPos += LL + ML;
                        myrec[j0 + u] = rr[u];
                    }
            }
            err_size |= outPos > kBlockMax;  // a block regenerates <= 128 KiB (four steps add < 2^21: no wrap between checks)
#endif
        }
#ifdef MZD_Q4_PROF
        if (blockIdx.x == 0 && lane == 0) printf("C2: cycles %lld wait_in %lld\n", clock64() - prof_t0, prof_in);
#endif
        if (has && t.n_seq > 0) {
            BlockSum *bs = &sums[t.block];
            bs->lit_total = litPos;
            bs->out_total = outPos;
        }
        shs->stC2[lane] = err_size ? MZD_ERR_CORRUPT_SIZES : MZD_OK;
    } else {
        // ================= wave P: the chains' bitstreams, ahead of stage A (as k_seq_pipe) =================
        const uint8_t *inb = in - MZD_IN_PAD;
        const uint8_t *sbase = in + t.in_off;
        const bool work = has && t.n_seq > 0;
        int low = (int)t.in_size;  // prefetch touches: everything at or above `low` has been requested
        constexpr int kAhead = MZD_PIPE_AHEAD, kLine = 128;
        uint32_t rlow = ((uint32_t)t.in_off + MZD_IN_PAD + t.in_size + 31u) & ~31u;  // ring: nothing yet
        uint8_t *ring = shs->ring[lane];
        uint32_t iter = 0;
        for (;;) {
            const uint32_t h0 = __hip_atomic_load(&shs->head1[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t h1 = __hip_atomic_load(&shs->head1[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t h2 = __hip_atomic_load(&shs->head1[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t h3 = __hip_atomic_load(&shs->head1[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t hd = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(min(h0, h1), min(h2, h3)));
            // the chain's window offset; stage A reads the 8 bytes AT it (k_seq_pipe's stage A read the 8 bytes at its
            // refill offset = 8 below its window: the same protocol with the same numbers)
            const uint32_t raw = __hip_atomic_load(&shs->progress[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int cur = (int)(raw - ((uint32_t)t.in_off + MZD_IN_PAD));
            const bool inside = cur >= -56 && cur <= (int)t.in_size;
            if (work && inside) {
                for (int g = 0; g < 4 && raw <= rlow + 88u && rlow >= 32u; g++) {
                    const uint32_t u = rlow - 32u;
                    const uint64_t w0 = ld64u(inb + u), w1 = ld64u(inb + u + 8), w2 = ld64u(inb + u + 16), w3 = ld64u(inb + u + 24);
                    uint64_t *d = (uint64_t *)(ring + (u & (kPipeRing - 1)));
                    d[0] = w0; d[1] = w1; d[2] = w2; d[3] = w3;
                    if ((u & (kPipeRing - 1)) == 0) *(uint64_t *)(ring + kPipeRing) = w0;
                    rlow = u;
                }
                asm volatile("" ::: "memory");
                __hip_atomic_store(&shs->ring_low[lane], rlow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_store(&shs->ring_low[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            {
                const int target = inside ? max(cur - kAhead, 0) : low;
#ifdef MZD_Q4_TOUCH_WAIT
                for (int g = 0; g < MZD_PIPE_TOUCHES && has && low > target && (iter & MZD_PIPE_TOUCH_EVERY) == 0; g++) {
                    low = max(low - kLine, 0);
                    touch_line(sbase + (low & ~3));
                }
#else
                // (all of an iteration's touches in flight together, ONE wait: waited for one by one, eight misses in a row kept
                // this wavefront from its rings for eight memory latencies and the chain wavefronts polled for them)
                // (two lines every other iteration: 8.53 ms; eight every eighth, k_seq_pipe's rhythm: 8.58; sixteen-iteration rhythms: worse)
#ifndef MZD_Q4_TOUCH_EVERY
#define MZD_Q4_TOUCH_EVERY 1
#define MZD_Q4_TOUCHES 2
#endif
                if ((iter & MZD_Q4_TOUCH_EVERY) == 0) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int g = 0; g < MZD_Q4_TOUCHES; g++) {
                        if (has && low > target) {
                            low = max(low - kLine, 0);
                            acc += *(const uint32_t *)(sbase + (low & ~3));
                        }
                    }
                    asm volatile("" ::"v"(acc));
                }
#endif
            }
            if (hd >= nmax) break;
            iter++;
#ifndef MZD_Q4_PSLEEP
#define MZD_Q4_PSLEEP 1  // (2 -> 1: 8.65 -> 8.55 ms; 0: the same)
#endif
            __builtin_amdgcn_s_sleep(MZD_Q4_PSLEEP);
        }
    }
    __syncthreads();
#ifdef MZD_Q4_PROF
    if ((blockIdx.x == 0 || blockIdx.x == 700) && threadIdx.x == 0) {
        const long long w2 = wall_clock64(), c2 = clock64();
        printf("workgroup %u: staging %.1f us, whole %.1f us, shader clock %.0f MHz\n", blockIdx.x, (prof_w1 - prof_w0) / 100.0,
               (w2 - prof_w0) / 100.0, (double)(c2 - prof_c0) / ((w2 - prof_w0) / 100.0));
    }
#endif
    // decode-stage errors come first, as in the reference, where DecodeSequences runs to its end
    // before ExecuteSequences starts
    if (lw == 6 && has && t.n_seq > 0) {
        int st = shs->stA[lane];
        if (st == MZD_OK) st = shs->stC[lane];
        if (st == MZD_OK) st = shs->stC2[lane];
        if (st != MZD_OK) atomicCAS(&sums[t.block].status, MZD_OK, st);
    }
}

}  // namespace mzd
