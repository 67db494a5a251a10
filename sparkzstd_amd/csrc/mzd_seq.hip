// mzd_seq.hip -- what the sequence stage's kernels share (code tables, the register bit window, the pipeline's constants:
// sequences.go:64-206, fse.go:253-290) and, for the parity tests only (-DMZD_TEST_KERNELS), k_seq and k_seq_pipe.  The stage's kernel is
// k_seq_q4 (mzd_seq_q4.hip).  Split out of mzd_kernels.hip in round 6; included by mzd_api.hip behind mzd_huf.hip.
#pragma once

namespace mzd {

// ------------------------------------------------------------------------------------------
// k_seq: FSE sequence decode.  One wavefront per workgroup, lane = one block's chain.
//
// LDS cell (built from the host cells {baseline, nbits, symbol} while staging), 2 bytes:
//   next(10) | symbol(6)       nbits = acc_log - highbit(next), baseline = (next << nbits) - size
//   (fse.go:209-213 run backwards) -> 61 chains per CU
// Constant LDS table CT[kind][symbol] = base_value(24) | extra_bits(8)  (predefined.go:5-20,36-50).

__constant__ uint32_t c_ll_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                       20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400,
                                       0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
__constant__ uint8_t c_ll_extra[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                       1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t c_ml_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                       21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                       39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                       4099, 8195, 16387, 32771, 65539};
__constant__ uint8_t c_ml_extra[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                       0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                       2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

__device__ __forceinline__ int hist_dec(int x) { return x > 0 ? x - 1 : x - 4; }

// Unmasked variant of the register bit window for k_seq: a valid sequence bitstream is consumed
// exactly to bit 0, so bytes below the stream start are never interpreted (over-reads are
// detected through the bit budget `rem`); the input blob carries MZD_IN_PAD bytes of slack.
struct SeqBits {
    const uint8_t *pd;  // address of D's bytes == stream + ptr - 8
    uint64_t C, D;
    int k;
    __device__ __forceinline__ int init(const uint8_t *start, int len)
    {
        BackBits b;
        const int r = b.init(start, len);  // masked loads once, for streams shorter than 16 bytes
        C = b.C; D = b.D; k = b.k;
        pd = start + (len - 16);
        return r;
    }
    __device__ __forceinline__ void refill()
    {
        const int nb = k >> 3;
        const int sh = nb * 8;
        C = (C << sh) | ((D >> 1) >> (63 - sh));
        pd -= nb;
        k &= 7;
        // ordering point: the old D must be dead before the new D is requested, otherwise the
        // compiler keeps both alive, copies at the loop back edge and waits vmcnt(0) for the copy
        asm volatile("" ::"v"((uint32_t)C), "v"((uint32_t)(C >> 32)) : "memory");
        D = ld64u(pd);
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(((C << k) >> 1) >> (63 - n)); }
};

// top n (0..31) bits of the 64-bit left-justified window T; n == 0 -> 0 (v_bfe_u32 width 0)
__device__ __forceinline__ uint32_t top_bits(uint64_t T, uint32_t n)
{
    return __builtin_amdgcn_ubfe((uint32_t)(T >> 32), 32u - n, n);
}

#ifdef MZD_TEST_KERNELS  /* round 6: second implementations of the sequence stage for the parity tests (libmzd_test.so) */
// LDS after the cell slots and the constant table.  The decode wavefront hands every decoded
// sequence to the helper wavefront through `queue` (all chains of a wavefront are at the same step
// index, so one head / tail pair serves the whole wavefront).
template <int DEPTH>
struct SeqShared {
    uint32_t progress[64];  // bytes of each chain's bitstream still unread (published every 32 steps)
    uint32_t head;          // steps produced by the decode wavefront
    uint32_t tail;          // steps consumed by the helper wavefront
    uint32_t pad[2];
    uint64_t queue[DEPTH][64];  // LL:17 | ML:18 | offset value:28 | valid:1
};

__global__ __launch_bounds__(128) void k_seq(const uint8_t *__restrict__ in, const SeqTask *__restrict__ tasks,
                                             uint32_t n_tasks, const uint32_t *__restrict__ fse_entries,
                                             uint64_t *__restrict__ recs, TileBase *__restrict__ tiles,
                                             BlockSum *sums)
{
    constexpr int NCH = kSeqChains16;
    constexpr int CELL_BYTES = 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *CT = (uint32_t *)(smem + (size_t)NCH * kSeqCellsPerChain * CELL_BYTES);  // [2][64]
    constexpr int kSeqQueueDepth = kSeqQueue16;
    SeqShared<kSeqQueueDepth> *shs = (SeqShared<kSeqQueueDepth> *)(CT + 128);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t tid = blockIdx.x * NCH + lane;
    const bool has = lane < NCH && tid < n_tasks;
    SeqTask t;
    if (has) t = tasks[tid];
    else {
        t.n_seq = 0; t.in_size = 0; t.ll_off = t.of_off = t.ml_off = 0; t.ll_log = t.of_log = t.ml_log = 0;
        t.in_off = 0; t.rec_off = 0; t.tile_off = 0; t.block = 0; t.hist_known = 0;
    }
    if (wave == 0) {
        CT[lane] = lane < 36 ? (c_ll_base[lane] | ((uint32_t)c_ll_extra[lane] << 24)) : 0u;
        CT[64 + lane] = lane < 53 ? (c_ml_base[lane] | ((uint32_t)c_ml_extra[lane] << 24)) : 0u;
        shs->progress[lane] = t.in_size;
        if (lane == 0) { shs->head = 0; shs->tail = 0; }
    }
    // stage the three tables of every chain of this workgroup (both wavefronts copy)
    for (int ch = 0; ch < NCH; ch++) {
        if (blockIdx.x * NCH + ch >= n_tasks) break;
        uint32_t off[3], lg[3];
        off[0] = (uint32_t)__shfl((int)t.ll_off, ch, 64);
        off[1] = (uint32_t)__shfl((int)t.ml_off, ch, 64);
        off[2] = (uint32_t)__shfl((int)t.of_off, ch, 64);
        lg[0] = (uint32_t)__shfl((int)t.ll_log, ch, 64);
        lg[1] = (uint32_t)__shfl((int)t.ml_log, ch, 64);
        lg[2] = (uint32_t)__shfl((int)t.of_log, ch, 64);
#pragma unroll
        for (int kind = 0; kind < 3; kind++) {
            const uint32_t n = 1u << lg[kind];
            const uint32_t base = (uint32_t)ch * kSeqCellsPerChain + (uint32_t)kind * 512;
            for (uint32_t i = threadIdx.x; i < n; i += 128) {
                uint32_t e = fse_entries[off[kind] + i];  // baseline(16) | nbits(8) | symbol(8)
                uint32_t baseline = e & 0xFFFF, nb = (e >> 16) & 0xFF, sym = e >> 24;
                const uint32_t next = (baseline + n) >> nb;
                ((uint16_t *)smem)[base + i] = (uint16_t)(next | (sym << 10));
            }
        }
    }
    __syncthreads();

    // wave-uniform trip count in an SGPR; both wavefronts compute the same value
    const uint32_t nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(has ? t.n_seq : 0u));

    if (wave == 1) {
        // ---- helper wavefront.  (1) It drains the sequence queue: repeat-offset resolution
        // (sequence_execution.go:65-114) on a concrete-or-symbolic history, record packing, running
        // sums, tile bases and ALL global stores -- so the decode wavefront never has a store in
        // flight when it waits for its prefetched bits.  (2) It walks ahead of every chain's read
        // cursor and touches the bitstream lines so that the decode wavefront's refills hit L2
        // instead of stalling 64 lanes on one lane's HBM miss.
        const uint8_t *sbase = in + t.in_off;
        int low = (int)t.in_size;  // everything at or above `low` has been requested
        uint32_t sink = 0;
        constexpr int kAhead = 1024, kLine = 128;
        int h0, h1, h2;
        if (t.hist_known) { h0 = 1; h1 = 4; h2 = 8; }  // framedecompressor.go:48,59
        else { h0 = -1; h1 = -2; h2 = -3; }
        uint32_t litPos = 0, outPos = 0;
        int status = MZD_OK;
        uint64_t *myrec = recs + t.rec_off;
        TileBase *mytile = tiles + t.tile_off;
        uint32_t head_seen = 0;  // the counterpart's counter is only re-read when the cached value runs out
        for (uint32_t j = 0; j < nmax; j++) {
            if ((j & 31) == 0) {
                const int cur = (int)__hip_atomic_load(&shs->progress[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int target = max(cur - kAhead, 0);
                int guard = 0;
                while (has && low > target && guard < 16) {
                    low = max(low - kLine, 0);
                    sink ^= *(const volatile uint32_t *)(sbase + (low & ~3));
                    guard++;
                }
            }
            while (head_seen <= j) {
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head_seen <= j) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            const uint64_t q = shs->queue[j % kSeqQueueDepth][lane];
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&shs->tail, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool act = (q >> 63) != 0 && status == MZD_OK;
            const uint32_t LL = (uint32_t)q & kRecLlMask;
            const uint32_t ML = (uint32_t)(q >> kRecMlShift) & kRecMlMask;
            const uint32_t ofv = (uint32_t)(q >> kRecOffShift) & (kRecOffSymbolic - 1);
            if ((j & 63) == 0 && act) mytile[j >> 6] = TileBase{litPos, outPos};
            const bool isnew = ofv > 3;
            const int idx = isnew ? 4 : (int)ofv - 1 + (LL == 0 ? 1 : 0);  // 0..3 repeat cases, 4 = new offset
            int off = idx == 0 ? h0 : (idx == 1 ? h1 : (idx == 2 ? h2 : hist_dec(h0)));
            if (isnew) off = (int)(ofv - 3);
            if (act) {
                if (off == 0) status = MZD_ERR_OFFSET;
                if (idx >= 2) h2 = h1;
                if (idx >= 1) { h1 = h0; h0 = off; }
            }
            if (act && status == MZD_OK) {
                const uint32_t offfield = off > 0 ? (uint32_t)off : (kRecOffSymbolic | (uint32_t)(-off - 1));
                myrec[j] = (uint64_t)LL | ((uint64_t)ML << kRecMlShift) | ((uint64_t)offfield << kRecOffShift);
                litPos += LL;
                outPos += LL + ML;
                if (outPos > kBlockMax) status = MZD_ERR_CORRUPT_SIZES;  // a block regenerates <= 128 KiB
            }
        }
        if (has && t.n_seq > 0) {
            BlockSum *bs = &sums[t.block];
            bs->lit_total = litPos;
            bs->out_total = outPos;
            bs->hist[0] = h0;
            bs->hist[1] = h1;
            bs->hist[2] = h2;
            if (status != MZD_OK) atomicCAS(&bs->status, MZD_OK, status);
        }
        if (sink == 0x9E3779B9u && lane == 77) sums[0].reach = sink;  // keeps the touches alive; never true
        return;
    }

    // ---- decode wavefront: table lookups, bit fields, state updates -- and nothing else
    const uint32_t slot = (uint32_t)lane * kSeqCellsPerChain;
    const int alL = t.ll_log, alM = t.ml_log, alO = t.of_log;
    SeqBits br;
    int rem = 0;
    int status = MZD_OK;
    uint32_t sL = 0, sM = 0, sO = 0;
    if (has && t.n_seq > 0) {
        rem = br.init(in + t.in_off, (int)t.in_size);
        if (rem < 0) {
            status = MZD_ERR_BAD_PADDING;
            rem = 0;
        } else {
            // initial states in the order LL, OF, ML (sequences.go:145-159)
            sL = br.peek(alL); br.k += alL;
            sO = br.peek(alO); br.k += alO;
            br.refill();
            sM = br.peek(alM); br.k += alM;
            rem -= alL + alO + alM;
            if (rem < 0) status = MZD_ERR_SEQ_BITS;
        }
    } else {
        br.pd = in; br.C = br.D = 0; br.k = 0;
    }
    const uint32_t sizeL = 1u << alL, sizeM = 1u << alM, sizeO = 1u << alO;
    sL += sizeL; sM += sizeM; sO += sizeO;  // pre-biased states
    const int nbL0 = alL - 31, nbM0 = alM - 31, nbO0 = alO - 31;  // nbits = acc_log - 31 + clz(next)

    // One sequence step.  SLOW == false is the hot variant: all six bit fields are cut from one
    // 64-bit window; a lane that needs more than 64 - k bits (very long offsets / lengths) does
    // NOT advance in that iteration ("stalls": every update is predicated off) and is reported
    // through the return value.  The hot loop then leaves at its normal bottom, the stalled lanes
    // run one SLOW step (refills between fields) outside it, and the loop resumes.  This keeps a
    // single definition of every loop-carried register in the hot loop.
    // Returns (stall, packed queue entry).
    // raw table cells of the three current states; issued BEFORE the refill arithmetic so that the LDS
    // latency overlaps it.  States are kept pre-biased by the table size (sX = state + size) and the
    // slot pointers are biased the other way, which removes the "- size" of fse.go:213 from the chain.
    const uint16_t *c16L = (const uint16_t *)smem + slot - sizeL;
    const uint16_t *c16M = (const uint16_t *)smem + slot + 512 - sizeM;
    const uint16_t *c16O = (const uint16_t *)smem + slot + 1024 - sizeO;
    auto load_cells = [&](uint32_t &xl, uint32_t &xm, uint32_t &xo) {
        xl = c16L[sL]; xm = c16M[sM]; xo = c16O[sO];
    };
    auto step = [&](auto slow_tag, uint32_t i, bool only, uint64_t &entry, uint32_t xl, uint32_t xm, uint32_t xo) -> bool {
        constexpr bool SLOW = decltype(slow_tag)::value;
        const bool base_act = only && i < t.n_seq && status == MZD_OK;
        const bool lastseq = (i + 1 == t.n_seq);
        // ---- table cells for the three current states
        uint32_t symL, symM, symO, nbL, nbM, nbO, baseL, baseM, baseO, exL, exM;
        uint32_t cl, cm;
        symL = xl >> 10; symM = xm >> 10; symO = xo >> 10;
        cl = CT[symL]; cm = CT[64 + symM];
        const uint32_t nl = xl & 1023, nm = xm & 1023, no = xo & 1023;
        nbL = (uint32_t)(nbL0 + __builtin_clz(nl | 1));
        nbM = (uint32_t)(nbM0 + __builtin_clz(nm | 1));
        nbO = (uint32_t)(nbO0 + __builtin_clz(no | 1));
        baseL = nl << nbL;  // biased: baseline + size
        baseM = nm << nbM;
        baseO = no << nbO;
        exL = cl >> 24; exM = cm >> 24;
        const uint32_t exO = symO;
        if (lastseq) { nbL = 0; nbM = 0; nbO = 0; }  // no state update after the last sequence (sequences.go:178)
        // cumulative bit offsets in stream order: OF extra, ML extra, LL extra, LL state, ML state, OF state
        const uint32_t o2 = exO + exM, o3 = o2 + exL, o4 = o3 + nbL, o5 = o4 + nbM;
        const int total = (int)(o5 + nbO);

        uint32_t ofx, mlx, llx, aL, aM, aO;
        bool act, stall = false;
        if (!SLOW) {
            stall = base_act && (br.k + total > 63);  // k must stay < 64: the refill shifts by 8 * (k >> 3)
            act = base_act && !stall;
            const uint64_t T = br.C << br.k;
            ofx = top_bits(T, exO);
            mlx = top_bits(T << exO, exM);
            llx = top_bits(T << o2, exL);
            aL = top_bits(T << o3, nbL);
            aM = top_bits(T << o4, nbM);
            aO = top_bits(T << o5, nbO);
            // idle, finished, failed and stalled lanes must not advance: the refill pointer is unclamped
            br.k += act ? total : 0;
        } else {
            act = base_act;
            const uint32_t m = act ? 0xFFFFFFFFu : 0u;
            const int wO = (int)(exO & m), wM = (int)(exM & m), wL = (int)(exL & m);
            const int vL = (int)(nbL & m), vM = (int)(nbM & m), vO = (int)(nbO & m);
            ofx = br.peek(wO); br.k += wO; br.refill();
            mlx = br.peek(wM); br.k += wM;
            llx = br.peek(wL); br.k += wL; br.refill();
            aL = br.peek(vL); br.k += vL;
            aM = br.peek(vM); br.k += vM;
            aO = br.peek(vO); br.k += vO;
        }
        // ---- values (sequences.go:99-120)
        const uint32_t ofv = (1u << exO) + ofx;
        const uint32_t ML = (cm & 0xFFFFFF) + mlx;
        const uint32_t LL = (cl & 0xFFFFFF) + llx;
        if (act) {
            rem -= total;
            if (rem < 0) status = MZD_ERR_SEQ_BITS;  // over-read (cursor would pass -1)
            if (ofv >= kRecOffSymbolic) status = MZD_ERR_UNSUPPORTED;  // offset value >= 2^28
            // next states: state = Baseline + Read(NumberOfBits) (fse.go:282-290), order LL, ML, OF.
            // In range by construction: the host checked baseline + 2^nbits <= size for every cell, and
            // idle / finished / failed lanes do not get here.
            sL = baseL + aL; sM = baseM + aM; sO = baseO + aO;
        }
        const bool emit = act && status == MZD_OK;
        entry = emit ? ((uint64_t)LL | ((uint64_t)ML << kRecMlShift) | ((uint64_t)ofv << kRecOffShift) | (1ull << 63)) : 0ull;
        return stall;
    };

    uint32_t i = 0;
    uint32_t tail_seen = 0;
    auto wait_space = [&](uint32_t at) {  // queue slot of step `at` is free once at - tail < depth
        while (at - tail_seen >= (uint32_t)kSeqQueueDepth) {
            tail_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)__hip_atomic_load(&shs->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (at - tail_seen >= (uint32_t)kSeqQueueDepth) __builtin_amdgcn_s_sleep(1);
        }
    };
    while (i < nmax) {
        bool stalled = false;
        bool any_stall = false;
        uint64_t entry = 0;
        do {
            if ((i & 31) == 0 && has)  // bytes not yet requested by the refills (for the helper wavefront)
                shs->progress[lane] = (uint32_t)max((int)(br.pd - (in + t.in_off)), 0);
            uint32_t xl, xm, xo;
            load_cells(xl, xm, xo);
            br.refill();
            stalled = step(std::false_type{}, i, true, entry, xl, xm, xo);
            any_stall = __any(stalled) != 0;
            if (!any_stall) {
                // hand the step to the helper wavefront (space in the queue: i - tail < depth)
                wait_space(i);
                shs->queue[i % kSeqQueueDepth][lane] = entry;
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&shs->head, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            i++;
        } while (i < nmax && !any_stall);
        if (any_stall) {
            // lanes that advanced keep their entry; stalled lanes produce theirs now
            uint64_t e2 = 0;
            uint32_t xl, xm, xo;
            load_cells(xl, xm, xo);
            step(std::true_type{}, i - 1, stalled, e2, xl, xm, xo);
            if (stalled) entry = e2;
            wait_space(i - 1);
            shs->queue[(i - 1) % kSeqQueueDepth][lane] = entry;
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&shs->head, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (has && t.n_seq > 0) {
        if (status == MZD_OK && rem != 0) status = MZD_ERR_SEQ_BITS;  // sequences.go:197-204
        if (status != MZD_OK) atomicCAS(&sums[t.block].status, MZD_OK, status);
    }
}

#endif  // MZD_TEST_KERNELS (k_seq)

// ------------------------------------------------------------------------------------------
// k_seq_pipe: the sequence decode as a THREE-STAGE PIPELINE ACROSS THE SIMDs OF ONE CU.
//
// The LDS-resident tables bound a CU to 54-56 chains = one wavefront, and a lone wavefront pays
// ~4.4 cycles per instruction of whatever type plus ~100 cycles per DEPENDENT LDS round trip: the
// per-step instruction stream and its LDS trips ARE the step latency.  So the step is cut by
// dependence, not by data: only what the next state needs stays on the serial chain, everything
// else moves to other wavefronts (= other SIMDs) that follow a few queue slots behind and work
// on batches of four steps (one poll and one LDS latency per batch instead of per step).
//   wave 0 (A, the chain): cells of the three states, extra-bit COUNTS, refill, the three
//       next-state bit fields, state update.  ONE LDS trip per step; it never cuts the extra
//       bits and never forms a value.  Hands {bit window T at the cursor, symbol codes} to B.
//   wave 1 (B, stateless): cuts offset / match-length / literal-length extra bits out of T and
//       adds the base values (sequences.go:99-120); hands {LL, ML, offset value} to C.
//   wave 2 (C): running sums + tile bases, repeat-offset resolution on a concrete-or-symbolic
//       history (sequence_execution.go:65-114), record packing, the record stores.
//   wave 3 (P): feeds the bitstreams.  Keeps 128 bytes of every chain's stream in an LDS ring
//       (32-byte units) from which A refills its bit window with one ds_read_b64 per step, and
//       touches the lines further below the cursors so that its own unit loads hit L1 / L2.
//       (A used to gather its refill bytes from global memory: 57 distinct lines per step, every
//       128-byte line fetched ~40 times -- that address path bounded the step at full chain count.)
//
// LDS cell (2 bytes): next(10) | c6(6).  next = (baseline + size) >> nbits, from which nbits =
// acc_log - highbit(next) and baseline + size = next << nbits (fse.go:209-213 backwards).  c6 is
// the symbol RE-CODED so that the extra-bit count is arithmetic: count = max(0, (c6 >> 2) - K)
// with K = 3 for literal lengths and 7 for match lengths (seq_code6 below; predefined.go:5-20,
// 36-50 are the counts it reproduces).  Stage B looks base values up by c6.  The few symbols
// that do not fit (literal length >= 8192, match length >= 1027) get next = 0: "escape".
//
// A's hot step has no per-sequence predicate except ONE: a lane takes the general step instead
// (refills between fields, values formed in A itself, queue entry mode 1) when
//   - the step needs more bits than the window holds (k + total > 63) or the stream has left,
//   - a cell is an escape (next = 0 makes clz = -1 and nbits negative = above any limit as unsigned),
//   - it is the lane's last sequence (no state update, sequences.go:178).
// Lanes without work, failed or finished are PARKED: bit budget 0 and a dummy state, so they
// never move and need no exec masking.
//
// LDS: [CTc 128 dwords][PipeShared: counters, queues, bitstream rings][cells: nch x 1280 x u16], nch <= kPipeMaxChains at launch.

constexpr int kPipeRing = 128;  // bytes of every chain's bitstream wave P keeps in LDS for stage A
constexpr int kPipeBatch = 4, kPipeDepth = 8;  // steps per consumer batch; queue depth (two batches)
#ifndef MZD_PIPE_TOUCH_EVERY
#define MZD_PIPE_TOUCH_EVERY 7  // mask on wave P's iteration count: it touches (and waits for the misses) only when
                               // (iter & mask) == 0, so that the ring refills of the other iterations are not held up
#endif
#ifndef MZD_PIPE_TOUCHES
#define MZD_PIPE_TOUCHES 8  // lines wave P touches per chain and iteration at most
#endif
#ifndef MZD_PIPE_AHEAD
#define MZD_PIPE_AHEAD 512  // bytes wave P keeps touched below every chain's cursor
#endif
#ifdef MZD_PIPE_STATS  // whole-pass statistics of stage A (tools/pipe_stats.py): unlike -DMZD_PIPE_PROF, every workgroup counts
__device__ unsigned long long g_pipe_stats[8];  // workgroups, steps, cycles of stage A, queue-full polls, ring polls
#endif
struct PipeShared {
    uint32_t head1, tail1, head2, tail2;  // steps produced / consumed on the A->B and B->C queues
    uint32_t progress[64];                // per chain: bytes of bitstream not yet requested by A
    int32_t stC[64];                      // final status of stage C
    uint64_t q1t[kPipeDepth][64];         // mode 0: bit window T; mode 1: LL:17 | ML:18 | offset value:29
    uint32_t q1p[kPipeDepth][64];         // mode 0: byte 0/1/2 = high byte of the LL/ML/OF cell; mode 1: bit 31
    uint64_t q2[kPipeDepth][64];          // LL:17 | ML:18 | offset value:29 (2^28 = "too large")
    uint32_t ring_low[64];                // per chain: lowest offset (from in - MZD_IN_PAD) wave P has put in the ring
    uint8_t ring[64][kPipeRing + 8];      // per chain: 128 bytes of bitstream at (offset & 127) + the first 8 again
};
constexpr int kPipeFixedLds = 512 + (int)sizeof(PipeShared);
constexpr int kPipeMaxChains = (160 * 1024 - kPipeFixedLds) / (kSeqCellsPerChain * 2);
static_assert(kPipeFixedLds % 16 == 0 && kPipeMaxChains >= 56 && offsetof(PipeShared, ring) % 8 == 0, "k_seq_pipe LDS layout");
constexpr uint32_t kPipeEscape = 64;

// symbol -> c6 (see above); kind 0 = literal lengths, 1 = match lengths
__device__ __forceinline__ uint32_t seq_code6(int kind, uint32_t s)
{
    if (kind == 0) {
        if (s < 20) return s;                 // 0..15: 0 bits (classes 0-3); 16..19: 1 bit (class 4)
        if (s < 22) return 20 + (s - 20);     // 2 bits (class 5)
        if (s < 24) return 24 + (s - 22);     // 3 bits (class 6)
        if (s == 24) return 28;               // 4 bits (class 7); class 8 (5 bits) does not exist
        if (s < 32) return 36 + 4 * (s - 25); // 6..12 bits (classes 9..15)
        return kPipeEscape;                   // 13..16 bits
    }
    if (s < 36) return s;                     // 0..31: 0 bits (classes 0-7); 32..35: 1 bit (class 8)
    if (s < 38) return 36 + (s - 36);         // 2 bits (class 9)
    if (s < 40) return 40 + (s - 38);         // 3 bits (class 10)
    if (s < 42) return 44 + (s - 40);         // 4 bits (class 11)
    if (s == 42) return 48;                   // 5 bits (class 12); class 13 (6 bits) does not exist
    if (s == 43) return 56;                   // 7 bits (class 14)
    if (s == 44) return 60;                   // 8 bits (class 15)
    return kPipeEscape;                       // 9..16 bits
}

__device__ __forceinline__ uint32_t ffbh_raw(uint32_t x)  // v_ffbh_u32: clz, and -1 for 0 (wanted, see escape)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t sub_sat(uint32_t a, uint32_t b)  // max(0, a - b) in one instruction
{
    uint32_t r;
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#ifdef MZD_TEST_KERNELS
__global__ __launch_bounds__(256) void k_seq_pipe(const uint8_t *__restrict__ in, const SeqTask *__restrict__ tasks,
                                                  uint32_t n_tasks, const uint32_t *__restrict__ fse_entries,
                                                  uint64_t *__restrict__ recs, TileBase *__restrict__ tiles,
                                                  BlockSum *sums, uint32_t nch, uint64_t in_base)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *CTc = (uint32_t *)smem;  // [2][64] by c6: base(24) | extra(8)   (predefined.go:5-20,36-50)
    PipeShared *shs = (PipeShared *)(smem + 512);
    uint16_t *cells = (uint16_t *)(smem + kPipeFixedLds);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t tid = blockIdx.x * nch + lane;
    const bool has = (uint32_t)lane < nch && tid < n_tasks;
    SeqTask t;
    if (has) {
        t = tasks[tid];
        t.in_off -= in_base;  // the launch's window of the blob: bitstreams are addressed with 32-bit offsets from it
    } else {
        t.n_seq = 0; t.in_size = 0; t.ll_off = t.of_off = t.ml_off = 0; t.ll_log = t.of_log = t.ml_log = 0;
        t.in_off = 0; t.rec_off = 0; t.tile_off = 0; t.block = 0; t.hist_known = 0;
    }
    in += in_base;
    if (wave == 3) {
        CTc[lane] = 0;
        CTc[64 + lane] = 0;
        shs->progress[lane] = (uint32_t)t.in_off + MZD_IN_PAD + t.in_size;
        shs->ring_low[lane] = (has && t.n_seq > 0) ? 0xFFFFFFFFu : 0u;  // nothing in the ring yet / nothing needed
        shs->stC[lane] = MZD_OK;
        if (lane == 0) { shs->head1 = 0; shs->tail1 = 0; shs->head2 = 0; shs->tail2 = 0; }
        __builtin_amdgcn_s_waitcnt(0);  // the zero fill above before the scattered fill below (same wavefront: LDS is in order)
        if (lane < 36 && seq_code6(0, lane) != kPipeEscape) CTc[seq_code6(0, lane)] = c_ll_base[lane] | ((uint32_t)c_ll_extra[lane] << 24);
        if (lane < 53 && seq_code6(1, lane) != kPipeEscape) CTc[64 + seq_code6(1, lane)] = c_ml_base[lane] | ((uint32_t)c_ml_extra[lane] << 24);
    }
#ifdef MZD_PIPE_PROF
    const long long prof_k0 = clock64();
#endif
    // ---- stage the three tables of every chain of this workgroup: ONE flat loop over the cells of
    // all chains (the LDS cell array is exactly [chain][1280]), 8 independent loads in flight per
    // thread; a loop per chain and table serialises ~340 dependent memory round trips (0.35 ms of a
    // 3.9 ms round).  The table descriptors of the chains go through LDS (the A->B queue is idle yet).
    {
        uint32_t *desc = (uint32_t *)&shs->q1t[0][0];  // [chain][4]: ll_off, ml_off, of_off, logs
        if (wave == 0) {
            desc[4 * lane + 0] = t.ll_off;
            desc[4 * lane + 1] = t.ml_off;
            desc[4 * lane + 2] = t.of_off;
            desc[4 * lane + 3] = has ? ((uint32_t)t.ll_log | ((uint32_t)t.ml_log << 8) | ((uint32_t)t.of_log << 16)) : 0x00FFFFFFu;
        }
        __syncthreads();
        const uint32_t ncell = min(nch, n_tasks - blockIdx.x * nch) * kSeqCellsPerChain;
        constexpr int UNR = 8;
        for (uint32_t idx0 = threadIdx.x; idx0 < ncell; idx0 += 256 * UNR) {
            uint32_t e[UNR], n[UNR], c6k[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const uint32_t idx = idx0 + 256 * u;
                const uint32_t ch = idx / kSeqCellsPerChain, r = idx - ch * kSeqCellsPerChain;
                const uint32_t kind = r >= 1024 ? 2u : (r >> 9);
                const uint32_t i = r - (kind << 9);
                const uint32_t lg = (desc[4 * min(ch, 63u) + 3] >> (8 * kind)) & 0xFF;
                n[u] = 1u << (lg & 31);
                ok[u] = idx < ncell && lg <= 9 && i < n[u];
                c6k[u] = kind;
                e[u] = ok[u] ? fse_entries[desc[4 * min(ch, 63u) + kind] + i] : 0u;  // baseline(16) | nbits(8) | symbol(8)
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const uint32_t baseline = e[u] & 0xFFFF, nb = (e[u] >> 16) & 0xFF, sym = e[u] >> 24;
                const uint32_t c6 = c6k[u] == 2 ? sym : seq_code6((int)c6k[u], sym);
                if (ok[u])
                    cells[idx0 + 256 * u] = c6 == kPipeEscape ? (uint16_t)0 : (uint16_t)(((baseline + n[u]) >> (nb & 31)) | (c6 << 10));
            }
        }
    }
    __syncthreads();

    // wave-uniform trip count; every wavefront computes the same value
    const uint32_t nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(has ? t.n_seq : 0u));
    int status = MZD_OK;

    if (wave == 0) {
        // ================= stage A: the serial chain =================
        const int alL = t.ll_log, alM = t.ml_log, alO = t.of_log;
        // bit window as in SeqBits, with the refill address as a 32-bit offset from the start of the
        // window's front slack (a launch covers < 4 GiB of the blob, slack included: mzd_batch_run cuts the windows)
        const uint8_t *inb = in - MZD_IN_PAD;
        uint64_t C = 0, D = 0;
        uint32_t off = 0;
        int k = 0, rem = 0;
        auto refill = [&]() {
            const int nb = k >> 3, sh = nb * 8;
            C = (C << sh) | ((D >> 1) >> (63 - sh));
            off -= (uint32_t)nb;
            k &= 7;
            asm volatile("" ::"v"((uint32_t)C), "v"((uint32_t)(C >> 32)) : "memory");  // see SeqBits::refill
            D = ld64u(inb + off);
        };
        auto peek = [&](int n) -> uint32_t { return (uint32_t)(((C << k) >> 1) >> (63 - n)); };
        uint32_t sL = 0, sM = 0, sO = 0;
        bool live = has && t.n_seq > 0;
        if (live) {
            SeqBits br;
            rem = br.init(in + t.in_off, (int)t.in_size);
            C = br.C; D = br.D; k = br.k; off = (uint32_t)(br.pd - inb);
            if (rem < 0) {
                status = MZD_ERR_BAD_PADDING;  // sequences.go:141-143
                live = false;
            } else {
                // initial states in the order LL, OF, ML (sequences.go:145-159)
                sL = peek(alL); k += alL;
                sO = peek(alO); k += alO;
                refill();
                sM = peek(alM); k += alM;
                rem -= alL + alO + alM;
                if (rem < 0) { status = MZD_ERR_SEQ_BITS; live = false; }
            }
        }
        const uint32_t sizeL = 1u << alL, sizeM = 1u << alM, sizeO = 1u << alO;
        sL += sizeL; sM += sizeM; sO += sizeO;  // states are kept pre-biased by the table size
        const uint32_t slot = live ? (uint32_t)lane * kSeqCellsPerChain : 0u;
        uint32_t last_i = t.n_seq - 1;
        // parked: limit 0, cell 0 of its slot, cursor 0 = the (readable) front slack of the window; the hot loop's
        // ring check, ring_low <= off - 40 as unsigned numbers, is always true for it
        auto park = [&]() { off = 0; C = D = 0; k = 0; rem = 0; sL = sizeL; sM = sizeM; sO = sizeO; live = false; last_i = 0xFFFFFFFFu; };
        if (!live) park();
        const uint32_t nbL0 = (uint32_t)(alL - 31), nbM0 = (uint32_t)(alM - 31), nbO0 = (uint32_t)(alO - 31);  // nbits = acc_log - 31 + clz(next)
        const uint16_t *cL = cells + slot - sizeL;
        const uint16_t *cM = cells + slot + 512 - sizeM;
        const uint16_t *cO = cells + slot + 1024 - sizeO;

        uint32_t tail_seen = 0;
        uint32_t polls = 0;  // diagnostics (-DMZD_PIPE_PROF prints it): queue-full polls | ring-not-ready polls << 16
#ifdef MZD_PIPE_STATS
        const long long stats_t0 = clock64();
#endif
#ifdef MZD_PIPE_PROF
        long long prof_wait = 0, prof_t0 = clock64(), prof_r0 = wall_clock64();
        if (blockIdx.x == 0 && lane == 0) printf("A: staging + init %lld cycles\n", prof_t0 - prof_k0);
#endif
        auto wait_space = [&](uint32_t at) {  // slot of step `at` is free once at - tail1 < depth
#ifdef MZD_PIPE_PROF
            const long long w0 = clock64();
#endif
            while (at - tail_seen >= (uint32_t)kPipeDepth) {
                tail_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->tail1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (at - tail_seen >= (uint32_t)kPipeDepth) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_PIPE_PROF
            prof_wait += clock64() - w0;
#endif
        };
        (void)wait_space;  // used by the C++ statement of the hot loop only
        // (symbol, next, base | extra << 24) of a literal-length / match-length cell, escape or not
        auto full_cell = [&](int kind, uint32_t x, uint32_t idx, uint32_t toff, uint32_t size, uint32_t &next, uint32_t &ct) {
            next = x & 1023;
            ct = CTc[kind * 64 + (x >> 10)];
            if (next == 0) {  // escape: the symbol is only in the host cell
                const uint32_t e = fse_entries[toff + idx];
                const uint32_t sym = e >> 24;
                next = ((e & 0xFFFF) + size) >> ((e >> 16) & 0xFF);
                ct = kind == 0 ? (c_ll_base[min(sym, 35u)] | ((uint32_t)c_ll_extra[min(sym, 35u)] << 24))
                               : (c_ml_base[min(sym, 52u)] | ((uint32_t)c_ml_extra[min(sym, 52u)] << 24));
            }
        };
        // General step of sequence `idx` for the lanes in `mine` (their queue entries of this step are
        // rewritten as mode 1); the other lanes' entries are already in the slot.
        auto general_step = [&](uint32_t idx, bool mine) {
            const bool lastseq = idx == last_i;
            const uint32_t xl = cL[sL], xm = cM[sM], xo = cO[sO];
            uint32_t nl = 1, nm = 1, cl = 0, cm = 0;
            if (mine) {
                full_cell(0, xl, sL - sizeL, t.ll_off, sizeL, nl, cl);
                full_cell(1, xm, sM - sizeM, t.ml_off, sizeM, nm, cm);
            }
            const uint32_t no = xo & 1023, exO = xo >> 10;
            uint32_t nbL = nbL0 + (uint32_t)__builtin_clz(nl | 1);
            uint32_t nbM = nbM0 + (uint32_t)__builtin_clz(nm | 1);
            uint32_t nbO = nbO0 + (uint32_t)__builtin_clz(no | 1);
            if (lastseq) { nbL = 0; nbM = 0; nbO = 0; }  // sequences.go:178
            const uint32_t exL = cl >> 24, exM = cm >> 24;
            const int total = (int)(exO + exM + exL + nbL + nbM + nbO);
            bool ok = mine;
            if (mine && total > rem) {  // the cursor would pass the start of the stream
                status = MZD_ERR_SEQ_BITS;
                ok = false;
            }
            const uint32_t m = ok ? 0xFFFFFFFFu : 0u;  // lanes that do not step must not move their cursor
            const int wO = (int)(exO & m), wM = (int)(exM & m), wL = (int)(exL & m);
            const int vL = (int)(nbL & m), vM = (int)(nbM & m), vO = (int)(nbO & m);
            const uint32_t ofx = peek(wO); k += wO; refill();
            const uint32_t mlx = peek(wM); k += wM;
            const uint32_t llx = peek(wL); k += wL; refill();
            const uint32_t aL = peek(vL); k += vL;
            const uint32_t aM = peek(vM); k += vM;
            const uint32_t aO = peek(vO); k += vO;
            if (ok) {
                rem -= total;
                sL = (nl << nbL) + aL; sM = (nm << nbM) + aM; sO = (no << nbO) + aO;  // fse.go:282-290
                const uint32_t ofv = min((1u << exO) + ofx, kRecOffSymbolic);  // exO <= 31: no wrap
                shs->q1t[idx % kPipeDepth][lane] = (uint64_t)((cl & 0xFFFFFF) + llx) |
                                                   ((uint64_t)((cm & 0xFFFFFF) + mlx) << kRecMlShift) |
                                                   ((uint64_t)ofv << kRecOffShift);
                shs->q1p[idx % kPipeDepth][lane] = 0x80000000u;
            }
            if (mine && (lastseq || !ok)) {
                if (ok && rem != 0) status = MZD_ERR_SEQ_BITS;  // sequences.go:197-204
                park();
            }
        };

        // ---- the hot loop.  Runs steps until a lane needs the general step (returns the mask of those
        // lanes; their step is NOT done, everybody's queue entry IS written, head1 not yet moved) or
        // nmax is reached.  One step = refill, three cell reads, bit counts, three state fields.
        uint32_t i = 0;
        const uint32_t lane4 = (uint32_t)lane * 4u, lane8 = (uint32_t)lane * 8u, vzero = 0;
        // LDS byte addresses of cL / cM / cO
        const uint32_t cbL = kPipeFixedLds + 2u * (slot - sizeL), cbM = kPipeFixedLds + 2u * (slot + 512 - sizeM),
                       cbO = kPipeFixedLds + 2u * (slot + 1024 - sizeO);
        while (i < nmax) {
            uint64_t smask = 0;
#ifdef MZD_PIPE_CXX_STEP
            do {
                wait_space(i);
                shs->progress[lane] = off;
                const uint32_t xl = cL[sL], xm = cM[sM], xo = cO[sO];
                refill();  // overlaps the LDS latency of the cells
                const uint32_t exO = xo >> 10;
                const uint32_t exL = sub_sat(xl >> 12, 3u), exM = sub_sat(xm >> 12, 7u);
                const uint32_t nl = xl & 1023, nm = xm & 1023, no = xo & 1023;
                const uint32_t nbL = nbL0 + ffbh_raw(nl);  // escape: next = 0 -> clz = -1 -> nbits < 0
                const uint32_t nbM = nbM0 + ffbh_raw(nm);
                const uint32_t nbO = nbO0 + ffbh_raw(no);
                // bit offsets in stream order: OF extra, ML extra, LL extra | LL state, ML state, OF state
                const uint32_t o3 = exO + exM + exL;
                const uint32_t c1 = o3 + nbL, c2 = c1 + nbM, total = c2 + nbO;
                // k stays < 64 (the refill shifts by 8 * (k >> 3)); never past the start of the stream;
                // unsigned: a parked lane has limit 0; an escape makes its nbits negative, and OR-ing
                // them in keeps bit 31 set even if the sum wrapped back
                const bool go = (total | nbL | nbM) <= (uint32_t)min(63 - k, rem);
                const bool last = i == last_i;  // never true for a parked lane (last_i = ~0)
                const uint64_t T = C << k;
                const uint32_t X = (uint32_t)((T << o3) >> 32);  // the <= 26 state bits start at bit 31
                const uint32_t tb = 32 + o3;
                const uint32_t aL = __builtin_amdgcn_ubfe(X, tb - c1, nbL);
                const uint32_t aM = __builtin_amdgcn_ubfe(X, tb - c2, nbM);
                const uint32_t aO = __builtin_amdgcn_ubfe(X, tb - total, nbO);
                const bool adv = go && !last;
                const int n = adv ? (int)total : 0;
                sL = adv ? (nl << nbL) + aL : sL;
                sM = adv ? (nm << nbM) + aM : sM;
                sO = adv ? (no << nbO) + aO : sO;
                k += n;
                rem -= n;
                shs->q1t[i % kPipeDepth][lane] = T;
                shs->q1p[i % kPipeDepth][lane] =
                    __builtin_amdgcn_perm(xo, __builtin_amdgcn_perm(xm, xl, 0x0c0c0501u), 0x0c050100u);
                smask = __builtin_amdgcn_ballot_w64(last || (live && !go));
                i++;
                if (!smask) {
                    asm volatile("" ::: "memory");
                    __hip_atomic_store(&shs->head1, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } while (i < nmax && !smask);
#else
            {
                // The same step, hand-scheduled: as a lone wavefront pays ~4.4 cycles per instruction of any
                // kind, the instruction count IS the step latency (~68 here; hipcc's version of the C++
                // statement above: ~110).
                // REFILL from LDS: a per-lane gather of the bitstream from global memory (57 distinct lines
                // per step) was what bounded the step at 57 chains -- each 128-byte line was fetched ~40
                // times.  Wave P now keeps 128 bytes of every chain's bitstream in an LDS ring (one 32-byte
                // load per chain every ~10 steps) and a step reads the 8 bytes below its window from the ring
                // (byte offset & 127; the ring repeats its first 8 bytes at the end), merged into the window
                // one step later, in the shadow of that step's cell reads.  Once per batch of four steps (and
                // at every entry) the lanes check that P is at least 40 bytes ahead of them.  Two register
                // pairs alternate (v[232:233], v[234:235]).
                // The loop body is the step EIGHT times, one instance per queue slot: the slot addresses are
                // immediates, queue space is checked and the cursor published to wave P once per batch of four
                // (stage B consumes whole batches), head1 is published and nmax checked at the end of a batch
                // (so i may overshoot nmax by up to 3 steps of parked lanes, inside a batch whose slots are
                // known to be free).  The last sequence of a lane is a "no go" through the per-lane countdown
                // `left`.  Temporaries are fixed registers v200..v235 / s86.
                static_assert(kPipeDepth == 8 && kPipeBatch == 4 && kPipeRing == 128, "the unrolled loop assumes 2 batches of 4 slots, a 128-byte ring");
                const uint64_t livemask = __builtin_amdgcn_ballot_w64(live);
                const uint32_t sel1 = 0x0c0c0501u, sel2 = 0x0c050100u;
                uint32_t sLb = sL, sMb = sM, sOb = sO;  // the states alternate between two register sets
                uint32_t left = last_i - i;  // steps before the lane's last sequence (parked lane: huge)
                uint32_t rem1 = (uint32_t)rem + 1u;
                uint32_t Dlo = (uint32_t)D, Dhi = (uint32_t)(D >> 32);
                const uint32_t ringl = 512u + (uint32_t)offsetof(PipeShared, ring) + (uint32_t)lane * (kPipeRing + 8);
// the cursor goes to wave P, then: queue space for the batch, and the ring at least 40 bytes below the cursor
#define MZD_PIPE_RINGCHK(TAG)                                                                               \
    "ds_write_b32 %[lane4], %[off] offset:%[o_prog]\n"                                                      \
    "L_pipe_ring" TAG "_%=:\n\t"                                                                            \
    "ds_read_b32 v200, %[lane4] offset:%[o_rlow]\n\t"                                                       \
    "v_add_u32 v201, -40, %[off]\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
    "v_cmp_gt_u32 vcc, v200, v201\n\t"                                                                      \
    "s_cbranch_vccz L_pipe_go" TAG "_%=\n\t"                                                                \
    "s_add_u32 %[polls], %[polls], 0x10000\n\t"                                                             \
    "s_sleep 1\n\t"                                                                                         \
    "s_branch L_pipe_ring" TAG "_%=\n"
#define MZD_PIPE_CHECK(TAG)                                                                                 \
    "L_pipe_top" TAG "_%=:\n\t"                                                                             \
    "s_sub_u32 s86, %[i], %[tail]\n\t"                                                                      \
    "s_cmp_lt_u32 s86, 5\n\t" /* i + 3 - tail1 < depth */                                                   \
    "s_cbranch_scc1 L_pipe_spc" TAG "_%=\n"                                                                 \
    "L_pipe_poll" TAG "_%=:\n\t"                                                                            \
    "ds_read_b32 v200, %[vzero] offset:%[o_tail1]\n\t"                                                      \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                              \
    "v_readfirstlane_b32 %[tail], v200\n\t"                                                                 \
    "s_sub_u32 s86, %[i], %[tail]\n\t"                                                                      \
    "s_cmp_lt_u32 s86, 5\n\t"                                                                               \
    "s_cbranch_scc1 L_pipe_spc" TAG "_%=\n\t"                                                               \
    "s_add_u32 %[polls], %[polls], 1\n\t"                                                                   \
    "s_sleep 1\n\t"                                                                                         \
    "s_branch L_pipe_poll" TAG "_%=\n"                                                                      \
    "L_pipe_spc" TAG "_%=:\n\t"                                                                             \
    /* fast path: ring_low as read during the previous step (v236; it only ever decreases) */               \
    "v_add_u32 v201, -40, %[off]\n\t"                                                                       \
    "s_waitcnt lgkmcnt(0)\n\t" /* v236 was read a step ago */                                               \
    "v_cmp_gt_u32 vcc, v236, v201\n\t"                                                                      \
    "ds_write_b32 %[lane4], %[off] offset:%[o_prog]\n\t"                                                    \
    "s_cbranch_vccz L_pipe_go" TAG "_%=\n\t"                                                                \
    MZD_PIPE_RINGCHK(TAG)
// The step, ordered so that the LDS round trip of the NEXT step's cell reads runs behind this step's bookkeeping: a
// lone wavefront issues one instruction per ~4 cycles and nothing while it waits, so every instruction placed between
// the reads and their s_waitcnt is latency hidden.  On entry the three cells of this step are on their way (requested
// at the end of the step before, or by the prologue), the window C is normalised (k < 8), v228 holds the limit.
//   1. the recurrence: cells -> bit counts -> state fields -> new states -> the next step's cell reads (speculative: a
//      lane that does not "go" reads with a meaningless state; LDS reads outside the allocation return zero);
//   2. in their shadow: go / advance, the queue entry for stage B, then what used to open the next step: cursor,
//      ring read, refill of C with the bytes the step before read from the ring (DM), the next limit.
// DM: the 8 bytes the previous step read from the ring; DL: where this step's go
#ifdef MZD_ABL_NOWAIT  /* ablations: timing experiments only, wrong results */
#define MZD_ABL_W3 "s_nop 0\n\t"
#define MZD_ABL_W6 "s_nop 0\n\t"
#else
#define MZD_ABL_W3 "s_waitcnt lgkmcnt(3)\n\t"
#define MZD_ABL_W6 "s_waitcnt lgkmcnt(6)\n\t"
#endif
#ifdef MZD_ABL_NORING
#define MZD_ABL_RING(DL) "s_nop 0\n\t"
#else
#define MZD_ABL_RING(DL) "ds_read_b64 " DL ", v209\n\t"
#endif
#ifdef MZD_ABL_NOQW
#define MZD_ABL_QW(X) "s_nop 0\n\t"
#else
#define MZD_ABL_QW(X) X
#endif
#define MZD_PIPE_STEP(DM, DL, SA, SB, TAG, QT, QP, OUT, RLOW)                                               \
    "L_pipe_go" TAG "_%=:\n\t"                                                                              \
    MZD_ABL_W3                              /* the three cells (behind them: two queue writes, a ring read) */ \
    "v_lshrrev_b32 v215, 12, v203\n\t"                                                                      \
    "v_lshrrev_b32 v216, 12, v204\n\t"                                                                      \
    "v_and_b32 v217, 0x3ff, v203\n\t"       /* nl */                                                        \
    "v_and_b32 v218, 0x3ff, v204\n\t"       /* nm */                                                        \
    "v_and_b32 v219, 0x3ff, v205\n\t"       /* no */                                                        \
    "v_lshrrev_b32 v214, 10, v205\n\t"      /* exO */                                                       \
    "v_ffbh_u32 v220, v217\n\t"                                                                             \
    "v_ffbh_u32 v221, v218\n\t"                                                                             \
    "v_ffbh_u32 v222, v219\n\t"                                                                             \
    "v_sub_u32_e64 v215, v215, 3 clamp\n\t" /* exL */                                                       \
    "v_sub_u32_e64 v216, v216, 7 clamp\n\t" /* exM */                                                       \
    "v_add_u32 v220, v220, %[nbL0]\n\t"     /* nbL */                                                       \
    "v_add_u32 v221, v221, %[nbM0]\n\t"     /* nbM */                                                       \
    "v_add_u32 v222, v222, %[nbO0]\n\t"     /* nbO */                                                       \
    "v_add3_u32 v223, v214, v216, v215\n\t" /* o3 = exO + exM + exL */                                      \
    /* field positions in X.hi as NEGATED running sums (v_bfe_u32 takes the offset mod 32): -nbL, ... */    \
    "v_sub_u32 v224, 0, v220\n\t"           /* -nbL */                                                      \
    "v_add_u32 v229, v223, %[k]\n\t"        /* k + o3 */                                                    \
    "v_sub_u32 v225, v224, v221\n\t"        /* -(nbL + nbM) */                                              \
    "v_lshlrev_b64 v[210:211], v229, %[C]\n\t"          /* X = C << (k + o3): state bits from bit 63 */     \
    "v_sub_u32 v226, v225, v222\n\t"        /* -(nbL + nbM + nbO) */                                        \
    "v_perm_b32 v231, v204, v203, %[sel1]\n\t"                                                              \
    "v_bfe_u32 v224, v211, v224, v220\n\t" /* aL */                                                         \
    "v_bfe_u32 v225, v211, v225, v221\n\t" /* aM */                                                         \
    "v_sub_u32 v230, v223, v226\n\t"        /* total */                                                     \
    "v_bfe_u32 v226, v211, v226, v222\n\t" /* aO */                                                         \
    /* the new states go to the OTHER register set (a lane that does not advance is special: the exit     */ \
    /* code picks per lane; a parked lane's state is never used)                                          */ \
    "v_lshl_add_u32 %[sL" SB "], v217, v220, v224\n\t"                                                      \
    "v_lshl_add_u32 %[sM" SB "], v218, v221, v225\n\t"                                                      \
    "v_lshl_add_u32 %[sO" SB "], v219, v222, v226\n\t"                                                      \
    "v_perm_b32 v231, v205, v231, %[sel2]\n\t"          /* (the cells' high bytes for stage B: before the reads below overwrite them) */ \
    "v_or3_b32 v227, v230, v220, v221\n\t"                                                                  \
    "v_lshl_add_u32 v200, %[sL" SB "], 1, %[cbL]\n\t"                                                       \
    "v_lshl_add_u32 v201, %[sM" SB "], 1, %[cbM]\n\t"                                                       \
    "v_lshl_add_u32 v202, %[sO" SB "], 1, %[cbO]\n\t"                                                       \
    "ds_read_u16 v203, v200\n\t" /* the NEXT step's xl */                                                   \
    "ds_read_u16 v204, v201\n\t" /* xm */                                                                   \
    "ds_read_u16 v205, v202\n\t" /* xo */                                                                   \
    /* ---- in the shadow of those reads */                                                                 \
    "v_cmp_lt_u32 vcc, v227, v228\n\t"                  /* go (= advance; never at the last sequence) */    \
    "v_lshlrev_b64 v[212:213], %[k], %[C]\n\t"          /* T = C << k */                                    \
    "s_andn2_b64 %[smask], %[live], vcc\n\t"            /* special = live & ~go */                          \
    "v_cndmask_b32 v230, 0, v230, vcc\n\t"                                                                  \
    MZD_ABL_QW("ds_write_b64 %[lane8], v[212:213] offset:" QT "\n\t")                                       \
    "v_sub_u32 %[rem1], %[rem1], v230\n\t"                                                                  \
    "v_add_u32 %[k], %[k], v230\n\t"                                                                        \
    MZD_ABL_QW("ds_write_b32 %[lane4], v231 offset:" QP "\n\t")                                             \
    "s_add_u32 %[i], %[i], 1\n\t"                                                                           \
    "s_cmp_lg_u64 %[smask], 0\n\t"                                                                          \
    "s_cbranch_scc1 " OUT "\n\t"                                                                            \
    /* the cursor and the window for the next step: C <<= 8 * (k >> 3); k &= 7; the bytes that come in from DM */ \
    "v_lshrrev_b32 v207, 3, %[k]\n\t"                                                                       \
    "v_and_b32 v206, -8, %[k]\n\t"                                                                          \
    "v_sub_u32 %[off], %[off], v207\n\t"                                                                    \
    "v_and_b32 %[k], 7, %[k]\n\t"                                                                           \
    "v_and_b32 v209, 127, %[off]\n\t"                                                                       \
    "v_sub_u32 v208, 63, v206\n\t"                                                                          \
    "v_add_u32 v209, v209, %[ringl]\n\t"                                                                    \
    "v_lshlrev_b64 %[C], v206, %[C]\n\t"                                                                    \
    MZD_ABL_RING(DL)               /* the 8 bytes below the new window, for the refill after the next step */ \
    MZD_ABL_W6                     /* DM: everything older than the six operations of this step */          \
    "v_lshrrev_b64 v[210:211], 1, " DM "\n\t"                                                               \
    "v_sub_u32 v228, 64, %[k]\n\t"                                                                          \
    "v_lshrrev_b64 v[210:211], v208, v[210:211]\n\t"                                                        \
    "v_min3_u32 v228, v228, %[rem1], %[left]\n\t" /* limit = min(64 - k, rem + 1, steps before the last) */ \
    "v_lshl_add_u64 %[C], %[C], 0, v[210:211]\n\t"                                                          \
    "v_add_u32 %[left], -1, %[left]\n\t"                                                                    \
    RLOW
#define MZD_PIPE_PUBLISH(OUT)                                                                               \
    "v_mov_b32 v202, %[i]\n\t"                                                                              \
    "ds_write_b32 %[vzero], v202 offset:%[o_head1]\n\t"                                                     \
    "s_cmp_lt_u32 %[i], %[nmax]\n\t"                                                                        \
    "s_cbranch_scc0 " OUT "\n\t"
#define MZD_OUTE "L_pipe_oute_%="
#define MZD_OUTO "L_pipe_outo_%="
#define MZD_RLOW "ds_read_b32 v236, %[lane4] offset:%[o_rlow]\n\t" /* for the next batch's ring check */
#define MZD_DA "v[232:233]"
#define MZD_DB "v[234:235]"
                asm volatile(
                    // prologue = what the shadow of a step before would have done: the ring holds the bytes at the cursor
                    // (checked first: the very first entry, or a general step that moved the cursor far), cursor and window
                    // normalised and refilled from D (which the C++ side keeps valid), both lookahead pairs = the 8 bytes below
                    // the new window, the limit, and this step's cells requested; then the instance of slot i % 8
                    "v_mov_b32 v236, -1\n\t"  // no ring_low read ahead yet: the first batch check takes the slow path
                    MZD_PIPE_RINGCHK("e")
                    "L_pipe_goe_%=:\n\t"
                    "v_lshrrev_b32 v207, 3, %[k]\n\t"
                    "v_and_b32 v206, -8, %[k]\n\t"
                    "v_sub_u32 %[off], %[off], v207\n\t"
                    "v_and_b32 %[k], 7, %[k]\n\t"
                    "v_and_b32 v209, 127, %[off]\n\t"
                    "v_sub_u32 v208, 63, v206\n\t"
                    "v_add_u32 v209, v209, %[ringl]\n\t"
                    "v_lshlrev_b64 %[C], v206, %[C]\n\t"
                    "ds_read_b64 v[232:233], v209\n\t"
                    "ds_read_b64 v[234:235], v209\n\t"
                    "v_mov_b32 v210, %[Dlo]\n\t"
                    "v_mov_b32 v211, %[Dhi]\n\t"
                    "v_lshrrev_b64 v[210:211], 1, v[210:211]\n\t"
                    "v_sub_u32 v228, 64, %[k]\n\t"
                    "v_lshrrev_b64 v[210:211], v208, v[210:211]\n\t"
                    "v_min3_u32 v228, v228, %[rem1], %[left]\n\t"
                    "v_lshl_add_u64 %[C], %[C], 0, v[210:211]\n\t"
                    "v_add_u32 %[left], -1, %[left]\n\t"
                    "v_lshl_add_u32 v200, %[sLa], 1, %[cbL]\n\t"
                    "v_lshl_add_u32 v201, %[sMa], 1, %[cbM]\n\t"
                    "v_lshl_add_u32 v202, %[sOa], 1, %[cbO]\n\t"
                    "ds_read_u16 v203, v200\n\t"
                    "ds_read_u16 v204, v201\n\t"
                    "ds_read_u16 v205, v202\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_and_b32 s86, %[i], 7\n\t"
                    "s_cmp_eq_u32 s86, 0\n\t"
                    "s_cbranch_scc1 L_pipe_top0_%=\n\t"
                    "s_cmp_eq_u32 s86, 1\n\t"
                    "s_cbranch_scc1 L_pipe_go1_%=\n\t"
                    "s_cmp_eq_u32 s86, 2\n\t"
                    "s_cbranch_scc1 L_pipe_go2_%=\n\t"
                    "s_cmp_eq_u32 s86, 3\n\t"
                    "s_cbranch_scc1 L_pipe_go3_%=\n\t"
                    "s_cmp_eq_u32 s86, 4\n\t"
                    "s_cbranch_scc1 L_pipe_top4_%=\n\t"
                    "s_cmp_eq_u32 s86, 5\n\t"
                    "s_cbranch_scc1 L_pipe_go5_%=\n\t"
                    "s_cmp_eq_u32 s86, 6\n\t"
                    "s_cbranch_scc1 L_pipe_go6_%=\n\t"
                    "s_branch L_pipe_go7_%=\n"
                    MZD_PIPE_CHECK("0")
                    MZD_PIPE_STEP(MZD_DA, MZD_DB, "a", "b", "0", "%[qt0]", "%[qp0]", MZD_OUTE, "")
                    MZD_PIPE_STEP(MZD_DB, MZD_DA, "b", "a", "1", "%[qt1]", "%[qp1]", MZD_OUTO, "")
                    MZD_PIPE_STEP(MZD_DA, MZD_DB, "a", "b", "2", "%[qt2]", "%[qp2]", MZD_OUTE, "")
                    MZD_PIPE_STEP(MZD_DB, MZD_DA, "b", "a", "3", "%[qt3]", "%[qp3]", MZD_OUTO, MZD_RLOW)
                    MZD_PIPE_PUBLISH(MZD_OUTO)
                    MZD_PIPE_CHECK("4")
                    MZD_PIPE_STEP(MZD_DA, MZD_DB, "a", "b", "4", "%[qt4]", "%[qp4]", MZD_OUTE, "")
                    MZD_PIPE_STEP(MZD_DB, MZD_DA, "b", "a", "5", "%[qt5]", "%[qp5]", MZD_OUTO, "")
                    MZD_PIPE_STEP(MZD_DA, MZD_DB, "a", "b", "6", "%[qt6]", "%[qp6]", MZD_OUTE, "")
                    MZD_PIPE_STEP(MZD_DB, MZD_DA, "b", "a", "7", "%[qt7]", "%[qp7]", MZD_OUTO, MZD_RLOW)
                    MZD_PIPE_PUBLISH(MZD_OUTO)
                    "s_branch L_pipe_top0_%=\n"
                    "L_pipe_oute_%=:\n\t"  // left after an even slot: the new states are in set b
                    "v_cndmask_b32 %[sLa], %[sLa], %[sLb], vcc\n\t"  // vcc is still the last step's "go"
                    "v_cndmask_b32 %[sMa], %[sMa], %[sMb], vcc\n\t"
                    "v_cndmask_b32 %[sOa], %[sOa], %[sOb], vcc\n\t"
                    "s_branch L_pipe_done_%=\n"
                    "L_pipe_outo_%=:\n\t"  // after an odd slot: old states in set b, new ones in set a
                    "v_cndmask_b32 %[sLa], %[sLb], %[sLa], vcc\n\t"
                    "v_cndmask_b32 %[sMa], %[sMb], %[sMa], vcc\n\t"
                    "v_cndmask_b32 %[sOa], %[sOb], %[sOa], vcc\n"
                    "L_pipe_done_%=:\n\t"
                    // the C++ side's lookahead: the 8 bytes below the (not yet normalised) window, from memory
                    "s_waitcnt lgkmcnt(0)\n\t"  // the last step's ring read may still be on its way into these registers
                    "global_load_dwordx2 v[232:233], %[off], %[inb]\n\t"
                    "s_waitcnt vmcnt(0)\n\t"
                    "v_mov_b32 %[Dlo], v232\n\t"
                    "v_mov_b32 %[Dhi], v233\n\t"
                    : [sLa] "+v"(sL), [sMa] "+v"(sM), [sOa] "+v"(sO), [sLb] "+v"(sLb), [sMb] "+v"(sMb), [sOb] "+v"(sOb), [k] "+v"(k),
                      [rem1] "+v"(rem1), [left] "+v"(left), [off] "+v"(off), [C] "+v"(C), [Dlo] "+v"(Dlo), [Dhi] "+v"(Dhi), [i] "+s"(i),
                      [tail] "+s"(tail_seen), [polls] "+s"(polls), [smask] "=&s"(smask)
                    : [cbL] "v"(cbL), [cbM] "v"(cbM), [cbO] "v"(cbO), [nbL0] "v"(nbL0), [nbM0] "v"(nbM0), [nbO0] "v"(nbO0),
                      [lane4] "v"(lane4), [lane8] "v"(lane8), [vzero] "v"(vzero), [ringl] "v"(ringl), [nmax] "s"(nmax),
                      [live] "s"(livemask), [inb] "s"(inb), [sel1] "s"(sel1), [sel2] "s"(sel2),
                      [o_tail1] "n"(512 + offsetof(PipeShared, tail1)), [o_head1] "n"(512 + offsetof(PipeShared, head1)),
                      [o_prog] "n"(512 + offsetof(PipeShared, progress)), [o_rlow] "n"(512 + offsetof(PipeShared, ring_low)),
#define MZD_QT(S) (512 + offsetof(PipeShared, q1t) + (S) * 512)
#define MZD_QP(S) (512 + offsetof(PipeShared, q1p) + (S) * 256)
                      [qt0] "n"(MZD_QT(0)), [qt1] "n"(MZD_QT(1)), [qt2] "n"(MZD_QT(2)), [qt3] "n"(MZD_QT(3)),
                      [qt4] "n"(MZD_QT(4)), [qt5] "n"(MZD_QT(5)), [qt6] "n"(MZD_QT(6)), [qt7] "n"(MZD_QT(7)),
                      [qp0] "n"(MZD_QP(0)), [qp1] "n"(MZD_QP(1)), [qp2] "n"(MZD_QP(2)), [qp3] "n"(MZD_QP(3)),
                      [qp4] "n"(MZD_QP(4)), [qp5] "n"(MZD_QP(5)), [qp6] "n"(MZD_QP(6)), [qp7] "n"(MZD_QP(7))
                    : "memory", "vcc", "scc", "s86",
                      "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213",
                      "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226",
                      "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236");
#undef MZD_PIPE_STEP
#undef MZD_PIPE_CHECK
#undef MZD_PIPE_RINGCHK
#undef MZD_PIPE_PUBLISH
#undef MZD_OUTE
#undef MZD_OUTO
#undef MZD_RLOW
#undef MZD_DA
#undef MZD_DB
#undef MZD_QT
#undef MZD_QP
                rem = (int)(rem1 - 1u);
                D = (uint64_t)Dlo | ((uint64_t)Dhi << 32);
            }
#endif
            // i has moved past the step; lanes in smask have not done it yet
            if (smask) general_step(i - 1, ((smask >> lane) & 1) != 0);
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->head1, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef MZD_PIPE_STATS
        if (lane == 0) {
            atomicAdd(&g_pipe_stats[0], 1ull);
            atomicAdd(&g_pipe_stats[1], (unsigned long long)nmax);
            atomicAdd(&g_pipe_stats[2], (unsigned long long)(clock64() - stats_t0));
            atomicAdd(&g_pipe_stats[3], (unsigned long long)(polls & 0xFFFF));
            atomicAdd(&g_pipe_stats[4], (unsigned long long)(polls >> 16));
        }
#endif
#ifdef MZD_PIPE_PROF
        if (blockIdx.x == 0 && lane == 0)
            printf("A: steps %u cycles %lld wait %lld real(100MHz) %lld queue-full polls %u ring polls %u\n", nmax, clock64() - prof_t0,
                   prof_wait, wall_clock64() - prof_r0, polls & 0xFFFF, polls >> 16);
        (void)polls;
#endif
    } else if (wave == 1) {
        // ================= stage B: field extraction and values, four steps at a time =================
        uint32_t head_seen = 0, tail_seen = 0;
#ifdef MZD_PIPE_PROF
        long long prof_wait = 0, prof_wait2 = 0, prof_t0 = clock64();
#endif
        for (uint32_t j0 = 0; j0 < nmax; j0 += kPipeBatch) {
            const uint32_t need = min(j0 + (uint32_t)kPipeBatch, nmax);
#ifdef MZD_PIPE_PROF
            const long long w0 = clock64();
#endif
            while (head_seen < need) {
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->head1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head_seen < need) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_PIPE_PROF
            prof_wait += clock64() - w0;
#endif
            asm volatile("" ::: "memory");
            uint64_t T[kPipeBatch];
            uint32_t P[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                T[u] = shs->q1t[(j0 + u) % kPipeDepth][lane];
                P[u] = shs->q1p[(j0 + u) % kPipeDepth][lane];
            }
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->tail1, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint64_t q[kPipeBatch];
#ifdef MZD_EXP_FAST_BC  // timing experiment only (wrong results): what stage A can do when nothing holds it up
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) q[u] = T[u] ^ P[u];
#else
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                const uint32_t cl = CTc[__builtin_amdgcn_ubfe(P[u], 2, 6)];
                const uint32_t cm = CTc[64 + __builtin_amdgcn_ubfe(P[u], 10, 6)];
                const uint32_t exO = __builtin_amdgcn_ubfe(P[u], 18, 6);
                const uint32_t hi = (uint32_t)(T[u] >> 32);
                const uint32_t exL = cl >> 24, exM = cm >> 24;
                const uint32_t ofx = __builtin_amdgcn_ubfe(hi, 32u - exO, exO);
                const uint32_t Y = (uint32_t)((T[u] << exO) >> 32);
                const uint32_t mlx = __builtin_amdgcn_ubfe(Y, 32u - exM, exM);
                const uint32_t llx = __builtin_amdgcn_ubfe(Y, 32u - exM - exL, exL);
                const uint32_t ofv = min((1u << exO) + ofx, kRecOffSymbolic);  // exO <= 31: no wrap
                const uint64_t v = (uint64_t)((cl & 0xFFFFFF) + llx) | ((uint64_t)((cm & 0xFFFFFF) + mlx) << kRecMlShift) |
                                   ((uint64_t)ofv << kRecOffShift);
                q[u] = (P[u] >> 31) ? T[u] : v;
            }
#endif
#ifdef MZD_PIPE_PROF
            const long long w1 = clock64();
#endif
            while (j0 + (uint32_t)kPipeBatch - tail_seen > (uint32_t)kPipeDepth) {
                tail_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->tail2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (j0 + (uint32_t)kPipeBatch - tail_seen > (uint32_t)kPipeDepth) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_PIPE_PROF
            prof_wait2 += clock64() - w1;
#endif
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) shs->q2[(j0 + u) % kPipeDepth][lane] = q[u];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->head2, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#ifdef MZD_PIPE_PROF
        if (blockIdx.x == 0 && lane == 0)
            printf("B: cycles %lld wait_in %lld wait_out %lld\n", clock64() - prof_t0, prof_wait, prof_wait2);
#endif
    } else if (wave == 2) {
        // ================= stage C: sums, offset history, records =================
        // Branch-free per sequence: errors are sticky flags (a failed block's records, sums and history
        // are never used), the history update is a chain of selects, only the record store is masked.
        int h0, h1, h2;
        if (t.hist_known) { h0 = 1; h1 = 4; h2 = 8; }  // framedecompressor.go:48,59
        else { h0 = -1; h1 = -2; h2 = -3; }
        uint32_t litPos = 0, outPos = 0;
        uint32_t err_unsup = 0, err_off = 0, err_size = 0;
        uint64_t *myrec = recs + t.rec_off;
        TileBase *mytile = tiles + t.tile_off;
        const uint32_t my_n = has ? t.n_seq : 0u;
        uint32_t head_seen = 0;
#ifdef MZD_PIPE_PROF
        long long prof_wait = 0, prof_t0 = clock64();
#endif
        for (uint32_t j0 = 0; j0 < nmax; j0 += kPipeBatch) {
            const uint32_t need = min(j0 + (uint32_t)kPipeBatch, nmax);
#ifdef MZD_PIPE_PROF
            const long long w0 = clock64();
#endif
            while (head_seen < need) {
                head_seen = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&shs->head2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head_seen < need) __builtin_amdgcn_s_sleep(1);
            }
#ifdef MZD_PIPE_PROF
            prof_wait += clock64() - w0;
#endif
            asm volatile("" ::: "memory");
            uint64_t q[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) q[u] = shs->q2[(j0 + u) % kPipeDepth][lane];
            asm volatile("" ::: "memory");
            __hip_atomic_store(&shs->tail2, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef MZD_EXP_FAST_BC
            litPos += (uint32_t)(q[0] ^ q[1] ^ q[2] ^ q[3]);
            continue;
#endif
            if ((j0 & 63) == 0 && j0 < my_n) mytile[j0 >> 6] = TileBase{litPos, outPos};
            uint64_t rr[kPipeBatch];
#pragma unroll
            for (int u = 0; u < kPipeBatch; u++) {
                const uint32_t j = j0 + u;
                const bool act = j < my_n;
                const uint32_t lo = (uint32_t)q[u], hi = (uint32_t)(q[u] >> 32);
                const uint32_t LL = lo & kRecLlMask;
                const uint32_t ML = __builtin_amdgcn_alignbit(hi, lo, kRecMlShift) & kRecMlMask;
                const uint32_t ofv = hi >> (kRecOffShift - 32);
                // 0 = not active (history untouched), 1..4 = repeat cases 0..3, 5 = new offset
                uint32_t idx = ofv > 3 ? 5u : ofv + (LL == 0 ? 1u : 0u);
                idx = act ? idx : 0u;
                int off = (int)(ofv - 3);                 // idx 5
                off = idx == 4 ? hist_dec(h0) : off;      // sequence_execution.go:65-114
                off = idx == 3 ? h2 : off;
                off = idx == 2 ? h1 : off;
                off = idx <= 1 ? h0 : off;
                h2 = idx >= 3 ? h1 : h2;
                h1 = idx >= 2 ? h0 : h1;
                h0 = idx >= 2 ? off : h0;
                err_unsup |= act && ofv >= kRecOffSymbolic;  // offset value >= 2^28
                err_off |= act && off == 0;
                litPos += act ? LL : 0u;
                outPos += act ? LL + ML : 0u;
                err_size |= outPos > kBlockMax;  // a block regenerates <= 128 KiB
                const uint32_t offfield = off > 0 ? (uint32_t)off : (kRecOffSymbolic | (uint32_t)(-off - 1));
                rr[u] = (uint64_t)lo | ((uint64_t)((hi & ((1u << (kRecOffShift - 32)) - 1)) | (offfield << (kRecOffShift - 32))) << 32);
            }
            // the batch's records: two 16-byte stores per lane instead of four 8-byte ones (every store is a
            // scatter over the chains' record streams through the CU's one address path)
            if (j0 + (uint32_t)kPipeBatch <= my_n) {
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
                *(u64x2 *)(myrec + j0) = u64x2{rr[0], rr[1]};
                *(u64x2 *)(myrec + j0 + 2) = u64x2{rr[2], rr[3]};
            } else {
#pragma unroll
                for (int u = 0; u < kPipeBatch; u++)
                    if (j0 + u < my_n) myrec[j0 + u] = rr[u];
            }
        }
        status = err_unsup ? MZD_ERR_UNSUPPORTED : (err_off ? MZD_ERR_OFFSET : (err_size ? MZD_ERR_CORRUPT_SIZES : MZD_OK));
        if (has && t.n_seq > 0) {
            BlockSum *bs = &sums[t.block];
            bs->lit_total = litPos;
            bs->out_total = outPos;
            bs->hist[0] = h0;
            bs->hist[1] = h1;
            bs->hist[2] = h2;
        }
        shs->stC[lane] = status;
#ifdef MZD_PIPE_PROF
        if (blockIdx.x == 0 && lane == 0) printf("C: cycles %lld wait_in %lld\n", clock64() - prof_t0, prof_wait);
#endif
    } else {
        // ================= wave P: the chains' bitstreams, ahead of stage A =================
        // Keeps the 128 bytes around every chain's cursor in the chain's LDS ring, 32-byte units at (offset & 127):
        // the unit [low - 32, low) may replace [low + 96, low + 128) once A's published cursor is <= low + 88 (A
        // reads nothing at or above cursor + 8); ring_low tells A how far down the ring reaches.  (The prefetch
        // touches far below the cursor are wave B's.)
        const uint8_t *inb = in - MZD_IN_PAD;
        const uint8_t *sbase = in + t.in_off;
        const bool work = has && t.n_seq > 0;
        int low = (int)t.in_size;  // prefetch touches: everything at or above `low` has been requested
        constexpr int kAhead = MZD_PIPE_AHEAD, kLine = 128;
        uint32_t rlow = ((uint32_t)t.in_off + MZD_IN_PAD + t.in_size + 31u) & ~31u;  // ring: nothing yet
        uint8_t *ring = shs->ring[lane];
        uint32_t iter = 0;
        for (;;) {
            const uint32_t hd = (uint32_t)__builtin_amdgcn_readfirstlane(
                (int)__hip_atomic_load(&shs->head1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            // parked lanes point outside the stream
            const uint32_t raw = __hip_atomic_load(&shs->progress[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int cur = (int)(raw - ((uint32_t)t.in_off + MZD_IN_PAD));  // A publishes its refill offset from in - MZD_IN_PAD
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
            {  // prefetch touches after the ring work (before it: 28.78 vs 28.62 ms): they are HBM misses by design and P
               // WAITS for each -- unthrottled touches (from a wavefront that never waits) crowd the CU's miss path: 30.7 ms
                const int target = inside ? max(cur - kAhead, 0) : low;
                for (int g = 0; g < MZD_PIPE_TOUCHES && has && low > target && (iter & MZD_PIPE_TOUCH_EVERY) == 0; g++) {
                    low = max(low - kLine, 0);
                    touch_line(sbase + (low & ~3));
                }
            }
            if (hd >= nmax) break;
            iter++;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    // decode-stage errors come first, as in the reference, where DecodeSequences runs to its end
    // before ExecuteSequences starts
    if (wave == 0 && has && t.n_seq > 0) {
        int st = status;
        if (st == MZD_OK) st = shs->stC[lane];
        if (st != MZD_OK) atomicCAS(&sums[t.block].status, MZD_OK, st);
    }
}

#endif  // MZD_TEST_KERNELS (k_seq_pipe)

}  // namespace mzd
