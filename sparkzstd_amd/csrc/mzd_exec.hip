// mzd_exec.hip -- k_exec: sequence execution with a workgroup per frame (sequence_execution.go:14-63, ringbuffer.go:102-277): what frames
// of 4 GiB and more take (the other executors keep frame positions in 32 bits), exec_variant 1.  Split out of mzd_kernels.hip in round 6.
#pragma once

namespace mzd {

// ------------------------------------------------------------------------------------------
// k_exec: sequence execution + Raw/RLE blocks.  One workgroup per frame, several per CU.
//
// The window (ringbuffer.go) is split in two: the CHUNK of the block currently being regenerated
// lives in LDS (cap bytes, chunk boundaries fall on 64-sequence tile boundaries), everything older
// is final and already in the frame's HBM slab.  LDS layout:
//      [ chunk buffer cap + 32 ][ validity bitmap cap / 8 + 16 ][ control ]
//   * a match whose source lies entirely before the chunk reads HBM/L2 with plain unaligned
//     16-byte loads: nothing to wait for;
//   * inside the chunk, execution is a DATAFLOW: bit p of the bitmap says "output byte p of the
//     chunk is written"; a match copy runs as soon as exactly its source bytes are valid, so
//     64-sequence tiles execute on all wavefronts with no ordering between tiles and no false
//     dependencies (the reference's serial loop sequence_execution.go:16-53 is the degenerate
//     schedule of the same graph).  Progress: the earliest unexecuted match of a chunk always has
//     all its sources valid and every wavefront walks its tiles in increasing order;
//   * a small LDS footprint keeps several frames resident per CU, which is what hides the
//     dependency-chain latency of each one;
//   * byte-misaligned LDS dword READS are replayed 64x on gfx950 (tools/ubench), misaligned dword
//     WRITES are not: copies read aligned dwords, funnel-shift with v_alignbyte, write misaligned;
//   * a tile that regenerates more than a chunk (one very long sequence) is executed in order
//     straight in HBM by one wavefront.

#ifdef MZD_EXEC_STATS
__device__ unsigned long long g_exec_stats[32];
#define EXEC_STAT(i, n) do { const unsigned long long n_ = (unsigned long long)(n); if (lane == 0) atomicAdd(&g_exec_stats[i], n_); } while (0)
#else
#define EXEC_STAT(i, n) do { } while (0)
#endif
struct ExecShared {
    int error;
    uint32_t next_tile;  // first tile of the next chunk (written by thread 0)
    uint32_t chunk_end;  // block-relative output position where the current chunk ends
    uint32_t pad;
};

__device__ __forceinline__ int sel3(uint32_t k, int a, int b, int c) { return k == 0 ? a : (k == 1 ? b : c); }
__device__ __forceinline__ int resolve_hist(int v, int H0, int H1, int H2)
{
    if (v > 0) return v;
    uint32_t u = (uint32_t)(-v - 1);
    return sel3(u & 3, H0, H1, H2) - (int)(u >> 2);
}

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_shr(uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
// wave64 inclusive scan on the DPP path (row_shr 1/2/4/8, row_bcast 15/31): ~100 cycles
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
    v += dpp_shr<0x111, 0xf, 0xf>(v);
    v += dpp_shr<0x112, 0xf, 0xf>(v);
    v += dpp_shr<0x114, 0xf, 0xe>(v);
    v += dpp_shr<0x118, 0xf, 0xc>(v);
    v += dpp_shr<0x142, 0xa, 0xf>(v);
    v += dpp_shr<0x143, 0xc, 0xf>(v);
    return v;
}

// bits [bit, bit+n) of a 64-bit window, n <= 32
__device__ __forceinline__ uint64_t span_mask(uint32_t bit, uint32_t n)
{
    return ((n >= 32 ? 0xFFFFFFFFull : ((1ull << n) - 1))) << bit;
}
__device__ __forceinline__ void publish(uint32_t *vmap, uint32_t pos, uint32_t n)  // n <= 32
{
    // data bytes were stored by this wavefront BEFORE this point; DS operations of a wavefront execute
    // in order, so only the compiler has to be kept from sinking those stores below the OR
    asm volatile("" ::: "memory");
    const uint64_t m = span_mask(pos & 31, n);
    const uint32_t w = pos >> 5;
    atomicOr(&vmap[w], (uint32_t)m);
    if ((uint32_t)(m >> 32)) atomicOr(&vmap[w + 1], (uint32_t)(m >> 32));
}
__device__ __forceinline__ uint32_t ld32u_g(const uint8_t *p) { return ((const U32U *)p)->v; }
__device__ __forceinline__ void st32u_l(uint8_t *p, uint32_t v) { ((U32U *)p)->v = v; }
// two consecutive (4-byte aligned) LDS dwords with one instruction; `addr` = LDS byte address
__device__ __forceinline__ uint64_t lds_read2_u32(uint32_t addr)
{
    uint64_t v;
    asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// stores n (0..16) bytes held in w0..w3 (+ wt = bytes [n-4, n) when n >= 4) to LDS at d.
// k_exec is bound by the CU's one scalar unit, and what it executes is mostly the exec-mask
// bookkeeping of conditional stores (s_and_saveexec / s_or / branch per `if`).  So there are TWO
// size classes instead of a condition per dword: for n >= 4 all four dword stores are issued, the
// ones past the end collapsing onto the tail dword (position min(4k, n-4), data selected between
// word k and the tail word); for n < 4 three byte stores at positions 0, n/2, n-1.
__device__ __forceinline__ void lds_store_upto16(uint8_t *d, uint32_t n, uint32_t w0, uint32_t w1, uint32_t w2,
                                                 uint32_t w3, uint32_t wt)
{
    if (n >= 4) {
        // a byte-misaligned LDS dword store costs the LDS pipe one cycle per active lane (tools/ubench), and
        // the pipe is what k_exec fills most (SQ_LDS_IDX_ACTIVE): n <= 8 -- the common case -- stops at two
        const uint32_t last = n - 4;
        st32u_l(d, w0);
        st32u_l(d + min(4u, last), last >= 4 ? w1 : wt);
        if (n > 8) {
            st32u_l(d + min(8u, last), last >= 8 ? w2 : wt);
            st32u_l(d + min(12u, last), last >= 12 ? w3 : wt);
        }
    } else if (n) {
        const uint32_t h = n >> 1, e = n - 1;
        d[0] = (uint8_t)w0;
        d[h] = (uint8_t)(w0 >> (8 * h));
        d[e] = (uint8_t)(w0 >> (8 * e));
    }
}

// In-order execution of one tile straight in HBM by one wavefront (tiles that regenerate more
// than a chunk).  Every copy is wavefront-cooperative; writes are made visible before the next
// copy reads them (same CU: s_waitcnt is enough at workgroup scope).
__device__ void exec_tile_in_hbm(uint8_t *out, uint64_t outPos, const uint8_t *lits, bool litRle, uint32_t rleWord,
                                 uint32_t LL, uint32_t ML, int off, uint32_t dstL, uint32_t dstM, uint32_t srcL,
                                 bool valid, int lane)
{
    for (int sIdx = 0; sIdx < 64; sIdx++) {
        const uint32_t v = (uint32_t)__shfl((int)valid, sIdx, 64);
        if (!v) break;
        const uint32_t ll = (uint32_t)__shfl((int)LL, sIdx, 64), ml = (uint32_t)__shfl((int)ML, sIdx, 64);
        const uint32_t dl = (uint32_t)__shfl((int)dstL, sIdx, 64), dm = (uint32_t)__shfl((int)dstM, sIdx, 64);
        const uint32_t sl = (uint32_t)__shfl((int)srcL, sIdx, 64);
        const uint32_t o = (uint32_t)__shfl(off, sIdx, 64);
        uint8_t *d = out + outPos + dl;
        if (litRle) for (uint32_t j = lane; j < ll; j += 64) d[j] = (uint8_t)rleWord;
        else for (uint32_t j = lane; j < ll; j += 64) d[j] = lits[sl + j];
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): literal bytes are in memory before a match may read them
        if (ml == 0 || o == 0) continue;
        uint8_t *dmP = out + outPos + dm;
        const uint8_t *sp = dmP - o;
        if (o >= 64) {
            for (uint32_t base = 0; base < ml; base += 64) {  // each 64-byte step reads only finished bytes
                const uint32_t j = base + lane;
                uint8_t b = 0;
                if (j < ml) b = sp[j];
                if (j < ml) dmP[j] = b;
                __builtin_amdgcn_s_waitcnt(0);
            }
        } else {
            uint32_t r = (uint32_t)lane % o;  // periodic fill from the final pattern [sp, sp+o)
            const uint32_t stepr = 64 % o;
            for (uint32_t j = lane; j < ml; j += 64) {
                dmP[j] = sp[r];
                r += stepr;
                if (r >= o) r -= o;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
}

#ifndef MZD_EXEC_WAVES_PER_SIMD
#define MZD_EXEC_WAVES_PER_SIMD 8
#endif
#ifndef MZD_EXEC_IDLE_SLEEP
#define MZD_EXEC_IDLE_SLEEP 1  // units of 64 cycles between two polls of a wavefront that found nothing to do
#endif
#ifndef MZD_EXEC_MAX_THREADS
#define MZD_EXEC_MAX_THREADS 256  // experiment builds: up to 1024 (sixteen wavefronts on ONE frame, the whole block in LDS) with MZD_EXEC_WAVES_PER_SIMD=4
#endif
__global__ __launch_bounds__(MZD_EXEC_MAX_THREADS, MZD_EXEC_WAVES_PER_SIMD) void k_exec(const uint8_t *__restrict__ in, uint8_t *out_blob,
                                               const DFrame *__restrict__ frames, const DBlock *__restrict__ blocks,
                                               const BlockSum *__restrict__ sums, const uint64_t *__restrict__ recs,
                                               const TileBase *__restrict__ tiles, const uint8_t *__restrict__ litbuf,
                                               int32_t *frame_status, uint64_t *frame_out_len, uint32_t cap,
                                               const uint32_t *__restrict__ order, uint32_t first)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef MZD_SHIFT_EXEC  /* experiment: the whole instruction stream four bytes later */
    asm volatile("s_nop 0");
#endif
    // this workgroup's frame: in the batch's execution order when it has one (heterogeneous batches: the largest first)
    const uint32_t fidx = order ? order[first + blockIdx.x] : first + blockIdx.x;
    uint8_t *buf = smem;                                        // cap + 32 bytes
    uint32_t *vmap = (uint32_t *)(smem + cap + 32);             // cap / 32 + 4 words
    ExecShared *sh = (ExecShared *)(smem + cap + 32 + (cap / 32 + 4) * 4);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const DFrame fr = frames[fidx];
    uint8_t *out = out_blob + fr.out_offset;

    if (tid == 0) sh->error = fr.plan_status;
    __syncthreads();
    uint64_t outPos = 0;              // bytes of this frame produced so far
    int H0 = 1, H1 = 4, H2 = 8;       // framedecompressor.go:48,59

    for (uint32_t bi = 0; bi < fr.n_blocks && sh->error == MZD_OK; bi++) {
        const DBlock b = blocks[fr.first_block + bi];
        if (b.type != MZD_BLOCK_COMPRESSED) {
            // Raw (framedecompressor.go:211-215) / RLE (:229-241): straight copy / fill
            if (outPos + b.size > fr.out_capacity) {
                if (tid == 0) sh->error = MZD_ERR_DST_FULL;
                __syncthreads();
                break;
            }
            uint8_t *dst = out + outPos;
            const uint32_t n16 = b.size >> 4;
            if (b.type == MZD_BLOCK_RAW) {
                const uint8_t *src = in + b.src_off;
                for (uint32_t i = tid; i < n16; i += nthr) *(U128U *)(dst + 16 * i) = *(const U128U *)(src + 16 * i);
                for (uint32_t i = (n16 << 4) + tid; i < b.size; i += nthr) dst[i] = src[i];
            } else {
                const uint32_t v = in[b.src_off] * 0x01010101u;
                const U128U f{v, v, v, v};
                for (uint32_t i = tid; i < n16; i += nthr) *(U128U *)(dst + 16 * i) = f;
                for (uint32_t i = (n16 << 4) + tid; i < b.size; i += nthr) dst[i] = (uint8_t)v;
            }
            outPos += b.size;
            __syncthreads();  // later blocks may read these bytes as far matches
            continue;
        }

        const BlockSum bsum = sums[fr.first_block + bi];
        const uint32_t litTotal = bsum.lit_total, seqOut = bsum.out_total;
        int err = bsum.huf_err != 0xFFFFFFFFu ? (int)(bsum.huf_err & 0xFF) : bsum.status;
        if (err == MZD_OK && b.n_seq == 0) err = b.pad[1];  // (zero sequences in the two-byte form: the planner's verdict, sequences.go:126-208)
        if (err == MZD_OK && litTotal > b.lit_regen) err = MZD_ERR_LITERALS;  // sequence_execution.go:27-29
        const uint32_t blockOut = seqOut + (b.lit_regen - min(litTotal, b.lit_regen));
        if (err == MZD_OK && blockOut > kBlockMax) err = MZD_ERR_CORRUPT_SIZES;
        if (err == MZD_OK && outPos + blockOut > fr.out_capacity) err = MZD_ERR_DST_FULL;
        if (err != MZD_OK) {
            if (tid == 0) sh->error = err;
            __syncthreads();
            break;
        }
        const uint8_t *lits = (b.lit_type == MZD_LIT_HUF ? litbuf : in) + b.lit_src;
        const bool litRle = b.lit_type == MZD_LIT_RLE;
        const uint32_t rleWord = litRle ? lits[0] * 0x01010101u : 0;
        const uint32_t ntiles = (b.n_seq + 63) >> 6;
        const uint64_t *brec = recs + b.rec_off;
        const TileBase *btile = tiles + b.tile_off;
        uint8_t *bout = out + outPos;  // HBM address of block-relative position 0

        uint32_t t0 = 0;  // first tile of the current chunk
        while (t0 < ntiles && sh->error == MZD_OK) {
            // ---- chunk = maximal run of tiles [t0, t1) regenerating at most `cap` bytes
            const uint32_t chunkStart = btile[t0].out_pos;
            if (tid == 0) {
                uint32_t lo = t0 + 1, hi = ntiles;  // largest t1 with out(t1) - chunkStart <= cap (out(ntiles) = seqOut)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    const uint32_t e = mid == ntiles ? seqOut : btile[mid].out_pos;
                    if (e - chunkStart <= cap) lo = mid;
                    else hi = mid - 1;
                }
                const uint32_t e = lo == ntiles ? seqOut : btile[lo].out_pos;
                sh->next_tile = lo;
                sh->chunk_end = e;
            }
            __syncthreads();
            const uint32_t t1 = sh->next_tile;
            const uint32_t chunkEnd = sh->chunk_end;
            const uint32_t chunkLen = chunkEnd - chunkStart;
            if (chunkLen > cap) {
                // ---- oversized tile: in-order execution in HBM by wavefront 0
                if (wave == 0) {
                    const uint32_t si = t0 * 64 + lane;
                    const bool valid = si < b.n_seq;
                    const uint64_t rec = valid ? brec[si] : 0ull;
                    const uint32_t LL = (uint32_t)rec & kRecLlMask;
                    const uint32_t ML = (uint32_t)(rec >> kRecMlShift) & kRecMlMask;
                    const uint32_t offf = (uint32_t)(rec >> kRecOffShift) & kRecOffMask;
                    int off = (int)offf;
                    if (offf & kRecOffSymbolic) {
                        uint32_t u = offf & (kRecOffSymbolic - 1);
                        off = sel3(u & 3, H0, H1, H2) - (int)(u >> 2);
                    }
                    const TileBase tb = btile[t0];
                    const uint32_t litEnd = tb.lit_pos + wave_incl_scan_dpp(LL);
                    const uint32_t outEnd = tb.out_pos + wave_incl_scan_dpp(LL + ML);
                    const uint32_t dstM = outEnd - ML, dstL = dstM - LL, srcL = litEnd - LL;
                    const bool bad = valid && ML > 0 && (off <= 0 || (uint64_t)off > outPos + dstM);
                    if (wave_any(bad)) {
                        if (lane == 0) atomicMax(&sh->error, MZD_ERR_OFFSET);
                    } else {
                        exec_tile_in_hbm(out, outPos, lits, litRle, rleWord, LL, ML, off, dstL, dstM, srcL, valid, lane);
                    }
                }
                __syncthreads();
                t0 = t1;
                continue;
            }
            const uint32_t mis = (uint32_t)((uintptr_t)(bout + chunkStart) & 15);
            uint8_t *lbuf = buf + mis;  // lbuf[q] = chunk-relative output byte q
            for (uint32_t i = tid; i < ((chunkLen + 31) >> 5) + 1; i += nthr) vmap[i] = 0;
            __syncthreads();

            // software pipeline: records / tile bases of the NEXT tile are loaded while the current one runs
            uint32_t tile = t0 + wave;
            uint64_t rec_n = 0;
            TileBase tb_n{0, 0};
            if (tile < t1) {
                const uint32_t si = tile * 64 + lane;
                rec_n = si < b.n_seq ? brec[si] : 0ull;
                tb_n = btile[tile];
            }
            for (; tile < t1; tile += nwaves) {
                const uint64_t rec = rec_n;
                const TileBase tb = tb_n;
                {
                    const uint32_t nt = tile + nwaves;
                    if (nt < t1) {
                        const uint32_t si = nt * 64 + lane;
                        rec_n = si < b.n_seq ? brec[si] : 0ull;
                        tb_n = btile[nt];
                    }
                }
                const bool valid = tile * 64 + lane < b.n_seq;
                EXEC_STAT(0, 1);
                const uint32_t LL = (uint32_t)rec & kRecLlMask;
                const uint32_t ML = (uint32_t)(rec >> kRecMlShift) & kRecMlMask;
                const uint32_t offf = (uint32_t)(rec >> kRecOffShift) & kRecOffMask;
                int off = (int)offf;
                if (offf & kRecOffSymbolic) {
                    uint32_t u = offf & (kRecOffSymbolic - 1);
                    off = sel3(u & 3, H0, H1, H2) - (int)(u >> 2);
                }
                const uint32_t litEnd = tb.lit_pos + wave_incl_scan_dpp(LL);
                const uint32_t outEnd = tb.out_pos + wave_incl_scan_dpp(LL + ML);
                const uint32_t dstMb = outEnd - ML, srcL = litEnd - LL;  // block-relative
                const bool bad = valid && ML > 0 && (off <= 0 || (uint64_t)off > outPos + dstMb);  // ringbuffer.go:206-214
                if (wave_any(bad)) {
                    if (lane == 0) atomicMax(&sh->error, MZD_ERR_OFFSET);
                }
                // chunk-relative positions
                const uint32_t dstM = dstMb - chunkStart, dstL = dstM - LL;
                const int srcM = (int)dstM - off;  // < 0: before the chunk (final, in HBM)

                // ---- literals (sequence_execution.go:19-34): they depend on nothing
                {
                    const uint32_t sLL = (valid && LL <= 32) ? LL : 0;  // up to two 16-byte loads per lane
                    if (wave_any(sLL != 0)) {
                        U128U a{rleWord, rleWord, rleWord, rleWord}, c{rleWord, rleWord, rleWord, rleWord};
                        uint32_t wt = rleWord;
                        const bool two = wave_any(sLL > 16);  // wave-uniform: a second 16-byte half exists somewhere
                        if (sLL) {
                            if (!litRle) {
#ifndef MZD_ABL_EXEC_NOLIT  /* ablations: timing experiments only, wrong results */
                                a = *(const U128U *)(lits + srcL);
                                if (two) c = *(const U128U *)(lits + srcL + (sLL > 16 ? 16 : 0));
#endif
#if !defined(MZD_ABL_EXEC_NOLIT) && !defined(MZD_ABL_EXEC_NOWT)
                                wt = ld32u_g(lits + srcL + (sLL >= 4 ? sLL - 4 : 0));
#endif
                            }
                            uint8_t *d = lbuf + dstL;
                            lds_store_upto16(d, min(sLL, 16u), a.x, a.y, a.z, a.w, wt);
                            if (two) lds_store_upto16(d + 16, sLL > 16 ? sLL - 16 : 0, c.x, c.y, c.z, c.w, wt);
                            publish(vmap, dstL, sLL);
                        }
                    }
                    uint64_t longs = wave_ballot(valid && LL > 32);
                    EXEC_STAT(10, __popcll(longs));
                    EXEC_STAT(11, __popcll(__ballot(sLL != 0)));
                    while (longs) {
                        const int src = __builtin_ctzll(longs);
                        longs &= longs - 1;
                        const uint32_t n = (uint32_t)__shfl((int)LL, src, 64);
                        const uint32_t d = (uint32_t)__shfl((int)dstL, src, 64);
                        const uint32_t s = (uint32_t)__shfl((int)srcL, src, 64);
                        if (litRle) for (uint32_t j = lane; j < n; j += 64) lbuf[d + j] = (uint8_t)rleWord;
                        else for (uint32_t j = lane; j < n; j += 64) lbuf[d + j] = lits[s + j];
                        for (uint32_t j = lane * 32; j < n; j += 64 * 32) publish(vmap, d + j, min(32u, n - j));
                    }
                }

                // ---- matches (sequence_execution.go:43-49, ringbuffer.go:242-277)
                bool pending = valid && ML > 0 && !bad;
                const bool overlap = (uint32_t)off < ML;
                const uint32_t span = overlap ? (uint32_t)max(off, 1) : ML;  // bytes that are true sources
                // (a) short matches sourced entirely before the chunk: final bytes in HBM, no waiting
                {
                    const bool g = pending && ML <= 32 && !overlap && srcM + (int)ML <= 0;
                    EXEC_STAT(1, __popcll(__ballot(pending)));
                    EXEC_STAT(2, __popcll(__ballot(g)));
                    if (wave_any(g)) {
                        const uint8_t *sp = bout + (int)chunkStart + srcM;  // may point into earlier blocks
                        U128U a{0, 0, 0, 0}, c{0, 0, 0, 0};
                        uint32_t wt = 0;
                        const bool two = wave_any(g && ML > 16);
                        if (g) {
#ifndef MZD_ABL_EXEC_NOFAR
                            a = *(const U128U *)sp;
                            if (two) c = *(const U128U *)(sp + (ML > 16 ? 16 : 0));
#endif
#if !defined(MZD_ABL_EXEC_NOFAR) && !defined(MZD_ABL_EXEC_NOWT)
                            wt = ld32u_g(sp + (ML >= 4 ? ML - 4 : 0));
#endif
                            uint8_t *d = lbuf + dstM;
                            lds_store_upto16(d, min(ML, 16u), a.x, a.y, a.z, a.w, wt);
                            if (two) lds_store_upto16(d + 16, ML > 16 ? ML - 16 : 0, c.x, c.y, c.z, c.w, wt);
                            publish(vmap, dstM, ML);
                        }
                        pending = pending && !g;
                    }
                }
                // (b) everything else: dataflow on the validity bitmap
                const bool isShort = ML <= 32;
                // readiness mask of a short match: source bytes that lie inside the chunk
                const int s0 = max(srcM, 0), s1 = srcM + (int)span;
                const uint64_t needm = (isShort && s1 > s0) ? span_mask((uint32_t)s0 & 31, (uint32_t)(s1 - s0)) : 0ull;
                const uint32_t needw = (uint32_t)s0 >> 5;
                const bool fastKind = isShort && !overlap && srcM >= 0;
#if defined(MZD_EXEC_CXX_LOOP) || defined(MZD_EXEC_STATS)
                uint32_t spins = 0;
                while (__any(pending)) {
                    EXEC_STAT(3, 1);
                    bool ready = false;
                    if (pending && isShort) {
                        // both words in ONE LDS instruction (ds_read2_b32: an aligned LDS instruction costs the pipe ~4.3
                        // cycles whatever the lanes); bits are only ever set, a stale word just delays the lane one pass
                        const uint64_t v = lds_read2_u32((uint32_t)(uintptr_t)(vmap + needw));
                        ready = (v & needm) == needm;
                    }
                    asm volatile("" ::: "memory");  // data reads below stay below the validity reads
                    bool progressed = false;
                    // (b1) short, non-overlapping, source inside the chunk: aligned dword reads + funnel
                    const bool fast = ready && fastKind;
                    if (__any(fast)) {
                        EXEC_STAT(4, 1);
                        EXEC_STAT(5, __popcll(__ballot(fast)));
                        progressed = true;
                        // wave-uniform bound on the dword loop from two ballots (a shuffle reduction costs ~450 cycles)
                        const uint32_t mlc = __any(fast && ML > 16) ? 32u : (__any(fast && ML > 8) ? 16u : 8u);
                        if (fast) {
                            // one exec region for the whole copy; inside, the two size classes of lds_store_upto16.
                            // The source is read with byte-misaligned 8-byte LDS reads: they cost the LDS pipe a cycle
                            // per ACTIVE lane, and a pass has ~7 -- cheaper than three aligned dword reads plus the
                            // funnel shifts per 8 bytes (an aligned LDS instruction costs ~4.3 cycles whatever the lanes).
                            const uint8_t *sp = lbuf + srcM;
                            uint8_t *d = lbuf + dstM;
                            if (ML >= 4) {
                                const uint32_t last = ML - 4;
                                const uint32_t xt = ((const U32U *)(sp + last))->v;  // source bytes [ML-4, ML)
                                for (uint32_t j = 0; 4 * j < mlc; j += 2) {
                                    const uint64_t x2 = ((const U64U *)(sp + 4 * j))->v;  // past the source: unused (and inside the buffer's slack)
                                    // lanes whose copy is complete drop out pairwise (LDS time is per active lane)
                                    if (j < 2 || 4 * j < ML) {
                                        st32u_l(d + min(4 * j, last), 4 * j <= last ? (uint32_t)x2 : xt);
                                        st32u_l(d + min(4 * j + 4, last), 4 * j + 4 <= last ? (uint32_t)(x2 >> 32) : xt);
                                    }
                                }
                            } else {
                                const uint32_t first = ((const U32U *)sp)->v;
                                const uint32_t h = ML >> 1, e = ML - 1;
                                d[0] = (uint8_t)first;
                                d[h] = (uint8_t)(first >> (8 * h));
                                d[e] = (uint8_t)(first >> (8 * e));
                            }
                            publish(vmap, dstM, ML);
                        }
                    }
                    // (b2) short matches that overlap themselves or straddle the chunk start: byte loop
                    const bool slowb = ready && !fastKind;
                    if (__any(slowb)) {
                        EXEC_STAT(6, 1);
                        EXEC_STAT(7, __popcll(__ballot(slowb)));
                        progressed = true;
                        const uint32_t n = slowb ? ML : 0;
                        const uint32_t nmax = wave_max_u32(n);
                        const uint8_t *gsrc = bout + (int)chunkStart;  // HBM address of chunk-relative position 0
                        for (uint32_t j = 0; j < nmax; j++) {
                            if (j < n) {
                                const int q = srcM + (int)j;
                                const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                lbuf[dstM + j] = v;
                            }
                        }
                        if (slowb) publish(vmap, dstM, ML);
                    }
                    pending = pending && !ready;
                    // (b3) at most one long match per iteration, whole wavefront, non-blocking readiness test
                    const uint64_t longs = __ballot(pending && !isShort);
                    if (longs) {
                        EXEC_STAT(8, 1);
                        const int src = __builtin_ctzll(longs);
                        const uint32_t n = (uint32_t)__shfl((int)ML, src, 64);
                        const uint32_t d = (uint32_t)__shfl((int)dstM, src, 64);
                        const int s = __shfl(srcM, src, 64);
                        const uint32_t o = (uint32_t)__shfl(off, src, 64);
                        const uint32_t sp2 = (uint32_t)__shfl((int)span, src, 64);
                        bool ok = true;
                        const int q0 = max(s, 0), q1 = s + (int)sp2;
                        if (q1 > q0) {
                            const uint32_t wf = (uint32_t)q0 >> 5, wl = (uint32_t)(q1 - 1) >> 5;
                            for (uint32_t wi = wf + lane; wi <= wl; wi += 64) {
                                uint32_t need = 0xFFFFFFFFu;
                                if (wi == wf) need &= 0xFFFFFFFFu << ((uint32_t)q0 & 31);
                                if (wi == wl) need &= 0xFFFFFFFFu >> (31 - ((uint32_t)(q1 - 1) & 31));
                                const uint32_t v = __hip_atomic_load(&vmap[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                ok = ok && ((v & need) == need);
                            }
                        }
                        asm volatile("" ::: "memory");
                        if (__all(ok)) {
                            progressed = true;
                            const uint8_t *gsrc = bout + (int)chunkStart;
                            if (o >= 64) {
                                // each 64-byte step only reads bytes written by earlier steps (in-order LDS)
                                for (uint32_t j = lane; j < n; j += 64) {
                                    const int q = s + (int)j;
                                    const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                    lbuf[d + j] = v;
                                }
                            } else {
                                // overlapping: periodic fill from the (final) pattern [s, s+o)
                                uint32_t r = (uint32_t)lane % o;
                                const uint32_t stepr = 64 % o;
                                for (uint32_t j = lane; j < n; j += 64) {
                                    const int q = s + (int)r;
                                    const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                    lbuf[d + j] = v;
                                    r += stepr;
                                    if (r >= o) r -= o;
                                }
                            }
                            for (uint32_t j = lane * 32; j < n; j += 64 * 32) publish(vmap, d + j, min(32u, n - j));
                            if (lane == src) pending = false;
                        }
                    }
                    if (!progressed) {
                        EXEC_STAT(9, 1);
                        if ((++spins & 15) == 0 &&
                            __hip_atomic_load(&sh->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != MZD_OK)
                            break;  // corrupt input: a skipped match would never validate its bytes
                        __builtin_amdgcn_s_sleep(MZD_EXEC_IDLE_SLEEP);
                    }
                }
#else
                // The same loop with its common iteration -- readiness test of the short matches, the copy of the ready
                // non-overlapping ones, publication -- as ONE hand-written statement.  k_exec is bound by the CU's scalar
                // unit, and what the compiler's version (above, kept for the statistics build and as the reference) spends
                // there is exec-mask bookkeeping: every wave-uniform `if (__any(..))` is a v_cndmask / v_cmp / s_cbranch
                // triple, every divergent `if` an s_and_saveexec / s_or pair (58 scalar + branch instructions per fast
                // iteration).  Here the pending lanes are MASKS in scalar registers (F: short, non-overlapping, sourced
                // inside the chunk; S: the other short ones; L: long ones), an iteration narrows exec step by step and
                // restores it once (~20).  The rare kinds (S: 0.08 passes per tile on the bench workload, L: 0.0002) stay
                // in C++.  Same stores in the same order as the C++ fast pass; DS operations of a wavefront execute in
                // order, so the bytes are in LDS before their validity bits.
                const bool fastK = fastKind && ML >= 3;  // (a match is >= 3 bytes by the format; the hand-written copy relies on it)
                uint64_t F = wave_ballot(pending && fastK);
                uint64_t S = wave_ballot(pending && isShort && !fastK);
                uint64_t L = wave_ballot(pending && !isShort);
                const uint32_t na = (uint32_t)(uintptr_t)(vmap + needw);
                const uint32_t nlo = (uint32_t)needm, nhi = (uint32_t)(needm >> 32);
                const uint32_t srcA = (uint32_t)(uintptr_t)lbuf + (uint32_t)srcM, dstA = (uint32_t)(uintptr_t)lbuf + dstM;
                const uint64_t pm = span_mask(dstM & 31, ML);
                const uint32_t pa = (uint32_t)(uintptr_t)(vmap + (dstM >> 5)), plo = (uint32_t)pm, phi = (uint32_t)(pm >> 32);
                uint32_t spins = 0;
                while (F | S | L) {
                    uint64_t RF, RS, T;
#define MZD_EXEC_BLOCK(O0, O4)                                                                                          \
    "ds_read_b64 v[56:57], %[src] offset:" #O0 "\n\t"   /* past the source: unused (and inside the buffer's slack) */    \
    "v_cmp_le_u32 vcc, " #O0 ", v58\n\t"                                                                                \
    "v_min_u32 v60, " #O0 ", v58\n\t"                                                                                   \
    "v_add_u32 v60, %[dst], v60\n\t"                                                                                    \
    "v_min_u32 v62, " #O4 ", v58\n\t"                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "v_cndmask_b32 v61, v59, v56, vcc\n\t"                                                                              \
    "v_cmp_le_u32 vcc, " #O4 ", v58\n\t"                                                                                \
    "v_add_u32 v62, %[dst], v62\n\t"                                                                                    \
    "ds_write_b32 v60, v61\n\t"                                                                                         \
    "v_cndmask_b32 v63, v59, v57, vcc\n\t"                                                                              \
    "ds_write_b32 v62, v63\n\t"
                    asm volatile(
                        "s_or_b64 exec, %[F], %[S]\n\t"            // the short matches still pending
                        "ds_read2_b32 v[56:57], %[na] offset1:1\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_bfi_b32 v56, v56, 0, %[nlo]\n\t"        // needed and not valid
                        "v_bfi_b32 v57, v57, 0, %[nhi]\n\t"
                        "v_or_b32 v56, v56, v57\n\t"
                        "v_cmp_eq_u32 vcc, 0, v56\n\t"             // ready
                        "s_and_b64 %[RS], vcc, %[S]\n\t"
                        "s_and_b64 %[RF], vcc, %[F]\n\t"
                        "s_cbranch_scc0 L_ex_done_%=\n\t"
                        "s_andn2_b64 %[F], %[F], %[RF]\n\t"
                        "s_mov_b64 exec, %[RF]\n\t"
                        "v_cmp_gt_u32 vcc, 4, %[ml]\n\t"
                        "v_add_u32 v58, -4, %[ml]\n\t"             // last = ML - 4
                        "s_and_saveexec_b64 %[T], vcc\n\t"         // T = the ready lanes
                        "s_cbranch_execz L_ex_no3_%=\n\t"
                        // three bytes
                        "ds_read_b32 v56, %[src]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b16 %[dst], v56\n\t"
                        "ds_write_b8_d16_hi %[dst], v56 offset:2\n"
                        "L_ex_no3_%=:\n\t"
                        "s_andn2_b64 exec, %[T], vcc\n\t"          // four bytes and more
                        "s_cbranch_execz L_ex_pub_%=\n\t"
                        // 4..8 bytes: two dword stores, the second at min(4, ML - 4) with the source bytes from there on
                        // (one 8-byte read, byte-misaligned: a cycle per active lane in the LDS pipe; no separate tail read)
                        "ds_read_b64 v[56:57], %[src]\n\t"
                        "v_min_u32 v60, 4, v58\n\t"
                        "v_lshlrev_b32 v61, 3, v60\n\t"
                        "v_add_u32 v60, %[dst], v60\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "ds_write_b32 %[dst], v56\n\t"
                        "v_lshrrev_b64 v[62:63], v61, v[56:57]\n\t"
                        "ds_write_b32 v60, v62\n\t"
                        "v_cmp_lt_u32 vcc, 8, %[ml]\n\t"           // lanes whose copy is complete drop out (LDS time is per active lane)
                        "s_and_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execz L_ex_pub_%=\n\t"
                        // longer: dword stores at min(4k, ML - 4), the ones past the end collapsing onto the tail dword
                        "v_add_u32 v59, %[src], v58\n\t"
                        "ds_read_b32 v59, v59\n\t"                 // source bytes [ML - 4, ML)
                        MZD_EXEC_BLOCK(8, 12)
                        "v_cmp_lt_u32 vcc, 16, %[ml]\n\t"
                        "s_and_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execz L_ex_pub_%=\n\t"
                        MZD_EXEC_BLOCK(16, 20)
                        "v_cmp_lt_u32 vcc, 24, %[ml]\n\t"
                        "s_and_b64 exec, exec, vcc\n\t"
                        "s_cbranch_execz L_ex_pub_%=\n\t"
                        MZD_EXEC_BLOCK(24, 28)
                        "L_ex_pub_%=:\n\t"
                        // (one 64-bit atomic on the aligned pair of bitmap words + a rare third word: k_exec 11.0 -> 11.15 ms)
                        "s_mov_b64 exec, %[T]\n\t"
                        "ds_or_b32 %[pa], %[plo]\n\t"
                        "v_cmp_ne_u32 vcc, 0, %[phi]\n\t"
                        "s_and_b64 exec, exec, vcc\n\t"
                        "ds_or_b32 %[pa], %[phi] offset:4\n"
                        "L_ex_done_%=:\n\t"
                        "s_mov_b64 exec, -1\n\t"
                        : [F] "+s"(F), [RF] "=&s"(RF), [RS] "=&s"(RS), [T] "=&s"(T)
                        : [S] "s"(S), [na] "v"(na), [nlo] "v"(nlo), [nhi] "v"(nhi), [src] "v"(srcA), [dst] "v"(dstA), [ml] "v"(ML),
                          [pa] "v"(pa), [plo] "v"(plo), [phi] "v"(phi)
                        : "memory", "vcc", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
#undef MZD_EXEC_BLOCK
                    bool progressed = RF != 0;
                    // short matches that overlap themselves or straddle the chunk start: byte loop
                    if (RS) {
                        progressed = true;
                        const bool slowb = (RS >> lane) & 1;
                        const uint32_t n = slowb ? ML : 0;
                        const uint32_t nmax = wave_max_u32(n);
                        const uint8_t *gsrc = bout + (int)chunkStart;  // HBM address of chunk-relative position 0
                        for (uint32_t j = 0; j < nmax; j++) {
                            if (j < n) {
                                const int q = srcM + (int)j;
                                const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                lbuf[dstM + j] = v;
                            }
                        }
                        if (slowb) publish(vmap, dstM, ML);
                        S &= ~RS;
                    }
                    // at most one long match per iteration, whole wavefront, non-blocking readiness test
                    if (L) {
                        const int src = __builtin_ctzll(L);
                        const uint32_t n = (uint32_t)__shfl((int)ML, src, 64);
                        const uint32_t d = (uint32_t)__shfl((int)dstM, src, 64);
                        const int s = __shfl(srcM, src, 64);
                        const uint32_t o = (uint32_t)__shfl(off, src, 64);
                        const uint32_t sp2 = (uint32_t)__shfl((int)span, src, 64);
                        bool ok = true;
                        const int q0 = max(s, 0), q1 = s + (int)sp2;
                        if (q1 > q0) {
                            const uint32_t wf = (uint32_t)q0 >> 5, wl = (uint32_t)(q1 - 1) >> 5;
                            for (uint32_t wi = wf + lane; wi <= wl; wi += 64) {
                                uint32_t need = 0xFFFFFFFFu;
                                if (wi == wf) need &= 0xFFFFFFFFu << ((uint32_t)q0 & 31);
                                if (wi == wl) need &= 0xFFFFFFFFu >> (31 - ((uint32_t)(q1 - 1) & 31));
                                const uint32_t v = __hip_atomic_load(&vmap[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                ok = ok && ((v & need) == need);
                            }
                        }
                        asm volatile("" ::: "memory");
                        if (__all(ok)) {
                            progressed = true;
                            const uint8_t *gsrc = bout + (int)chunkStart;
                            if (o >= 64) {
                                // each 64-byte step only reads bytes written by earlier steps (in-order LDS)
                                for (uint32_t j = lane; j < n; j += 64) {
                                    const int q = s + (int)j;
                                    const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                    lbuf[d + j] = v;
                                }
                            } else {
                                // overlapping: periodic fill from the (final) pattern [s, s+o)
                                uint32_t r = (uint32_t)lane % o;
                                const uint32_t stepr = 64 % o;
                                for (uint32_t j = lane; j < n; j += 64) {
                                    const int q = s + (int)r;
                                    const uint8_t v = q >= 0 ? lbuf[q] : gsrc[q];
                                    lbuf[d + j] = v;
                                    r += stepr;
                                    if (r >= o) r -= o;
                                }
                            }
                            for (uint32_t j = lane * 32; j < n; j += 64 * 32) publish(vmap, d + j, min(32u, n - j));
                            L &= L - 1;
                        }
                    }
                    if (!progressed) {
                        if ((++spins & 15) == 0 &&
                            __hip_atomic_load(&sh->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != MZD_OK)
                            break;  // corrupt input: a skipped match would never validate its bytes
                        __builtin_amdgcn_s_sleep(MZD_EXEC_IDLE_SLEEP);
                    }
                }
#endif
            }
            __syncthreads();
            // ---- the chunk leaves for HBM: head bytes, aligned 16-byte body, tail bytes
            {
                uint8_t *dst = bout + chunkStart;
                const uint32_t head = min(chunkLen, (16u - mis) & 15u);
                if ((uint32_t)tid < head) dst[tid] = lbuf[tid];
                const uint32_t body = (chunkLen - head) >> 4;
                const uint4 *lsrc = (const uint4 *)(lbuf + head);  // 16-byte aligned in LDS by construction
                uint4 *gdst = (uint4 *)(dst + head);
                for (uint32_t i = tid; i < body; i += nthr) gdst[i] = lsrc[i];
                for (uint32_t i = head + (body << 4) + tid; i < chunkLen; i += nthr) dst[i] = lbuf[i];
            }
            __syncthreads();  // flushed bytes are visible to the whole workgroup before the next chunk reads them
            t0 = t1;
        }
        // ---- literals after the last sequence (sequence_execution.go:55-59): straight to HBM (unless the Huffman
        // stage already put them there: a block without sequences whose place in the frame was known beforehand)
        if (!b.pad[0]) {
            const uint32_t rest = b.lit_regen - litTotal;
            uint8_t *d = bout + seqOut;
            if (litRle) {
                const uint32_t n16 = rest >> 4;
                const U128U f{rleWord, rleWord, rleWord, rleWord};
                for (uint32_t j = tid; j < n16; j += nthr) *(U128U *)(d + 16 * j) = f;
                for (uint32_t j = (n16 << 4) + tid; j < rest; j += nthr) d[j] = (uint8_t)rleWord;
            } else {
                const uint8_t *s = lits + litTotal;
                const uint32_t n16 = rest >> 4;
                for (uint32_t j = tid; j < n16; j += nthr) *(U128U *)(d + 16 * j) = *(const U128U *)(s + 16 * j);
                for (uint32_t j = (n16 << 4) + tid; j < rest; j += nthr) d[j] = s[j];
            }
        }
        {
            // offset history carried to the next block (framedecompressor.go:23; persists across blocks)
            const int n0 = resolve_hist(bsum.hist[0], H0, H1, H2);
            const int n1 = resolve_hist(bsum.hist[1], H0, H1, H2);
            const int n2 = resolve_hist(bsum.hist[2], H0, H1, H2);
            H0 = n0; H1 = n1; H2 = n2;
        }
        outPos += blockOut;
        __syncthreads();
    }
    if (tid == 0) {
        int e = sh->error;
        if (e == MZD_OK && fr.content_size != MZD_UNKNOWN_SIZE && outPos != fr.content_size) e = MZD_ERR_DST_FULL;
        frame_status[fidx] = e;
        frame_out_len[fidx] = outPos;
    }
}

}  // namespace mzd
