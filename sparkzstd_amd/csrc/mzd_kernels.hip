// mzd_kernels.hip -- the zstd block-decode hot path as hand-written HIP for gfx950
// (MI355X, CDNA4: 64-wide wavefronts, 160 KiB LDS per CU).  Integer / table work only:
// no MFMA.  Three stages, one kernel each (plus a trivial init):
//
//   k_huf   Huffman literal decode      replaces structure/huffman.go:221-264 DecodeStream
//                                       (+ the 1/4-stream dispatch literals.go:295-371)
//   k_seq   FSE sequence decode         replaces structure/sequences.go:126-206 DecodeSequences,
//                                       :64-123 DecodeSequence, fse/fse.go:253-290 state accessors,
//                                       and folds in sequence_execution.go:65-114 nextOffset
//   k_exec  sequence execution          replaces decompression/sequence_execution.go:14-63 and
//                                       ringbuffer.go:102-277 Push/Repeat/RepeatBeforeIndex, plus the
//                                       Raw / RLE block arms framedecompressor.go:211-215,229-241
//
// Mapping (see DESIGN.md for the reasoning and the roofline of each):
//   k_huf   one LANE per Huffman stream; the 4 streams of a literals section sit in 4 adjacent
//           lanes and share one decode table staged in LDS; 16 sections per wavefront.
//   k_seq   one LANE per block = one serial LL/ML/OF state chain; each chain's three FSE tables
//           live in LDS (that is what bounds the number of resident chains), the bitstream is
//           consumed through a 128-bit register window refilled ahead of use.
//   k_exec  one WORKGROUP per frame; the block being regenerated lives in a 128 KiB LDS buffer
//           (the "window" all near matches hit), 64-sequence tiles are executed by wavefronts
//           out of order with an in-order commit watermark, and the finished block leaves for HBM
//           in aligned 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mzd.h"
#include "mzd_device.h"

namespace mzd {

// ------------------------------------------------------------------------------------------
// small helpers

struct __attribute__((packed, aligned(1))) U64U { uint64_t v; };
struct __attribute__((packed, aligned(1))) U32U { uint32_t v; };
struct __attribute__((packed, aligned(1))) U128U { uint32_t x, y, z, w; };

__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { return ((const U64U *)p)->v; }

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= d) v += y;
    }
    return v;
}

// Backward bit reader (bitstream/reversebitstream.go semantics) held in registers.
//   C  = stream bytes [ptr, ptr+8) as a little-endian u64 (bit 63 = MSB of byte ptr+7)
//   k  = bits of C already consumed, counted from bit 63 downwards
//   D  = stream bytes [ptr-8, ptr), loaded one refill AHEAD of its use so that the memory
//        latency is off the serial decode chain.
// Bytes below the start of the stream read as zero (reversebitstream.go:23-27,67-75).
struct BackBits {
    const uint8_t *s;
    uint64_t C, D;
    int ptr, k;

    __device__ __forceinline__ uint64_t load_below(int at) const
    {
        // bytes [at, at+8) relative to s; zero for addresses below s
        int a = max(at, -8);
        uint64_t v = ld64u(s + a);
        if (at < 0) {
            int z = -at;
            v = z >= 8 ? 0ull : ((v >> (8 * z)) << (8 * z));
        }
        return v;
    }
    // returns number of real data bits R (after the padding marker), or -1 on bad padding
    __device__ __forceinline__ int init(const uint8_t *start, int len)
    {
        s = start;
        ptr = len - 8;
        C = load_below(ptr);
        D = load_below(ptr - 8);
        uint32_t last = (uint32_t)(C >> 56);
        if (last == 0) {  // huffman.go:235-237 / sequences.go:141-143: more than 8 padding bits
            k = 8;
            return -1;
        }
        k = __builtin_clz(last) - 24 + 1;  // zero bits above the marker + the marker itself
        return 8 * len - k;
    }
    // drop whole consumed bytes, pull the same number of bytes in from D, prefetch the next D
    __device__ __forceinline__ void refill()
    {
        int nb = k >> 3;
        int sh = nb * 8;
        C = (C << sh) | ((D >> 1) >> (63 - sh));
        ptr -= nb;
        k &= 7;
        D = load_below(ptr - 8);
    }
    // next n (0..32) unread bits, MSB first; requires k + n <= 64
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(((C << k) >> 1) >> (63 - n)); }
};

// ------------------------------------------------------------------------------------------
// k_init: reset the per-block summaries each run (blocks without sequences are never touched
// by k_seq: their offset-history transform is the identity).

__global__ void k_init(BlockSum *sums, uint32_t n_blocks)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    BlockSum b;
    b.lit_total = 0;
    b.out_total = 0;
    b.hist[0] = -1;  // symbolic "slot 0 at block start"
    b.hist[1] = -2;
    b.hist[2] = -3;
    b.status = MZD_OK;
    b.pad[0] = b.pad[1] = 0;
    sums[i] = b;
}

// ------------------------------------------------------------------------------------------
// k_huf: Huffman literal streams.  One wavefront per workgroup; lane = stream; 16 table slots.
//
// Restates huffman.go:221-264: after the padding marker the stream holds R data bits; each
// symbol is looked up with the next MaxBits unread bits (zero-extended below bit 0) and
// consumes NumberOfBits of them; the stream is valid iff exactly R bits are consumed when the
// expected number of symbols has been produced (:257-261 with literals.go:320,332,349,366).

constexpr int kHufQuads = 16;

__global__ __launch_bounds__(64) void k_huf(const uint8_t *__restrict__ in, const HufTask *__restrict__ tasks,
                                            uint32_t n_tasks, const uint16_t *__restrict__ huf_entries,
                                            uint8_t *__restrict__ litbuf, BlockSum *sums, uint32_t slot_cells)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *tbl_all = (uint16_t *)smem;
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x * 64 + lane;
    HufTask t;
    if (tid < n_tasks) t = tasks[tid];
    else { t.in_size = 0; t.out_size = 0; t.table_off = 0; t.max_bits = 0; t.in_off = 0; t.out_off = 0; t.block = 0; }

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
    if ((t.in_size | t.out_size) == 0) return;  // null task

    const uint16_t *tbl = tbl_all + (size_t)(lane >> 2) * slot_cells;
    const int mb = (int)t.max_bits;
    BackBits br;
    int rem = br.init(in + t.in_off, (int)t.in_size);
    int status = MZD_OK;
    if (rem < 0) status = MZD_ERR_BAD_PADDING;
    uint8_t *out = litbuf + t.out_off;
    uint32_t cnt = 0;
    const uint32_t want = t.out_size;

    if (status == MZD_OK) {
        // bulk: 16 symbols per iteration while at least 16*11 bits and 16 output slots remain
        while (cnt + 16 <= want && rem >= 16 * 11) {
            uint32_t w[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                br.refill();  // k < 8 afterwards; 4 symbols * 11 bits + 11-bit window <= 57
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
        }
        // tail: symbol by symbol
        while (cnt < want && rem > 0) {
            br.refill();
            uint32_t idx = (uint32_t)((br.C << br.k) >> (64 - mb));
            uint32_t e = tbl[idx];
            out[cnt++] = (uint8_t)(e & 0xFF);
            int nb = (int)(e >> 8);
            br.k += nb;
            rem -= nb;
        }
        if (rem != 0) status = MZD_ERR_HUF_BITS;         // huffman.go:257-261
        else if (cnt != want) status = MZD_ERR_HUF_LENGTH;  // literals.go:320,332,349,366
    }
    if (status != MZD_OK) atomicCAS(&sums[t.block].status, MZD_OK, status);
}

// ------------------------------------------------------------------------------------------
// k_seq: FSE sequence decode.  One wavefront per workgroup, lane = one block's chain.
//
// LDS cell formats (built from the host cells {baseline, nbits, symbol} while staging):
//   CELL16: next(10) | symbol(6)       nbits = acc_log - highbit(next), baseline = (next << nbits) - size
//           (fse.go:209-213 run backwards); 2 bytes -> 63 chains per CU
//   CELL32: baseline(10) | nbits(4) | extra_bits(5) | symbol(6); 4 bytes -> 31 chains per CU,
//           no second lookup for the extra-bit count on the serial chain
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

template <bool CELL16>
__global__ __launch_bounds__(64) void k_seq(const uint8_t *__restrict__ in, const SeqTask *__restrict__ tasks,
                                            uint32_t n_tasks, const uint32_t *__restrict__ fse_entries,
                                            uint64_t *__restrict__ recs, TileBase *__restrict__ tiles,
                                            BlockSum *sums)
{
    constexpr int NCH = CELL16 ? kSeqChains16 : kSeqChains32;
    constexpr int CELL_BYTES = CELL16 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *CT = (uint32_t *)(smem + (size_t)NCH * kSeqCellsPerChain * CELL_BYTES);  // [2][64]
    const int lane = threadIdx.x;
    const uint32_t tid = blockIdx.x * NCH + lane;
    const bool has = lane < NCH && tid < n_tasks;
    SeqTask t;
    if (has) t = tasks[tid];
    else {
        t.n_seq = 0; t.in_size = 0; t.ll_off = t.of_off = t.ml_off = 0; t.ll_log = t.of_log = t.ml_log = 0;
        t.in_off = 0; t.rec_off = 0; t.tile_off = 0; t.block = 0; t.hist_known = 0;
    }
    // constant tables
    CT[lane] = lane < 36 ? (c_ll_base[lane] | ((uint32_t)c_ll_extra[lane] << 24)) : 0u;
    CT[64 + lane] = lane < 53 ? (c_ml_base[lane] | ((uint32_t)c_ml_extra[lane] << 24)) : 0u;

    // stage the three tables of every chain of this wavefront
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
            for (uint32_t i = lane; i < n; i += 64) {
                uint32_t e = fse_entries[off[kind] + i];  // baseline(16) | nbits(8) | symbol(8)
                uint32_t baseline = e & 0xFFFF, nb = (e >> 16) & 0xFF, sym = e >> 24;
                if (CELL16) {
                    uint32_t next = (baseline + n) >> nb;
                    ((uint16_t *)smem)[base + i] = (uint16_t)(next | (sym << 10));
                } else {
                    uint32_t extra = kind == 0 ? c_ll_extra[min(sym, 35u)] : (kind == 1 ? c_ml_extra[min(sym, 52u)] : sym);
                    ((uint32_t *)smem)[base + i] = baseline | (nb << 10) | (extra << 14) | (sym << 19);
                }
            }
        }
    }
    __syncthreads();

    const uint32_t slot = (uint32_t)lane * kSeqCellsPerChain;
    const int alL = t.ll_log, alM = t.ml_log, alO = t.of_log;
    BackBits br;
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
        br.s = in; br.C = br.D = 0; br.ptr = 0; br.k = 0;
    }
    int h0, h1, h2;
    if (t.hist_known) { h0 = 1; h1 = 4; h2 = 8; }  // framedecompressor.go:48,59
    else { h0 = -1; h1 = -2; h2 = -3; }
    uint32_t litPos = 0, outPos = 0;
    uint64_t *myrec = recs + t.rec_off;
    TileBase *mytile = tiles + t.tile_off;

    const uint32_t nmax = wave_max_u32((has && status == MZD_OK) ? t.n_seq : 0u);
    for (uint32_t i = 0; i < nmax; i++) {
        const bool act = i < t.n_seq && status == MZD_OK;
        const bool lastseq = (i + 1 == t.n_seq);
        if (act && (i & 63) == 0) mytile[i >> 6] = TileBase{litPos, outPos};

        // ---- table cells for the three current states
        uint32_t symL, symM, symO, nbL, nbM, nbO, baseL, baseM, baseO, exL, exM;
        if (CELL16) {
            const uint16_t *c = (const uint16_t *)smem + slot;
            uint32_t xl = c[sL], xm = c[512 + sM], xo = c[1024 + sO];
            symL = xl >> 10; symM = xm >> 10; symO = xo >> 10;
            uint32_t nl = xl & 1023, nm = xm & 1023, no = xo & 1023;
            nbL = (uint32_t)alL - (31 - __builtin_clz(nl | 1));
            nbM = (uint32_t)alM - (31 - __builtin_clz(nm | 1));
            nbO = (uint32_t)alO - (31 - __builtin_clz(no | 1));
            baseL = (nl << nbL) - (1u << alL);
            baseM = (nm << nbM) - (1u << alM);
            baseO = (no << nbO) - (1u << alO);
        } else {
            const uint32_t *c = (const uint32_t *)smem + slot;
            uint32_t el = c[sL], em = c[512 + sM], eo = c[1024 + sO];
            baseL = el & 1023; nbL = (el >> 10) & 15; exL = (el >> 14) & 31; symL = el >> 19;
            baseM = em & 1023; nbM = (em >> 10) & 15; exM = (em >> 14) & 31; symM = em >> 19;
            baseO = eo & 1023; nbO = (eo >> 10) & 15; symO = eo >> 19;
        }
        const uint32_t cl = CT[symL], cm = CT[64 + symM];
        if (CELL16) { exL = cl >> 24; exM = cm >> 24; }
        const uint32_t exO = symO;
        if (lastseq) { nbL = 0; nbM = 0; nbO = 0; }  // no state update after the last sequence (sequences.go:178)
        const int total = (int)(exO + exM + exL + nbL + nbM + nbO);

        // ---- bits, in stream order: OF extra, ML extra, LL extra, LL state, ML state, OF state
        br.refill();
        uint32_t ofx, mlx, llx, aL, aM, aO;
        const bool slow = act && (br.k + total > 64);
        if (__any(slow)) {  // rare: very long offsets/lengths; refill between fields
            ofx = br.peek((int)exO); br.k += (int)exO; br.refill();
            mlx = br.peek((int)exM); br.k += (int)exM;
            llx = br.peek((int)exL); br.k += (int)exL; br.refill();
            aL = br.peek((int)nbL); br.k += (int)nbL;
            aM = br.peek((int)nbM); br.k += (int)nbM;
            aO = br.peek((int)nbO); br.k += (int)nbO;
        } else {
            uint64_t T = br.C << br.k;
            ofx = (uint32_t)((T >> 1) >> (63 - exO)); T <<= exO;
            mlx = (uint32_t)((T >> 1) >> (63 - exM)); T <<= exM;
            llx = (uint32_t)((T >> 1) >> (63 - exL)); T <<= exL;
            aL = (uint32_t)((T >> 1) >> (63 - nbL)); T <<= nbL;
            aM = (uint32_t)((T >> 1) >> (63 - nbM)); T <<= nbM;
            aO = (uint32_t)((T >> 1) >> (63 - nbO));
            br.k += total;
        }
        if (act) {
            rem -= total;
            if (rem < 0) status = MZD_ERR_SEQ_BITS;  // over-read (cursor would pass -1)
        }
        // next states: state = Baseline + Read(NumberOfBits) (fse.go:282-290), order LL, ML, OF
        // (masks are no-ops for valid tables; they keep idle / failed lanes inside their LDS slot)
        sL = (baseL + aL) & 511; sM = (baseM + aM) & 511; sO = (baseO + aO) & 255;

        // ---- values (sequences.go:99-120)
        const uint32_t ofv = (1u << exO) + ofx;
        const uint32_t ML = (cm & 0xFFFFFF) + mlx;
        const uint32_t LL = (cl & 0xFFFFFF) + llx;

        // ---- repeat-offset resolution (sequence_execution.go:65-114) on concrete-or-symbolic history
        int off, n0, n1 = h1, n2 = h2;
        if (ofv > 3) {
            off = (int)(ofv - 3);
            if (ofv - 3 >= kRecOffSymbolic && act) status = MZD_ERR_UNSUPPORTED;  // offset >= 2^28
            n2 = h1; n1 = h0;
        } else {
            const int idx = (int)ofv - 1 + (LL == 0 ? 1 : 0);
            off = idx == 0 ? h0 : (idx == 1 ? h1 : (idx == 2 ? h2 : hist_dec(h0)));
            if (off == 0 && act) status = MZD_ERR_OFFSET;
            if (idx >= 2) n2 = h1;
            if (idx >= 1) n1 = h0;
        }
        n0 = off;
        if (act) { h0 = n0; h1 = n1; h2 = n2; }  // finished lanes keep their final history
        if (act && status == MZD_OK) {
            const uint32_t offfield = off > 0 ? (uint32_t)off : (kRecOffSymbolic | (uint32_t)(-off - 1));
            myrec[i] = (uint64_t)LL | ((uint64_t)ML << kRecMlShift) | ((uint64_t)offfield << kRecOffShift);
            litPos += LL;
            outPos += LL + ML;
            if (outPos > kBlockMax) status = MZD_ERR_CORRUPT_SIZES;  // a block regenerates <= 128 KiB
        }
    }
    if (has && t.n_seq > 0) {
        if (status == MZD_OK && rem != 0) status = MZD_ERR_SEQ_BITS;  // sequences.go:197-204
        BlockSum *bs = &sums[t.block];
        bs->lit_total = litPos;
        bs->out_total = outPos;
        bs->hist[0] = h0;
        bs->hist[1] = h1;
        bs->hist[2] = h2;
        if (status != MZD_OK) atomicCAS(&bs->status, MZD_OK, status);
    }
}

// ------------------------------------------------------------------------------------------
// k_exec: sequence execution + Raw/RLE blocks.  One workgroup (16 wavefronts) per frame.
//
// LDS layout:  [ block buffer 128 KiB + 32 ][ validity bitmap 16 KiB ][ control ]
//   * the block being regenerated lives in LDS, shifted by `mis` so that LDS and HBM agree on
//     16-byte alignment; matches into EARLIER blocks of the frame read HBM (already flushed);
//   * validity bitmap: bit p set <=> output byte p of the block has been written.  Sequence
//     execution is a DATAFLOW: a match copy runs as soon as exactly its source bytes are valid,
//     so 64-sequence tiles execute on 16 wavefronts with no ordering between tiles and no false
//     dependencies (the reference's serial loop sequence_execution.go:16-53 is the degenerate
//     schedule of the same dataflow graph).  Progress: the earliest unexecuted match of a block
//     always has all its sources valid, and every wavefront walks its tiles in increasing order.
//   * byte-misaligned LDS dword READS are replayed 64x on gfx950 (tools/ubench), misaligned dword
//     WRITES are not: copies read aligned dwords, funnel-shift with v_alignbyte, write misaligned.

struct ExecShared {
    int error;
    uint32_t pad[3];
};

constexpr uint32_t kExecBufBytes = kBlockMax + 32;
constexpr uint32_t kExecMapWords = kBlockMax / 32 + 4;  // one bit per output byte (+ slack for the w+1 probe)
constexpr uint32_t kExecLdsBytes = kExecBufBytes + kExecMapWords * 4 + 16;

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

// bit mask helpers for the validity bitmap: bits [bit, bit+n) of a 64-bit window, n <= 32
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

__global__ __launch_bounds__(1024) void k_exec(const uint8_t *__restrict__ in, uint8_t *out_blob,
                                               const DFrame *__restrict__ frames, const DBlock *__restrict__ blocks,
                                               const BlockSum *__restrict__ sums, const uint64_t *__restrict__ recs,
                                               const TileBase *__restrict__ tiles, const uint8_t *__restrict__ litbuf,
                                               int32_t *frame_status, uint64_t *frame_out_len)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *buf = smem;
    uint32_t *vmap = (uint32_t *)(smem + kExecBufBytes);
    ExecShared *sh = (ExecShared *)(smem + kExecBufBytes + kExecMapWords * 4);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const DFrame fr = frames[blockIdx.x];
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
        int err = bsum.status;
        if (err == MZD_OK && litTotal > b.lit_regen) err = MZD_ERR_LITERALS;  // sequence_execution.go:27-29
        const uint32_t blockOut = seqOut + (b.lit_regen - min(litTotal, b.lit_regen));
        if (err == MZD_OK && blockOut > kBlockMax) err = MZD_ERR_CORRUPT_SIZES;
        if (err == MZD_OK && outPos + blockOut > fr.out_capacity) err = MZD_ERR_DST_FULL;
        if (err != MZD_OK) {
            if (tid == 0) sh->error = err;
            __syncthreads();
            break;
        }
        const uint32_t mis = (uint32_t)((uintptr_t)(out + outPos) & 15);
        uint8_t *lbuf = buf + mis;  // lbuf[p] = block-relative output byte p
        const uint8_t *lits = (b.lit_type == MZD_LIT_HUF ? litbuf : in) + b.lit_src;
        const bool litRle = b.lit_type == MZD_LIT_RLE;
        const uint32_t rleWord = litRle ? lits[0] * 0x01010101u : 0;
        const uint32_t ntiles = (b.n_seq + 63) >> 6;
        if (ntiles) {
            // clear the validity bits the sequences of this block will set
            const uint32_t words = (seqOut + 31) >> 5;
            for (uint32_t i = tid; i < words; i += nthr) vmap[i] = 0;
        }
        __syncthreads();

        const uint64_t *brec = recs + b.rec_off;
        const TileBase *btile = tiles + b.tile_off;

        // software pipeline: records / tile bases of the NEXT tile are loaded while the current one runs
        uint32_t tile = wave;
        uint64_t rec_n = 0;
        TileBase tb_n{0, 0};
        if (tile < ntiles) {
            const uint32_t si = tile * 64 + lane;
            rec_n = si < b.n_seq ? brec[si] : 0ull;
            tb_n = btile[tile];
        }
        for (; tile < ntiles; tile += nwaves) {
            const uint64_t rec = rec_n;
            const TileBase tb = tb_n;
            {
                const uint32_t nt = tile + nwaves;
                if (nt < ntiles) {
                    const uint32_t si = nt * 64 + lane;
                    rec_n = si < b.n_seq ? brec[si] : 0ull;
                    tb_n = btile[nt];
                }
            }
            const bool valid = tile * 64 + lane < b.n_seq;
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
            const uint32_t dstM = outEnd - ML, dstL = dstM - LL, srcL = litEnd - LL;
            const bool bad = valid && ML > 0 && (off <= 0 || (uint64_t)off > outPos + dstM);  // ringbuffer.go:206-214
            if (__any(bad)) {
                if (lane == 0) atomicMax(&sh->error, MZD_ERR_OFFSET);
            }
            const int srcM = (int)dstM - off;  // block-relative; negative = earlier blocks (HBM)

            // ---- literals (sequence_execution.go:19-34): they depend on nothing
            {
                const uint32_t sLL = (valid && LL <= 16) ? LL : 0;
                if (__any(sLL != 0)) {
                    uint32_t w0 = rleWord, w1 = rleWord, w2 = rleWord, w3 = rleWord, wt = rleWord;
                    if (!litRle && sLL) {
                        const U128U v = *(const U128U *)(lits + srcL);
                        w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
                        if (sLL >= 4) wt = ld32u_g(lits + srcL + sLL - 4);
                    }
                    uint8_t *d = lbuf + dstL;
                    if (sLL >= 4) st32u_l(d, w0);
                    if (sLL >= 8) st32u_l(d + 4, w1);
                    if (sLL >= 12) st32u_l(d + 8, w2);
                    if (sLL >= 16) st32u_l(d + 12, w3);
                    if (sLL >= 4 && (sLL & 3)) st32u_l(d + sLL - 4, wt);  // overlapped tail dword
                    if (sLL > 0 && sLL < 4) {
                        d[0] = (uint8_t)w0;
                        if (sLL > 1) d[1] = (uint8_t)(w0 >> 8);
                        if (sLL > 2) d[2] = (uint8_t)(w0 >> 16);
                    }
                    if (sLL) publish(vmap, dstL, sLL);
                }
                uint64_t longs = __ballot(valid && LL > 16);
                while (longs) {
                    const int src = __builtin_ctzll(longs);
                    longs &= longs - 1;
                    const uint32_t n = (uint32_t)__shfl((int)LL, src, 64);
                    const uint32_t d = (uint32_t)__shfl((int)dstL, src, 64);
                    const uint32_t s = (uint32_t)__shfl((int)srcL, src, 64);
                    if (litRle) for (uint32_t j = lane; j < n; j += 64) lbuf[d + j] = (uint8_t)rleWord;
                    else for (uint32_t j = lane; j < n; j += 64) lbuf[d + j] = lits[s + j];
                    // publish: lane k owns bytes [32k, 32k+32) of the run per round of 2 KiB
                    for (uint32_t j = lane * 32; j < n; j += 64 * 32) publish(vmap, d + j, min(32u, n - j));
                }
            }

            // ---- matches (sequence_execution.go:43-49, ringbuffer.go:242-277) as dataflow
            bool pending = valid && ML > 0 && !bad;
            const uint32_t span = min(ML, (uint32_t)max(off, 1));  // bytes that are true sources
            const bool isShort = ML <= 32;
            const bool overlap = (uint32_t)off < ML;
            uint32_t spins = 0;
            while (__any(pending)) {
                // readiness of short matches: all source bytes at positions >= 0 must be valid
                bool ready = false;
                if (pending && isShort) {
                    const int s0 = max(srcM, 0), s1 = srcM + (int)span;  // [s0, s1) inside this block
                    if (s1 <= s0) ready = true;
                    else {
                        const uint64_t m = span_mask((uint32_t)s0 & 31, (uint32_t)(s1 - s0));
                        const uint32_t w = (uint32_t)s0 >> 5;
                        const uint32_t v0 = __hip_atomic_load(&vmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const uint32_t v1 = __hip_atomic_load(&vmap[w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const uint64_t v = (uint64_t)v0 | ((uint64_t)v1 << 32);
                        ready = (v & m) == m;
                    }
                }
                asm volatile("" ::: "memory");  // data reads below stay below the validity reads
                bool progressed = false;
                // (1) short non-overlapping matches entirely inside this block: dword path
                const bool fast = ready && !overlap && srcM >= 0;
                if (__any(fast)) {
                    progressed = true;
                    uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                    uint32_t t0 = 0, t1 = 0;
                    // aligned dwords covering the source bytes [sbyte, sbyte + ML) of the LDS buffer
                    const uint32_t sbyte = fast ? (mis + (uint32_t)srcM) : 0;
                    const uint32_t sa = sbyte & 3;
                    const uint32_t *base = (const uint32_t *)(buf + (sbyte & ~3u));
                    const uint32_t ndw = fast ? ((sa + ML + 3) >> 2) : 0;
#pragma unroll
                    for (int j = 0; j < 9; j++)
                        if ((uint32_t)j < ndw) w[j] = base[j];
                    const uint32_t tb_byte = sbyte + ML - 4;  // tail dword = source bytes [ML-4, ML)
                    if (fast && ML >= 4 && (ML & 3)) {
                        const uint32_t *tp = (const uint32_t *)(buf + (tb_byte & ~3u));
                        t0 = tp[0];
                        t1 = tp[1];
                    }
                    if (fast) {
                        uint8_t *d = lbuf + dstM;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if ((uint32_t)(4 * j + 4) <= ML) st32u_l(d + 4 * j, __builtin_amdgcn_alignbyte(w[j + 1], w[j], sa));
                        }
                        if (ML >= 4 && (ML & 3)) st32u_l(d + ML - 4, __builtin_amdgcn_alignbyte(t1, t0, tb_byte & 3));
                        if (ML < 4) {  // ML == 3 (or less on odd streams)
                            const uint32_t x = __builtin_amdgcn_alignbyte(w[1], w[0], sa);
                            if (ML > 0) d[0] = (uint8_t)x;
                            if (ML > 1) d[1] = (uint8_t)(x >> 8);
                            if (ML > 2) d[2] = (uint8_t)(x >> 16);
                        }
                        publish(vmap, dstM, ML);
                    }
                }
                // (2) short matches that overlap themselves or reach into earlier blocks: byte loop
                const bool slowb = ready && !fast;
                if (__any(slowb)) {
                    progressed = true;
                    const uint32_t n = slowb ? ML : 0;
                    const uint32_t nmax = wave_max_u32(n);
                    for (uint32_t j = 0; j < nmax; j++) {
                        if (j < n) {
                            const int q = srcM + (int)j;
                            const uint8_t v = q >= 0 ? lbuf[q] : out[(int64_t)outPos + q];
                            lbuf[dstM + j] = v;
                        }
                    }
                    if (slowb) publish(vmap, dstM, ML);
                }
                pending = pending && !ready;
                // (3) at most one long match per iteration, whole wavefront, non-blocking readiness test
                uint64_t longs = __ballot(pending && !isShort);
                if (longs) {
                    const int src = __builtin_ctzll(longs);
                    const uint32_t n = (uint32_t)__shfl((int)ML, src, 64);
                    const uint32_t d = (uint32_t)__shfl((int)dstM, src, 64);
                    const int s = __shfl(srcM, src, 64);
                    const uint32_t o = (uint32_t)__shfl(off, src, 64);
                    const uint32_t sp2 = (uint32_t)__shfl((int)span, src, 64);
                    // readiness: every valid-map word overlapping [max(s,0), s+sp2) must be complete there
                    bool ok = true;
                    const int s0 = max(s, 0), s1 = s + (int)sp2;
                    if (s1 > s0) {
                        const uint32_t wf = (uint32_t)s0 >> 5, wl = (uint32_t)(s1 - 1) >> 5;
                        for (uint32_t wi = wf + lane; wi <= wl; wi += 64) {
                            uint32_t need = 0xFFFFFFFFu;
                            if (wi == wf) need &= 0xFFFFFFFFu << ((uint32_t)s0 & 31);
                            if (wi == wl) need &= 0xFFFFFFFFu >> (31 - ((uint32_t)(s1 - 1) & 31));
                            const uint32_t v = __hip_atomic_load(&vmap[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            ok = ok && ((v & need) == need);
                        }
                    }
                    asm volatile("" ::: "memory");
                    if (__all(ok)) {
                        progressed = true;
                        if (o >= 64) {
                            // each 64-byte step only reads bytes written by earlier steps (in-order LDS)
                            for (uint32_t j = lane; j < n; j += 64) {
                                const int q = s + (int)j;
                                const uint8_t v = q >= 0 ? lbuf[q] : out[(int64_t)outPos + q];
                                lbuf[d + j] = v;
                            }
                        } else {
                            // overlapping: periodic fill from the (final) pattern [s, s+o)
                            uint32_t r = (uint32_t)lane % o;
                            const uint32_t stepr = 64 % o;
                            for (uint32_t j = lane; j < n; j += 64) {
                                const int q = s + (int)r;
                                const uint8_t v = q >= 0 ? lbuf[q] : out[(int64_t)outPos + q];
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
                    if ((++spins & 15) == 0 &&
                        __hip_atomic_load(&sh->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != MZD_OK)
                        break;  // corrupt input: a skipped match would never validate its bytes
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        // ---- literals after the last sequence (sequence_execution.go:55-59)
        {
            const uint32_t rest = b.lit_regen - litTotal;
            uint8_t *d = lbuf + seqOut;
            if (litRle) {
                for (uint32_t j = tid; j < rest; j += nthr) d[j] = (uint8_t)rleWord;
            } else {
                const uint8_t *s = lits + litTotal;
                const uint32_t n4 = rest >> 2;
                for (uint32_t j = tid; j < n4; j += nthr) st32u_l(d + 4 * j, ld32u_g(s + 4 * j));
                for (uint32_t j = (n4 << 2) + tid; j < rest; j += nthr) d[j] = s[j];
            }
        }
        __syncthreads();
        // ---- the block leaves for HBM: head bytes, aligned 16-byte body, tail bytes
        {
            uint8_t *dst = out + outPos;
            const uint32_t head = min(blockOut, (16u - mis) & 15u);
            if ((uint32_t)tid < head) dst[tid] = lbuf[tid];
            const uint32_t body = (blockOut - head) >> 4;
            const uint4 *lsrc = (const uint4 *)(lbuf + head);  // 16-byte aligned in LDS by construction
            uint4 *gdst = (uint4 *)(dst + head);
            for (uint32_t i = tid; i < body; i += nthr) gdst[i] = lsrc[i];
            for (uint32_t i = head + (body << 4) + tid; i < blockOut; i += nthr) dst[i] = lbuf[i];
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
        frame_status[blockIdx.x] = e;
        frame_out_len[blockIdx.x] = outPos;
    }
}

}  // namespace mzd
