// mzd_kernels.hip -- the zstd block-decode hot path as hand-written HIP for gfx950: what every stage shares (helpers, the reverse bit
// reader, k_init) and the table builds (k_fse_build, k_huf_build).  Round 6 split the stages into files of their own: mzd_huf.hip /
// mzd_huf_w.hip, mzd_seq.hip / mzd_seq_q4.hip, mzd_exec.hip / mzd_exec_c.hip / mzd_exec_blk.hip (mzd_exec_b.hip: helpers + a test kernel),
// mzd_parse.hip, mzd_util.hip -- mzd_api.hip includes them in this order.  The overview below is of all of them.
// (MI355X, CDNA4: 64-wide wavefronts, 160 KiB LDS per CU).  Integer / table work only:
// no MFMA.  Three stages, one kernel each (plus a trivial init):
//
//   k_huf   Huffman literal decode      replaces structure/huffman.go:221-264 DecodeStream
//                                       (+ the 1/4-stream dispatch literals.go:295-371)
//   k_seq_pipe / k_seq  FSE sequence decode   replaces structure/sequences.go:126-206 DecodeSequences,
//                                       :64-123 DecodeSequence, fse/fse.go:253-290 state accessors,
//                                       and folds in sequence_execution.go:65-114 nextOffset
//   k_exec  sequence execution          replaces decompression/sequence_execution.go:14-63 and
//                                       ringbuffer.go:102-277 Push/Repeat/RepeatBeforeIndex, plus the
//                                       Raw / RLE block arms framedecompressor.go:211-215,229-241
//
// Mapping (see DESIGN.md for the reasoning and the roofline of each):
//   k_huf   one LANE per Huffman stream; the 4 streams of a literals section sit in 4 adjacent
//           lanes and share one decode table staged in LDS; 16 sections per wavefront.
//   k_seq_pipe  one LANE per block = one serial LL/ML/OF state chain; each chain's three FSE tables
//           live in LDS (that is what bounds the number of resident chains to one wavefront per CU).
//           The step is a three-stage pipeline across the CU's SIMDs: only the recurrence stays on
//           the chain wavefront (hand-scheduled ISA), field extraction and record/history work
//           follow in two more wavefronts through LDS queues.  k_seq is the two-wavefront
//           predecessor (fallback for blobs >= 4 GiB).
//   k_exec  one WORKGROUP per frame, up to 16 per CU; the chunk (8 KiB) of the block being regenerated
//           lives in LDS, older output is final in the frame's HBM slab; 64-sequence tiles run as a
//           dataflow on a per-byte validity bitmap, finished chunks leave in aligned 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/mzd.h"
#include "mzd_device.h"

namespace mzd {

// ------------------------------------------------------------------------------------------
// small helpers

struct __attribute__((packed, aligned(1))) U64U { uint64_t v; };
struct __attribute__((packed, aligned(1))) U32U { uint32_t v; };
struct __attribute__((packed, aligned(1))) U128U { uint32_t x, y, z, w; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { return ((const U64U *)p)->v; }

// Loads / stores of data that is read (written) ONCE per pass, with the `nt` cache-policy bit: the line is served as usual but
// is the first to leave L2 again, so that what the pass re-reads -- a frame's own recent output -- stays.  NT = false: plain.
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef uint64_t u64_u __attribute__((aligned(1)));
template <bool NT>
__device__ __forceinline__ U128U ld128u_once(const uint8_t *p)
{
    if (!NT) return *(const U128U *)p;
    const u32x4 v = __builtin_nontemporal_load((const u32x4_u *)p);
    return U128U{v.x, v.y, v.z, v.w};
}
template <bool NT>
__device__ __forceinline__ void st128u_once(uint8_t *p, const U128U &v)
{
    if (!NT) {
        *(U128U *)p = v;
        return;
    }
    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, (u32x4_u *)p);
}
template <bool NT>
__device__ __forceinline__ uint64_t ld64u_once(const uint8_t *p)
{
    if (!NT) return ((const U64U *)p)->v;
    return __builtin_nontemporal_load((const u64_u *)p);
}
template <bool NT>
__device__ __forceinline__ uint64_t ld64_once(const uint64_t *p)
{
    if (!NT) return *p;
    return __builtin_nontemporal_load(p);
}

// Touch a line: an ordinary load (it allocates in the CU's vL1D and in L2; a `volatile` access would
// be emitted system-coherent, sc0 sc1, and bypass the vL1D) whose result is waited for and dropped.
__device__ __forceinline__ void touch_line(const uint8_t *p)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
}

// The same without waiting: the result lands in v255, which nothing else uses (fire and forget; only for
// wavefronts that never wait on vmcnt themselves, where a pending miss then delays nobody).
__device__ __forceinline__ void touch_line_nowait(const uint8_t *p)
{
    asm volatile("global_load_dword v255, %0, off" : : "v"(p) : "memory", "v255");
}

// wave votes on a bool without the int round trip of __any / __ballot (v_cndmask 0/1 + v_cmp per vote): the condition's
// lane mask is the ballot
__device__ __forceinline__ uint64_t wave_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
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
    b.huf_err = 0xFFFFFFFFu;
    b.reach = 0xFFFFFFFFu;
    sums[i] = b;
}

// ------------------------------------------------------------------------------------------
// k_fse_build: FSE decoding tables from their normalised counts, on the device (SURVEY 8f #1;
// replaces fse/fse.go:136-230 BuildDecodingTable for tables that arrive as MZD_FSE_FROM_COUNTS).
// It also lays the tables out on the device (tables that arrive built are copied).  Run once per
// batch at upload.  One LANE per table: the spread walk and the per-symbol "next"
// counters are serial within a table and a batch has hundreds of thousands of tables.  The
// symbols of the table under construction and the counters live in LDS ([index][lane], so the 64
// lanes of a wavefront hit distinct banks); global memory is read once (the counts) and written
// once (the finished cells baseline:16 | nbits:8 | symbol:8).
__global__ __launch_bounds__(64) void k_fse_build(const FseBuildDesc *__restrict__ tabs, uint32_t n_tabs,
                                                  const uint32_t *__restrict__ src, uint32_t *__restrict__ cells)
{
    __shared__ uint8_t sym[512][64];     // cell -> symbol
    __shared__ int16_t cnt[64][64];      // symbol -> normalised count (-1 == "less than one")
    __shared__ uint16_t nextv[64][64];   // symbol -> next state value to hand out (fse.go:192-213)
    const int lane = threadIdx.x;
    const uint32_t t = blockIdx.x * 64 + lane;
    if (t >= n_tabs) return;
    const FseBuildDesc d = tabs[t];
    const int lg = d.acc_log, size = 1 << lg;
    uint32_t *c = cells + d.dst_off;
    if (!d.ok) {  // rejected at upload: blocks that use it carry a status, nobody reads these cells
        for (int i = 0; i < size; i++) c[i] = 0;
        return;
    }
    if (d.n_sym == 0) {  // arrived built (predefined, RLE, or a host that builds its own tables): copy
        for (int i = 0; i < size; i++) c[i] = src[d.src_off + i];
        return;
    }
    const int nsym = min((int)d.n_sym, 64);
    for (int s = 0; s < nsym; s += 2) {
        const uint32_t w = src[d.src_off + (s >> 1)];
        cnt[s][lane] = (int16_t)(w & 0xFFFF);
        if (s + 1 < nsym) cnt[s + 1][lane] = (int16_t)(w >> 16);
    }
    // "less than one" symbols take the top cells (fse.go:146-155)
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        const int n = cnt[s][lane];
        if (n == -1) {
            sym[max(high, 0)][lane] = (uint8_t)s;
            high--;
            nextv[s][lane] = 1;
        } else {
            nextv[s][lane] = (uint16_t)n;
        }
    }
    // the others are spread with step size/2 + size/8 + 3 over the cells below (fse.go:160-184); the
    // upload check guarantees that the counts add up to the table size, so every cell is visited once
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        const int n = cnt[s][lane];
        for (int i = 0; i < n; i++) {
            sym[pos][lane] = (uint8_t)s;
            int guard = 0;
            do {
                pos = (pos + step) & mask;
            } while (pos > high && ++guard <= size);
        }
    }
    // per cell, in index order: nbits = acc_log - highbit(next), baseline = (next << nbits) - size
    for (int i = 0; i < size; i++) {
        const uint32_t s = sym[i][lane] & 63;
        const uint32_t n = nextv[s][lane];
        nextv[s][lane] = (uint16_t)(n + 1);
        const uint32_t nb = (uint32_t)lg - (31u - (uint32_t)__builtin_clz(n | 1));
        c[i] = (((n << nb) - (uint32_t)size) & 0xFFFF) | (nb << 16) | (s << 24);
    }
}

// ------------------------------------------------------------------------------------------
// k_huf_build: Huffman decode tables from their weights, on the device (replaces the table fill
// of structure/huffman.go:112-190 for tables that arrive as MZD_HUF_FROM_WEIGHTS; the weight
// decode itself, huffman.go:40-107 / fse.go:307-390, is a serial two-state stream and stays with
// the host's header parse).  Once per batch at upload, one LANE per table.
__global__ __launch_bounds__(64) void k_huf_build(const HufBuildDesc *__restrict__ tabs, uint32_t n_tabs,
                                                  const uint16_t *__restrict__ src, uint16_t *__restrict__ cells)
{
    __shared__ uint8_t len[256][64];  // symbol -> code length (0: absent)
    const int lane = threadIdx.x;
    const uint32_t t = blockIdx.x * 64 + lane;
    if (t >= n_tabs) return;
    const HufBuildDesc d = tabs[t];
    const int mb = d.ok ? d.max_bits : 1, size = 1 << mb;
    uint16_t *c = cells + d.dst_off;
    if (!d.ok) {  // rejected at upload: blocks that use it carry a status
        c[0] = 0; c[1] = 0;
        return;
    }
    if (d.n_weights == 0) {  // arrived built: copy
        for (int i = 0; i < size; i++) c[i] = src[d.src_off + i];
        return;
    }
    // code length = MaxBits + 1 - weight; the last symbol gets what is left of 2^MaxBits (huffman.go:125-131)
    const int nw = d.n_weights;
    uint32_t sum = 0;
    for (int s = 0; s < nw; s += 2) {
        const uint32_t e = src[d.src_off + (s >> 1)];  // symbol byte (even weight) | nbits byte (odd weight) << 8
        const uint32_t w0 = e & 0xFF, w1 = e >> 8;
        len[s][lane] = w0 ? (uint8_t)(mb + 1 - w0) : 0;
        if (w0) sum += 1u << (w0 - 1);
        if (s + 1 < nw) {
            len[s + 1][lane] = w1 ? (uint8_t)(mb + 1 - w1) : 0;
            if (w1) sum += 1u << (w1 - 1);
        }
    }
    const uint32_t left = (1u << mb) - sum;  // a power of two (checked at upload)
    len[nw][lane] = (uint8_t)(mb + 1 - (32 - __builtin_clz(left | 1)));
    // longest codes first from cell 0, ascending symbol inside a length (huffman.go:163-187)
    int at = 0;
    for (int l = mb; l >= 1; l--) {
        const int span = 1 << (mb - l);
        for (int s = 0; s <= nw; s++) {
            if (len[s][lane] != l) continue;
            const uint16_t cell = (uint16_t)(s | (l << 8));  // {symbol, nbits}
            for (int j = 0; j < span && at + j < size; j++) c[at + j] = cell;
            at += span;
        }
    }
}

}  // namespace mzd
