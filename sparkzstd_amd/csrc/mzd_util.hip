// mzd_util.hip -- k_xxh64 (content checksum: an extension), k_copy_ceiling (the measured copy ceiling of bench.py), k_test_backbits
// (the reverse bit reader on the reference's vectors).  Split out of mzd_kernels.hip in round 6.
#pragma once

namespace mzd {

// ------------------------------------------------------------------------------------------
// k_xxh64: content checksum of the regenerated frames (zstd frame format: the 4
// bytes after the last block are the low half of XXH64(content, seed 0)).  An EXTENSION: the
// reference never reads the checksum (framereader.go:84-94, Readme.md:62), so this runs only with
// mzd_options.verify_checksum.  Pure streaming read of the output: HBM-bound.
// XXH64 keeps four accumulators, accumulator k eats bytes [32 s + 8 k, +8) of stripe s: FOUR LANES
// per frame, one accumulator each (16 frames per wavefront), four stripes = one 128-byte line per
// quad in flight; lane 0 of the quad merges and finishes the < 32 tail bytes.
#ifndef MZD_XXH_UNROLL
#define MZD_XXH_UNROLL 8
#endif
__device__ __forceinline__ uint64_t xxh_rotl(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
constexpr uint64_t kXP1 = 0x9E3779B185EBCA87ull, kXP2 = 0xC2B2AE3D27D4EB4Full, kXP3 = 0x165667B19E3779F9ull,
                   kXP4 = 0x85EBCA77C2B2AE63ull, kXP5 = 0x27D4EB2F165667C5ull;
__device__ __forceinline__ uint64_t xxh_round(uint64_t acc, uint64_t in) { return xxh_rotl(acc + in * kXP2, 31) * kXP1; }
__device__ __forceinline__ uint64_t xxh_merge(uint64_t h, uint64_t v) { return (h ^ xxh_round(0, v)) * kXP1 + kXP4; }

__global__ __launch_bounds__(64) void k_xxh64(const uint8_t *__restrict__ out_blob, const DFrame *__restrict__ frames,
                                              uint32_t n_frames, int32_t *frame_status, const uint64_t *__restrict__ frame_out_len)
{
    const int lane = threadIdx.x, q = lane & 3;
    const uint32_t f = blockIdx.x * 16 + (lane >> 2);
    const bool in_range = f < n_frames;
    DFrame fr{};
    if (in_range) fr = frames[f];
    // only frames that carry a checksum and decoded without error
    const bool check = in_range && fr.has_checksum && frame_status[f] == MZD_OK;
    const uint64_t n = check ? frame_out_len[f] : 0;
    const uint8_t *p = out_blob + fr.out_offset;  // slabs are 256-byte aligned
    const uint64_t stripes = n >> 5;
    uint64_t v = q == 0 ? kXP1 + kXP2 : (q == 1 ? kXP2 : (q == 2 ? 0ull : 0ull - kXP1));
    const uint64_t *pp = (const uint64_t *)p + q;
    uint64_t s = 0;
    constexpr int U = MZD_XXH_UNROLL;  // stripes per batch: U loads of 8 bytes per lane in flight while U are mixed in
    if (stripes >= U) {
        uint64_t a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = pp[4 * u];
        for (; s + 2 * U <= stripes; s += U) {
            const uint64_t *nx = pp + 4 * (s + U);
            uint64_t b[U];
#pragma unroll
            for (int u = 0; u < U; u++) b[u] = nx[4 * u];
#pragma unroll
            for (int u = 0; u < U; u++) v = xxh_round(v, a[u]);
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = b[u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) v = xxh_round(v, a[u]);
        s += U;
    }
    for (; s < stripes; s++) v = xxh_round(v, pp[4 * s]);
    // convergence on lane 0 of the quad
    const int q0 = lane & ~3;
    const uint64_t v1 = __shfl(v, q0, 64), v2 = __shfl(v, q0 + 1, 64), v3 = __shfl(v, q0 + 2, 64), v4 = __shfl(v, q0 + 3, 64);
    if (q != 0 || !check) return;
    uint64_t h;
    if (n >= 32) {
        h = xxh_rotl(v1, 1) + xxh_rotl(v2, 7) + xxh_rotl(v3, 12) + xxh_rotl(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    } else {
        h = kXP5;  // seed 0
    }
    h += n;
    const uint8_t *t = p + (stripes << 5), *end = p + n;
    while (end - t >= 8) {
        h ^= xxh_round(0, ld64u(t));
        h = xxh_rotl(h, 27) * kXP1 + kXP4;
        t += 8;
    }
    if (end - t >= 4) {
        h ^= (uint64_t)((const U32U *)t)->v * kXP1;
        h = xxh_rotl(h, 23) * kXP2 + kXP3;
        t += 4;
    }
    while (t < end) {
        h ^= (uint64_t)(*t++) * kXP5;
        h = xxh_rotl(h, 11) * kXP1;
    }
    h ^= h >> 33; h *= kXP2; h ^= h >> 29; h *= kXP3; h ^= h >> 32;
    if ((uint32_t)h != fr.checksum) frame_status[f] = MZD_ERR_CHECKSUM;
}

// ------------------------------------------------------------------------------------------
// k_copy_ceiling: the achievable-copy ceiling the roofline fractions are quoted against next to the
// 8 TB/s nominal peak.  Plain streaming kernel, 16 bytes per lane, grid-stride, four
// independent loads in flight per lane: reads the n_read 16-byte words of src once and writes
// n_write words of dst once (words past n_read repeat the lane's last loaded value: a write-only
// stream, like an RLE fill; words past n_write are only read) -- the algorithmic bytes of a pass, C in and D out, and nothing else.
__global__ __launch_bounds__(256) void k_copy_ceiling(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst,
                                                       uint64_t n_read, uint64_t n_write)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    u32x4 v[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const uint64_t n_both = n_read > n_write ? n_read : n_write;  // (C > D, e.g. Raw blocks with headers: the extra words are only read)
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_both; i += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t j = i + u * stride;
            if (j < n_read) v[u] = __builtin_nontemporal_load(src + j);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t j = i + u * stride;
            if (j < n_write) __builtin_nontemporal_store(v[u], dst + j);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_test_backbits: the device's backward bit reader (BackBits, row B0 of SURVEY 8a) driven like
// bitstream/reversebitstream_test.go drives Reversebitstream: a list of Read(n) calls on a raw
// stream (no padding marker), values and BitsStillInStream() back.  Test hook only; one lane.
__global__ void k_test_backbits(const uint8_t *stream, uint32_t len, const uint8_t *nbits, uint32_t n_reads,
                                uint64_t *values, int64_t *bits_still)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    BackBits br;
    br.s = stream;
    br.ptr = (int)len - 8;
    br.C = br.load_below(br.ptr);
    br.D = br.load_below(br.ptr - 8);
    br.k = 0;
    int64_t cursor = 8ll * len - 1;  // reversebitstream.go:9-11: index of the next bit
    for (uint32_t i = 0; i < n_reads; i++) {
        const int n = nbits[i];  // 0..32
        if (br.k + n > 56) br.refill();
        values[i] = br.peek(n);
        br.k += n;
        cursor -= n;
        bits_still[i] = cursor;  // reversebitstream.go:13-15
    }
}

}  // namespace mzd
