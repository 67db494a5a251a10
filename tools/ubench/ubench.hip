// ubench.hip -- micro-benchmarks that steer the kernel design (run on the GPU box):
//   1. LDS access cost by width and misalignment (correctness + cycles per wave-instruction)
//   2. DPP-based wave64 inclusive scan / min-reduce vs the shuffle versions
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct __attribute__((packed, aligned(1))) U32U { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64U { uint64_t v; };

template <int WIDTH, bool WRITE>
__global__ void k_lds(uint64_t *res, uint32_t *sink, int mis, int stride, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16384; i += blockDim.x) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    uint8_t *p = smem + mis + lane * stride;
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; it++) {
        asm volatile("" ::: "memory");  // keep the reads inside the loop
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint8_t *q = p + u * 4096;
            if (WRITE) {
                if (WIDTH == 1) *q = (uint8_t)(acc + u);
                else if (WIDTH == 4) ((U32U *)q)->v = acc + u;
                else ((U64U *)q)->v = acc + u;
            } else {
                if (WIDTH == 1) acc += *q;
                else if (WIDTH == 4) acc += ((U32U *)q)->v;
                else acc += (uint32_t)((U64U *)q)->v;
            }
        }
        if (WRITE) acc += it;
    }
    __syncthreads();
    uint64_t t1 = clock64();
    if (lane == 0) res[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + lane] = acc + smem[lane];
}

// correctness of misaligned LDS dword read/write
__global__ void k_lds_check(uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t b[1024];
    int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) b[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t ok = 1;
    for (int m = 0; m < 8; m++) {
        uint32_t v = ((U32U *)(b + m + lane * 9))->v;
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) w |= (uint32_t)b[m + lane * 9 + k] << (8 * k);
        ok &= (v == w);
        uint64_t v8 = ((U64U *)(b + m + lane * 9))->v;
        uint64_t w8 = 0;
        for (int k = 0; k < 8; k++) w8 |= (uint64_t)b[m + lane * 9 + k] << (8 * k);
        ok &= (v8 == w8);
    }
    __syncthreads();
    // misaligned writes
    ((U32U *)(b + 1 + lane * 13))->v = 0xA0B0C0D0u + lane;
    __syncthreads();
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) r |= (uint32_t)b[1 + lane * 13 + k] << (8 * k);
    ok &= (r == 0xA0B0C0D0u + lane);
    out[lane] = ok;
}

// ---- DPP scan
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, BOUND);
}
__device__ __forceinline__ uint32_t scan_dpp(uint32_t v)
{
    // row_shr:1,2,3 then row_shr:4 (bank mask e), row_shr:8 (bank mask c), row_bcast:15 (row mask a), row_bcast:31 (row mask c)
    uint32_t t;
    t = dpp<0x111, 0xf, 0xf, false>(0, v); v += t;   // row_shr:1
    t = dpp<0x112, 0xf, 0xf, false>(0, v); v += t;   // row_shr:2
    t = dpp<0x114, 0xf, 0xe, false>(0, v); v += t;   // row_shr:4
    t = dpp<0x118, 0xf, 0xc, false>(0, v); v += t;   // row_shr:8
    t = dpp<0x142, 0xa, 0xf, false>(0, v); v += t;   // row_bcast:15
    t = dpp<0x143, 0xc, 0xf, false>(0, v); v += t;   // row_bcast:31
    return v;
}
__device__ __forceinline__ uint32_t scan_shfl(uint32_t v, int lane)
{
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up((int)v, d, 64); if (lane >= d) v += y; }
    return v;
}
__global__ void k_scan(uint32_t *out, uint64_t *res, int iters)
{
    int lane = threadIdx.x;
    uint32_t x = (lane * 37 + 11) & 255;
    out[lane] = scan_dpp(x);
    out[64 + lane] = scan_shfl(x, lane);
    uint32_t a = x;
    uint64_t t0 = clock64();
    for (int i = 0; i < iters; i++) a = scan_dpp(a) & 1023;
    uint64_t t1 = clock64();
    for (int i = 0; i < iters; i++) a = scan_shfl(a, lane) & 1023;
    uint64_t t2 = clock64();
    out[128 + lane] = a;
    if (lane == 0) { res[0] = t1 - t0; res[1] = t2 - t1; }
}

int main()
{
    uint64_t *d_res; uint32_t *d_sink;
    CHECK(hipMalloc(&d_res, 64)); CHECK(hipMalloc(&d_sink, 1 << 20));
    uint32_t h[256];
    k_lds_check<<<1, 64>>>(d_sink);
    CHECK(hipMemcpy(h, d_sink, 256, hipMemcpyDeviceToHost));
    int ok = 1; for (int i = 0; i < 64; i++) ok &= h[i];
    printf("misaligned LDS b32/b64 read + b32 write correct: %d\n", ok);
    const int iters = 2000;
    auto run = [&](auto kern, const char *name, int mis, int stride, int threads) {
        uint64_t r;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
        kern<<<1, threads, 65536 + 4096>>>(d_res, d_sink, mis, stride, iters);
        hipDeviceSynchronize();
        hipMemcpy(&r, d_res, 8, hipMemcpyDeviceToHost);
        printf("%-14s mis=%d stride=%2d waves=%d : %.1f cycles per wave-instruction (per wave)\n", name, mis, stride, threads / 64, (double)r / (iters * 8.0));
    };
    for (int threads : {64, 1024}) {
        for (int stride : {4, 9, 11}) {
            for (int mis : {0, 1}) {
                run(k_lds<1, false>, "read u8", mis, stride, threads);
                run(k_lds<4, false>, "read b32", mis, stride, threads);
                run(k_lds<8, false>, "read b64", mis, stride, threads);
                run(k_lds<1, true>, "write b8", mis, stride, threads);
                run(k_lds<4, true>, "write b32", mis, stride, threads);
                run(k_lds<8, true>, "write b64", mis, stride, threads);
            }
        }
    }
    k_scan<<<1, 64>>>(d_sink, d_res, 10000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d_sink, 1024, hipMemcpyDeviceToHost));
    uint64_t r[2]; CHECK(hipMemcpy(r, d_res, 16, hipMemcpyDeviceToHost));
    int same = 1; for (int i = 0; i < 64; i++) same &= (h[i] == h[64 + i]);
    printf("dpp scan == shfl scan: %d ; dpp %.1f cycles, shfl %.1f cycles per scan\n", same, r[0] / 10000.0, r[1] / 10000.0);
    return 0;
}
