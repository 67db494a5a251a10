// gather.hip -- what the memory system does with k_exec_*'s staged-match loads: 16-byte loads at random byte addresses of a
// working set W (a frame's slab: written a while ago, so L2 / Infinity Cache / HBM by the number of frames in flight), ACT of 64
// lanes active, DEPTH loads in flight per lane, WAVES wavefronts per CU.  Prints loads/s, the 64-byte sectors they imply, and
// the time 420 M such loads (one config-4 pass: 30.7 staged matches x 13.7 M tiles) would take.
// build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(1))) U128U { uint32_t x, y, z, w; };
template <int DEPTH>
__global__ void k_gather(const uint8_t *base, uint64_t wmask, uint32_t iters, uint32_t act, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63;
    uint64_t s = (blockIdx.x * 977u + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    // every wavefront walks its own 128 KiB "slab" inside the working set (the frame it decodes), like the staged loads do
    const uint64_t slab = ((uint64_t)blockIdx.x * 131072ull) & wmask;
    if (lane < act) {
        for (uint32_t it = 0; it < iters; it++) {
            U128U v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const uint64_t off = (slab + ((s >> 33) & 131071ull)) & wmask;
                v[d] = *(const U128U *)(base + off);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; d++) acc += v[d].x ^ v[d].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const uint64_t cap = 2ull << 30;
    uint8_t *buf;
    uint32_t *sink;
    hipMalloc(&buf, cap + 64);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, cap + 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint64_t Ws[] = {16ull << 20, 128ull << 20, 640ull << 20, 2ull << 30};
    const int waves[] = {8, 20, 32};
    for (uint64_t W : Ws)
        for (int wv : waves)
            for (int depth : {1, 2, 4}) {
                const uint32_t blocks = 256u * wv, iters = 2048 / depth, act = 31;
                auto launch = [&]() {
                    if (depth == 1) k_gather<1><<<blocks, 64>>>(buf, W - 1, iters, act, sink);
                    else if (depth == 2) k_gather<2><<<blocks, 64>>>(buf, W - 1, iters, act, sink);
                    else k_gather<4><<<blocks, 64>>>(buf, W - 1, iters, act, sink);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double loads = (double)blocks * act * iters * depth;
                printf("W %5llu MiB  waves/CU %2d  in flight per lane %d : %7.1f G loads/s  (%.2f TB/s of 64-byte sectors)  -> 420 M loads in %.2f ms\n",
                       (unsigned long long)(W >> 20), wv, depth, loads / ms / 1e6, loads * 64 / ms / 1e9, 420e6 / (loads / ms));
            }
    return 0;
}
