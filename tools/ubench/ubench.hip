// ubench.hip -- micro-benchmarks that steer the kernel design (run on the GPU box):
//   1. LDS access cost by width and misalignment (correctness + cycles per wave-instruction)
//   2. DPP-based wave64 inclusive scan / min-reduce vs the shuffle versions
//   4. issue rates of one CU: VALU (per SIMD) and scalar ALU (per CU) with 16 resident wavefronts
//   3. cost of predicating an LDS store: exec mask vs a per-lane dump address vs an out-of-range address
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



// Issue rates: 16 wavefronts (4 per SIMD) run `iters` x 32 independent instructions of one kind.
template <int KIND>  // 0: v_add_u32, 1: s_add_u32, 2: v_cndmask_b32, 3: s_and_saveexec/s_or pair
__global__ void k_issue(uint64_t *res, uint32_t *sink, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    __syncthreads();
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
                         "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                         "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 1) {
            asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                         "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                         "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                         "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                         "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                         ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (KIND == 2) {
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if (KIND == 4) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[20:21]\n v_cndmask_b32_e64 %1, %1, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %4, s[20:21]\n v_cndmask_b32_e64 %5, %5, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %6, s[20:21]\n v_cndmask_b32_e64 %7, %7, %7, s[20:21]\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]\n v_cndmask_b32_e64 %1, %1, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %4, s[20:21]\n v_cndmask_b32_e64 %5, %5, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %6, s[20:21]\n v_cndmask_b32_e64 %7, %7, %7, s[20:21]\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]\n v_cndmask_b32_e64 %1, %1, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %4, s[20:21]\n v_cndmask_b32_e64 %5, %5, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %6, s[20:21]\n v_cndmask_b32_e64 %7, %7, %7, s[20:21]\n v_cndmask_b32_e64 %0, %0, %0, s[20:21]\n v_cndmask_b32_e64 %1, %1, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %2, s[20:21]\n v_cndmask_b32_e64 %3, %3, %3, s[20:21]\n v_cndmask_b32_e64 %4, %4, %4, s[20:21]\n v_cndmask_b32_e64 %5, %5, %5, s[20:21]\n v_cndmask_b32_e64 %6, %6, %6, s[20:21]\n v_cndmask_b32_e64 %7, %7, %7, s[20:21]\n " : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21");
        } else if (KIND == 5) {
            asm volatile("v_add_u32 %0, s22, %0\n v_add_u32 %1, s22, %1\n v_add_u32 %2, s22, %2\n v_add_u32 %3, s22, %3\n v_add_u32 %4, s22, %4\n v_add_u32 %5, s22, %5\n v_add_u32 %6, s22, %6\n v_add_u32 %7, s22, %7\n v_add_u32 %0, s22, %0\n v_add_u32 %1, s22, %1\n v_add_u32 %2, s22, %2\n v_add_u32 %3, s22, %3\n v_add_u32 %4, s22, %4\n v_add_u32 %5, s22, %5\n v_add_u32 %6, s22, %6\n v_add_u32 %7, s22, %7\n v_add_u32 %0, s22, %0\n v_add_u32 %1, s22, %1\n v_add_u32 %2, s22, %2\n v_add_u32 %3, s22, %3\n v_add_u32 %4, s22, %4\n v_add_u32 %5, s22, %5\n v_add_u32 %6, s22, %6\n v_add_u32 %7, s22, %7\n v_add_u32 %0, s22, %0\n v_add_u32 %1, s22, %1\n v_add_u32 %2, s22, %2\n v_add_u32 %3, s22, %3\n v_add_u32 %4, s22, %4\n v_add_u32 %5, s22, %5\n v_add_u32 %6, s22, %6\n v_add_u32 %7, s22, %7\n " : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s22");
        } else if (KIND == 6) {
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0\n " : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if (KIND == 7) {
            asm volatile("v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n v_cndmask_b32 %4, %4, %4, vcc\n v_cndmask_b32 %5, %5, %5, vcc\n v_cndmask_b32 %6, %6, %6, vcc\n v_cndmask_b32 %7, %7, %7, vcc\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n v_cndmask_b32 %4, %4, %4, vcc\n v_cndmask_b32 %5, %5, %5, vcc\n v_cndmask_b32 %6, %6, %6, vcc\n v_cndmask_b32 %7, %7, %7, vcc\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n v_cndmask_b32 %4, %4, %4, vcc\n v_cndmask_b32 %5, %5, %5, vcc\n v_cndmask_b32 %6, %6, %6, vcc\n v_cndmask_b32 %7, %7, %7, vcc\n v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n v_cndmask_b32 %4, %4, %4, vcc\n v_cndmask_b32 %5, %5, %5, vcc\n v_cndmask_b32 %6, %6, %6, vcc\n v_cndmask_b32 %7, %7, %7, vcc\n " : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else {
            asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]\n s_and_saveexec_b64 s[22:23], vcc\n s_or_b64 exec, exec, s[22:23]\n"
                         ::: "s20", "s21", "s22", "s23", "scc");
        }
    }
    __syncthreads();
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// LDS store predication.  `nact` of 64 lanes store to scattered addresses; the others are disabled by
//   MODE 0: the exec mask (if), MODE 1: redirecting them to a per-lane dump dword,
//   MODE 2: redirecting them to an address beyond the workgroup's LDS allocation (hardware drops it).
template <int MODE>
__global__ void k_pred(uint64_t *res, uint32_t *sink, int nact, int iters)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = (lane * nact) / 64 != ((lane + 1) * nact) / 64 || (nact == 64);
    const uint32_t real = (uint32_t)(wave * 2048 + ((lane * 37) & 511) * 4 + 1);   // scattered, misaligned
    const uint32_t dump = 60000u + (uint32_t)threadIdx.x * 4u;
    const uint32_t addr = MODE == 0 ? real : (active ? real : (MODE == 1 ? dump : 0x00FF0000u));
    uint32_t acc = lane;
    __syncthreads();
    uint64_t t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {
                if (active) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(acc), "n"(0) : "memory");
            } else if (MODE == 3) {  // exec mask, dword-ALIGNED scattered addresses
                if (active) asm volatile("ds_write_b32 %0, %1" ::"v"(addr & ~3u), "v"(acc) : "memory");
            } else if (MODE == 4) {  // exec mask, atomic OR (no return) on aligned scattered words
                if (active) asm volatile("ds_or_b32 %0, %1" ::"v"(addr & ~3u), "v"(acc) : "memory");
            } else if (MODE == 5) {  // exec mask, byte stores
                if (active) asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(acc) : "memory");
            } else if (MODE == 7) {  // exec mask, byte-misaligned 8-byte stores
                if (active) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"((uint64_t)acc) : "memory");
            } else if (MODE == 8) {  // exec mask, 2-byte stores at odd addresses
                if (active) asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(acc) : "memory");
            } else if (MODE == 9) {  // exec mask, 8-byte stores at 4-byte (not 8-byte) aligned addresses
                if (active) asm volatile("ds_write_b64 %0, %1" ::"v"((addr & ~3u) | 4u), "v"((uint64_t)acc) : "memory");
            } else if (MODE == 6) {  // exec mask, aligned dword READS
                uint32_t r_;
                if (active) { asm volatile("ds_read_b32 %0, %1" : "=v"(r_) : "v"(addr & ~3u) : "memory"); }
            } else {
                asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(acc) : "memory");
            }
            acc += u;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) res[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + smem[lane];
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


    {
        auto runi = [&](auto kern, const char *name, int threads) {
            uint64_t r;
            kern<<<1, threads>>>(d_res, d_sink, 20000);
            hipDeviceSynchronize();
            hipMemcpy(&r, d_res, 8, hipMemcpyDeviceToHost);
            const double instrs = 20000.0 * 32 * (threads / 64);
            printf("issue rate %-28s %2d waves on one CU: %.2f cycles per wave-instruction (CU-wide), %.2f per SIMD\n", name,
                   threads / 64, (double)r / instrs, (double)r / instrs * 4);
        };
        for (int threads : {64, 256, 1024}) {
            runi(k_issue<0>, "v_add_u32", threads);
            runi(k_issue<2>, "v_cndmask_b32", threads);
            runi(k_issue<1>, "s_add_u32", threads);
            runi(k_issue<3>, "s_and_saveexec / s_or exec", threads);
            runi(k_issue<4>, "v_cndmask_e64 sgpr mask", threads);
            runi(k_issue<7>, "v_cndmask vcc independent", threads);
            runi(k_issue<5>, "v_add_u32 v, s, v", threads);
            runi(k_issue<6>, "v_cmp_lt_u32 vcc", threads);
        }
    }
    for (int nact : {4, 16, 64}) {
        auto runp = [&](auto kern, const char *name) {
            uint64_t r;
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            kern<<<1, 1024, 65536>>>(d_res, d_sink, nact, iters);
            hipDeviceSynchronize();
            hipMemcpy(&r, d_res, 8, hipMemcpyDeviceToHost);
            printf("store predication %-12s active lanes %2d, 16 waves: %.1f cycles per wave-instruction (per wave)\n", name, nact, (double)r / (iters * 8.0));
        };
        runp(k_pred<0>, "exec mask");
        runp(k_pred<1>, "dump slot");
        runp(k_pred<2>, "out of range");
        runp(k_pred<3>, "aligned b32");
        runp(k_pred<4>, "atomic or b32");
        runp(k_pred<5>, "write b8");
        runp(k_pred<6>, "aligned read");
        runp(k_pred<7>, "misaligned b64");
        runp(k_pred<8>, "odd b16");
        runp(k_pred<9>, "b64 at 4 mod 8");
    }
    k_scan<<<1, 64>>>(d_sink, d_res, 10000);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d_sink, 1024, hipMemcpyDeviceToHost));
    uint64_t r[2]; CHECK(hipMemcpy(r, d_res, 16, hipMemcpyDeviceToHost));
    int same = 1; for (int i = 0; i < 64; i++) same &= (h[i] == h[64 + i]);
    printf("dpp scan == shfl scan: %d ; dpp %.1f cycles, shfl %.1f cycles per scan\n", same, r[0] / 10000.0, r[1] / 10000.0);
    return 0;
}
