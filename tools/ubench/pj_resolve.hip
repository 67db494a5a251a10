// pj_resolve.hip -- the RESOLVE step of a pointer-jumping sequence executor, on the match graphs of real frames:
// what it costs the CU to turn "every output byte of a tile points at the byte it copies" into bytes, whatever else the executor
// does.  A workgroup takes one 16 KiB output tile that tools/pj_resolve.py built from the oracle's sequence trace of a BASELINE
// config-4 frame: src[p] (16 bits) = the in-tile byte that byte p copies, or p itself for a ROOT -- a literal, or a match byte whose
// source lies before the tile (its value is handed in: as if every far match had been an L2 hit that cost nothing) -- and val[p] =
// the roots' bytes.  The kernel
//   (b) halves the chains, src[p] <- src[src[p]], round after round until nothing moves (chunks of 2 KiB that did not move in a
//       round are skipped in the next: most bytes are two or three copies deep, a few are hundreds);
//   (c) gathers val[p] <- val[src[p]] and stores the tile, 16 bytes per lane.
// It leaves out the setup that produces src[] from the sequence records (k_exec_c's bitmap / run-table machinery, ~a third of that
// kernel) and every far-match load: a LOWER bound for the executor.  The tile is resolved REPS times from a pristine copy kept in
// LDS, so that the time is the CU's (LDS pipe), not memory's; the LDS-to-LDS copy is timed alone with rounds = 0 and subtracted.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libpj_resolve.so pj_resolve.hip
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int kTile = 16384, kThreads = 1024, kChunk = 2 * kThreads;  // (sixteen wavefronts: what the LDS pipe needs to run at its rate)

__global__ __launch_bounds__(kThreads) void k_pj_resolve(const uint16_t *__restrict__ src_g, const uint8_t *__restrict__ val_g, uint8_t *__restrict__ out_g,
                                                         uint32_t reps, uint32_t max_rounds, uint32_t *rounds_out)
{
    __shared__ __attribute__((aligned(16))) uint16_t src0[kTile];  // pristine
    __shared__ __attribute__((aligned(16))) uint16_t src[kTile];
    __shared__ __attribute__((aligned(16))) uint8_t val0[kTile];
    __shared__ __attribute__((aligned(16))) uint8_t val[kTile];
    __shared__ uint32_t moved[2][kTile / kChunk];  // per 2 KiB chunk: did a pointer move in this round (double-buffered by round parity)
    const int tid = threadIdx.x;
    const size_t tile = blockIdx.x;
    for (int i = tid; i < kTile / 8; i += kThreads) ((uint4 *)src0)[i] = ((const uint4 *)(src_g + tile * kTile))[i];
    for (int i = tid; i < kTile / 16; i += kThreads) ((uint4 *)val0)[i] = ((const uint4 *)(val_g + tile * kTile))[i];
    __syncthreads();
    uint32_t rounds_used = 0;
    for (uint32_t rep = 0; rep < reps; rep++) {
        for (int i = tid; i < kTile / 8; i += kThreads) ((uint4 *)src)[i] = ((const uint4 *)src0)[i];
        for (int i = tid; i < kTile / 16; i += kThreads) ((uint4 *)val)[i] = ((const uint4 *)val0)[i];
        if (tid < kTile / kChunk) { moved[0][tid] = 1u; moved[1][tid] = 0u; }
        __syncthreads();
        // (b) pointer doubling; a thread owns bytes 2 tid, 2 tid + 1 of every 2 KiB chunk (one aligned dword of src)
        uint32_t r = 0;
        for (; r < max_rounds; r++) {
            const int cur = r & 1, nxt = cur ^ 1;
            bool any = false;
            for (int c = 0; c < kTile / kChunk; c++) {
                if (!moved[cur][c]) continue;  // (uniform per workgroup)
                any = true;
                const int p = c * kChunk + 2 * tid;
                const uint32_t s2 = *(const uint32_t *)&src[p];
                const uint32_t a = s2 & 0xFFFFu, b = s2 >> 16;
                const uint32_t a2 = src[a], b2 = src[b];
                const bool mv = a2 != a || b2 != b;
                if (mv) *(uint32_t *)&src[p] = a2 | (b2 << 16);
                if (__builtin_amdgcn_ballot_w64(mv) != 0ull && (tid & 63) == 0) moved[nxt][c] = 1u;
            }
            __syncthreads();
            if (tid < kTile / kChunk) moved[cur][tid] = 0u;
            __syncthreads();
            if (!any) break;
        }
        rounds_used = r;
        // (c) the bytes: val[p] <- val[src[p]] (roots read themselves), then the tile leaves, 16 bytes per lane
        if (max_rounds) {
            for (int c = 0; c < kTile / kChunk; c++) {
                const int p = c * kChunk + 2 * tid;
                const uint32_t s2 = *(const uint32_t *)&src[p];
                const uint32_t va = val[s2 & 0xFFFFu], vb = val[s2 >> 16];
                *(uint16_t *)&val[p] = (uint16_t)(va | (vb << 8));
            }
            __syncthreads();
        }
        for (int i = tid; i < kTile / 16; i += kThreads) ((uint4 *)(out_g + tile * kTile))[i] = ((const uint4 *)val)[i];
        __syncthreads();
    }
    if (tid == 0) rounds_out[tile] = rounds_used;
}

extern "C" int pj_resolve_run(const uint16_t *src, const uint8_t *val, uint8_t *out, uint32_t n_tiles, uint32_t reps, uint32_t max_rounds, uint32_t *rounds,
                              float *ms)
{
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    k_pj_resolve<<<n_tiles, kThreads>>>(src, val, out, 1, max_rounds, rounds);  // warm-up
    hipEventRecord(e0);
    k_pj_resolve<<<n_tiles, kThreads>>>(src, val, out, reps, max_rounds, rounds);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    hipEventElapsedTime(ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
