// mzd_api.hip -- C-ABI implementation (include/mzd.h): context, batch residency, kernel
// launches, result download.  Host code; compiled with hipcc together with the kernels.
//
// There is NO CPU fallback: without a HIP device mzd_create fails with MZD_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mzd.h"
#include "mzd_device.h"

// unity build: the kernels live in their own file but are compiled in this translation unit
#include "mzd_kernels.hip"
#include "mzd_huf.hip"
#include "mzd_seq.hip"
#include "mzd_exec.hip"
#include "mzd_util.hip"
#include "mzd_huf_w.hip"
#include "mzd_seq_q4.hip"
#include "mzd_exec_b.hip"
#include "mzd_exec_c.hip"
#include "mzd_exec_blk.hip"
#include "mzd_parse.hip"

using namespace mzd;

struct mzd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    mzd_options opt{};
    std::string last_error;
    hipStream_t stream2 = nullptr;  // the execution kernel of the head of a split batch runs here
    hipStream_t stream3 = nullptr;  // the frames with the longest chains of a heterogeneous batch (created by the first pass that has such)
    hipEvent_t ev_head_ready = nullptr, ev_head_done = nullptr, ev_init_done = nullptr, ev_huf_done = nullptr;
    int num_cus = 256;
    // HIP events around every kernel of every mzd_batch_run since the last mzd_timing_reset
    std::vector<hipEvent_t> ev;  // kEvPerRun per run
    std::vector<uint8_t> run_split;
    size_t runs = 0;
    bool timing = true;
    bool attr_set = false;
    uint32_t test_fixup_bail = 0;  // mzd_debug_force_fixup_bail: workgroup 1 of every frame gives up at this step of the fix-up walk
    uint64_t test_large_frame = 0; // mzd_debug_plan_unit_bytes: frames of this many bytes and more are planned block by block (0: the default)
};

// temporaries of the device-side planning pass (kept by a streaming slot, freed at once otherwise)
struct ParseTemps {
    uint64_t *d_foff = nullptr, *d_flen = nullptr;
    mzd::ParseScratch *d_scratch = nullptr;
    mzd::FrameCount *d_counts = nullptr;
    mzd::FrameBase *d_bases = nullptr;
    mzd::FseBuildDesc *d_fse_tabs = nullptr;
    uint32_t *d_fse_src = nullptr;
    mzd::HufBuildDesc *d_huf_tabs = nullptr;
    uint16_t *d_huf_src = nullptr;
    uint32_t *d_keys = nullptr, *d_perm = nullptr;  // ordering of heterogeneous work lists: keys out, permutation in
    uint8_t *d_sorted = nullptr;                    // the list being gathered
    mzd::ParseUnit *d_units = nullptr;              // what the lanes of k_parse walk: whole frames, or the blocks of a large frame
    uint64_t *d_starts = nullptr;                   // k_parse_index: block starts of the large frames
    uint32_t *d_capoff = nullptr, *d_nfound = nullptr;
    size_t cap_units = 0, cap_starts = 0, cap_capoff = 0, cap_nfound = 0;
    size_t cap_foff = 0, cap_flen = 0, cap_scratch = 0, cap_counts = 0, cap_bases = 0, cap_fse_tabs = 0, cap_fse_src = 0,
           cap_huf_tabs = 0, cap_huf_src = 0, cap_keys = 0, cap_perm = 0, cap_sorted = 0;
};
static void free_parse_temps(ParseTemps &t)
{
    (void)hipFree(t.d_foff);
    (void)hipFree(t.d_flen);
    (void)hipFree(t.d_scratch);
    (void)hipFree(t.d_counts);
    (void)hipFree(t.d_bases);
    (void)hipFree(t.d_fse_tabs);
    (void)hipFree(t.d_fse_src);
    (void)hipFree(t.d_huf_tabs);
    (void)hipFree(t.d_huf_src);
    (void)hipFree(t.d_keys);
    (void)hipFree(t.d_perm);
    (void)hipFree(t.d_sorted);
    (void)hipFree(t.d_units);
    (void)hipFree(t.d_starts);
    (void)hipFree(t.d_capoff);
    (void)hipFree(t.d_nfound);
    t = ParseTemps();
}
struct DevCaps {  // bytes allocated behind the pointers of a recycled batch (0 = exact / unknown)
    size_t frame_order = 0;
    size_t in = 0, out = 0, frames = 0, blocks = 0, sums = 0, huf_tasks = 0, seq_tasks = 0, fse = 0, huf = 0, recs = 0, tiles = 0,
           lit = 0, status = 0, out_len = 0;
};

struct mzd_dbatch {
    ParseTemps tmp;
    DevCaps cap;
    // device memory
    uint8_t *d_in_alloc = nullptr;  // owned input allocation (with padding) or null when adopted
    const uint8_t *d_in = nullptr;
    uint64_t in_size = 0;
    uint8_t *d_out = nullptr;
    bool own_out = false;
    bool has_chunks = false;          // a frame description of the batch is a CHUNK of a frame (MZD_FRAME_CONTINUES)
    int32_t *d_frame_hist = nullptr;  // ... then: the offset history behind every frame's last block, three per frame
    DFrame *d_frames = nullptr;
    DBlock *d_blocks = nullptr;
    BlockSum *d_sums = nullptr;
    HufTask *d_huf_tasks = nullptr;
    SeqTask *d_seq_tasks = nullptr;
    uint32_t *d_fse_entries = nullptr;
    uint32_t n_fse_entries = 0, n_fse_built = 0;  // device cells; tables built from counts
    std::vector<uint32_t> fse_dev_off;             // first device cell of every table (+ total)
    std::vector<uint32_t> huf_dev_off;
    uint32_t n_huf_built = 0;
    float fse_build_ms = 0;  // k_fse_build at upload (tables that came as normalised counts)
    uint16_t *d_huf_entries = nullptr;
    uint64_t *d_recs = nullptr;
    uint32_t last_pass = 0;       // MZD_PASS_* of the last mzd_batch_run
    hipStream_t run_stream = nullptr;  // ... the stream it ended on, and whether anybody has waited for it since
    bool run_pending = false;
    bool trimmed = false;         // mzd_batch_trim: only the output, the statuses and the layout are left
    TileBase *d_tiles = nullptr;
    uint8_t *d_litbuf = nullptr;
    int32_t *d_status = nullptr;
    uint64_t *d_out_len = nullptr;
    // geometry
    uint32_t n_frames = 0, n_blocks = 0, n_huf_tasks = 0, n_seq_tasks = 0;
    uint32_t huf_slot_cells = 2;
    uint32_t seq_cells[3] = {512, 512, 256};  // largest LL / ML / OF table of the batch's sequence tasks (cells)
    std::vector<uint32_t> frame_seq_task;  // host: index of the first SeqTask of every frame (+ total)
    std::vector<uint64_t> frame_out_off, frame_out_cap;  // host: output slab of every frame
    uint32_t n_multi = 0;  // frames of more than one block: the ones block mode's fix-up walk has anything to do for
    uint32_t *d_walk = nullptr;  // block mode: [0] their number as k_blk_scan found it, [1 ...] the frames
    size_t cap_walk = 0;
    double max_frame_serial_ms = 0;  // the longest frame as ONE wavefront's job, from its blocks' sequence counts (0: not known -- planned on the device)
    std::vector<uint64_t> frame_in_lo, frame_in_hi;      // host: extent of the frame's sequence bitstreams in the blob (lo > hi: none)
    float parse_ms = 0;  // k_parse<0> + k_parse<1> (device-side planning only)
    // heterogeneous batches (real data: blocks of 10 and of 40 000 sequences side by side): work lists ordered by size, so
    // that the units a workgroup / wavefront holds at a time are of similar length and the long ones start first
    bool seq_sorted = false;           // d_seq_tasks is in descending n_seq order (not in frame order: no head / tail split)
    // ... in up to three launches: [0, seq_class_end[0]) chains of >= 2048 sequences, then >= 256, then the rest, each with the
    // LDS slot of ITS largest tables (short chains come with small tables: less to stage, two workgroups per CU)
    uint32_t seq_class_end[3] = {0, 0, 0};
    uint32_t seq_class_cells[3][3] = {{512, 512, 256}, {512, 512, 256}, {512, 512, 256}};
    bool huf_sorted = false;           // d_huf_tasks' quads are grouped by table size class, longest streams first
    uint32_t huf_class_end[3] = {0, 0, 0};  // quads of class 0 (tables <= 32 cells), 1 (<= 256), 2 end here
    uint32_t huf_class_long[3] = {0, 0, 0};  // ... of which the first so many have a stream of kHufLongStream bytes or more
    uint32_t *d_frame_order = nullptr;  // execution order of the frames (largest first), or null
    // a heterogeneous batch whose sequence stage is as long as its longest chain: the first long_frames frames of d_frame_order hold the
    // long chains, and their chains are the first long_tasks of d_seq_tasks (0: no such grouping)
    uint32_t long_frames = 0, long_tasks = 0;
    // block mode of the execution stage (few large frames; mzd_exec_blk.hip): allocated by the first run that takes it
    BJob *d_jobs = nullptr;
    uint32_t *d_heads = nullptr;  // [0] number of jobs, [1 + j] first block of job j
    BFrame *d_bframes = nullptr;
    uint32_t *d_fixdone = nullptr;  // block mode: steps each fix-up workgroup of a frame finished (64 per frame)
    size_t cap_fixdone = 0;
    uint8_t *d_planes = nullptr;  // (passes - 1) copies of the output layout
    uint8_t *d_pat = nullptr;     // the passes' patterns, by frame-relative position
    size_t cap_jobs = 0, cap_heads = 0, cap_bframes = 0, cap_planes = 0, cap_pat = 0;
    uint32_t pat_n = 0, pat_np = 0;  // what d_pat holds
    uint64_t out_size = 0;
    uint64_t n_recs = 0, n_tiles = 0, lit_bytes = 0;  // extent of the scratch arrays (mzd_batch_debug_read)
    uint64_t huf_out_bytes = 0;                        // literals the Huffman stage regenerates (scratch or in place)
    mzd_batch_stats stats{};
};

namespace {

constexpr size_t kEvPerRun = 14;  // ([12], [13]: the third stream's execution launch of a pass in two groups of frames)
thread_local std::string g_create_error;

#define HIP_TRY(ctx, expr)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char buf_[256];                                                                   \
            snprintf(buf_, sizeof buf_, "%s failed: %s", #expr, hipGetErrorString(e_));       \
            (ctx)->last_error = buf_;                                                         \
            return MZD_ERR_DEVICE;                                                            \
        }                                                                                     \
    } while (0)

template <class T>
int upload_vec(mzd_ctx *ctx, const std::vector<T> &v, T **dptr)
{
    *dptr = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc((void **)dptr, bytes));
    if (!v.empty()) HIP_TRY(ctx, hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MZD_OK;
}

// Experiment hooks (residency caps, chains per workgroup, entropy-stages-only runs) exist only in builds made with
// -DMZD_EXPERIMENTS (tools/experiments); the release library reads no environment variable anywhere.
#ifdef MZD_EXPERIMENTS
inline const char *exp_env(const char *name) { return getenv(name); }
#else
constexpr const char *exp_env(const char *) { return nullptr; }
#endif


// ---- heterogeneous work lists are ordered by size (see mzd_dbatch): chains of one workgroup run until the longest is done,
// the 64 Huffman streams of a wavefront until the longest is done, a frame is one serial job of the execution stage.
// A lane per stream lasts as long as its stream: the quads of a class whose longest stream regenerates this many bytes or more take
// k_huf_seg (a wavefront per stream, its segments in parallel) -- real data's Huffman stage, alone: 3.05 -> 1.12 ms at 1 GiB
constexpr uint32_t kHufLongStream = 4096;
struct ListOrder {
    std::vector<uint32_t> seq_perm, huf_perm, frame_order;  // empty: leave the list in frame order
    uint32_t huf_class_end[3] = {0, 0, 0}, huf_class_long[3] = {0, 0, 0};
    uint32_t seq_class_end[3] = {0, 0, 0}, seq_class_logs[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // LL, ML, OF accuracy logs per class
    uint32_t long_frames = 0, long_tasks = 0;  // see mzd_dbatch
};
inline uint32_t seq_key(uint32_t n_seq, uint32_t ll, uint32_t ml, uint32_t of) { return std::min(n_seq, 0xFFFFFu) | (ll << 20) | (ml << 24) | (of << 28); }
// seq_nseq[i]: seq_key() of chain i; huf_key[q]: MaxBits << 24 | longest stream of quad q (capped); frame_cap[f]: output bound
void plan_order(const uint32_t *seq_nseq, size_t ns, const uint32_t *huf_key, size_t nq, const uint64_t *frame_cap, size_t nf,
                uint64_t in_size, ListOrder &o)
{
    uint64_t sum = 0;
    uint32_t mx = 0;
    for (size_t i = 0; i < ns; i++) { sum += seq_nseq[i] & 0xFFFFFu; mx = std::max(mx, seq_nseq[i] & 0xFFFFFu); }
    // (the sorted list is decoded in launches that address the bitstreams with 32-bit offsets from the blob's start)
    if (ns > 64 && (uint64_t)mx * ns >= 2 * sum && in_size < (1ull << 32) - 2 * MZD_IN_PAD - 4096) {
        o.seq_perm.resize(ns);
        for (size_t i = 0; i < ns; i++) o.seq_perm[i] = (uint32_t)i;
        std::stable_sort(o.seq_perm.begin(), o.seq_perm.end(),
                         [&](uint32_t x, uint32_t y) { return (seq_nseq[x] & 0xFFFFFu) > (seq_nseq[y] & 0xFFFFFu); });
        for (size_t i = 0; i < ns; i++) {
            const uint32_t k = seq_nseq[o.seq_perm[i]], n = k & 0xFFFFFu;
            const int c = n >= 2048 ? 0 : (n >= 256 ? 1 : 2);
            o.seq_class_end[c] = (uint32_t)i + 1;
            o.seq_class_logs[c][0] = std::max(o.seq_class_logs[c][0], (k >> 20) & 15u);
            o.seq_class_logs[c][1] = std::max(o.seq_class_logs[c][1], (k >> 24) & 15u);
            o.seq_class_logs[c][2] = std::max(o.seq_class_logs[c][2], (k >> 28) & 15u);
        }
        for (int c = 1; c < 3; c++) o.seq_class_end[c] = std::max(o.seq_class_end[c], o.seq_class_end[c - 1]);
    }
    uint64_t hsum = 0;
    uint32_t hmx = 0, bits_lo = 99, bits_hi = 0;
    std::vector<uint64_t> key(nq);  // class << 40 | (0xFFFFFF - longest stream of the quad): ascending
    for (size_t q = 0; q < nq; q++) {
        const uint32_t longest = huf_key[q] & 0xFFFFFFu, mb = huf_key[q] >> 24;
        hsum += longest;
        hmx = std::max(hmx, longest);
        bits_lo = std::min(bits_lo, mb);
        bits_hi = std::max(bits_hi, mb);
        const uint64_t cls = mb <= 5 ? 0 : (mb <= 8 ? 1 : 2);
        key[q] = (cls << 40) | (uint64_t)(0xFFFFFFu - longest);
    }
    if (nq > 64 && ((uint64_t)hmx * nq >= 2 * hsum || (bits_lo <= 8 && bits_hi > 8) || (bits_lo <= 5 && bits_hi > 5))) {
        o.huf_perm.resize(nq);
        for (size_t q = 0; q < nq; q++) o.huf_perm[q] = (uint32_t)q;
        std::stable_sort(o.huf_perm.begin(), o.huf_perm.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
        for (size_t q = 0; q < nq; q++) {
            const uint64_t kq = key[o.huf_perm[q]];
            o.huf_class_end[kq >> 40] = (uint32_t)q + 1;
            if (0xFFFFFFu - (uint32_t)(kq & 0xFFFFFFu) >= kHufLongStream) o.huf_class_long[kq >> 40]++;  // (the first ones of the class: longest first)
        }
        for (int c = 1; c < 3; c++) o.huf_class_end[c] = std::max(o.huf_class_end[c], o.huf_class_end[c - 1]);
    }
    uint64_t csum = 0, cmx = 0;
    for (size_t f = 0; f < nf; f++) { csum += frame_cap[f]; cmx = std::max(cmx, frame_cap[f]); }
    if (nf > 64 && cmx * nf >= 2 * csum) {
        o.frame_order.resize(nf);
        for (size_t f = 0; f < nf; f++) o.frame_order[f] = (uint32_t)f;
        std::stable_sort(o.frame_order.begin(), o.frame_order.end(), [&](uint32_t x, uint32_t y) { return frame_cap[x] > frame_cap[y]; });
    }
}
// The frames that hold the longest chains, apart (real data: one chain of 42 k sequences is 4.9 ms of a sequence stage whose work is
// 0.8 ms at 1 GiB -- and the execution of every OTHER frame, 4.9 ms of its own, waited for it).  When the longest chain is more than 2.5
// rounds of the chip's work: the chains of at least half its length (and 2 048 sequences) are the long ones, the frames that hold any of
// them go first in the frame order, their chains first in the task list, each part in its old order; mzd_batch_run decodes and executes
// the two groups on two streams.  ks: seq_key() of every chain, in frame order; frame_seq_task[f]: the first chain of frame f (+ total).
void group_long_frames(const uint32_t *ks, size_t ns, const std::vector<uint32_t> &frame_seq_task, uint32_t n_frames, int num_cus, ListOrder &order)
{
    if (order.seq_perm.empty() || order.frame_order.empty() || ns == 0) return;
    uint64_t sum = 0;
    for (size_t i = 0; i < ns; i++) sum += ks[i] & 0xFFFFFu;
    const uint32_t n_max = ks[order.seq_perm[0]] & 0xFFFFFu, lim = std::max(n_max / 2, 2048u);
    const uint64_t in_flight = (uint64_t)kQ4Chains * (uint64_t)std::max(num_cus, 1);
    size_t nl = 0;
    while (nl < ns && (ks[order.seq_perm[nl]] & 0xFFFFFu) >= lim) nl++;
    // (measured on the reference's corpus, longest chain x chains in flight / all sequences = 6.3 / 3.1 / 1.6 at 1 / 2 / 4 GiB: the pass
    // 9.86 -> 8.26, 10.84 -> 9.19, 14.14 -> 15.2 ms -- beyond 2.5 the grouping pays)
    if ((uint64_t)n_max * in_flight * 2 < 5 * sum || nl == 0 || nl > (size_t)kQ4Chains * 64) return;
    std::vector<uint8_t> is_long(n_frames, 0);
    auto frame_of = [&](uint32_t task) {
        return (uint32_t)(std::upper_bound(frame_seq_task.begin(), frame_seq_task.end(), task) - frame_seq_task.begin()) - 1u;
    };
    for (size_t i = 0; i < nl; i++) is_long[frame_of(order.seq_perm[i])] = 1;
    std::stable_partition(order.frame_order.begin(), order.frame_order.end(), [&](uint32_t f) { return is_long[f] != 0; });
    std::stable_partition(order.seq_perm.begin(), order.seq_perm.end(), [&](uint32_t t) { return is_long[frame_of(t)] != 0; });
    for (uint32_t f = 0; f < n_frames; f++) order.long_frames += is_long[f];
    for (size_t i = 0; i < ns; i++) order.long_tasks += is_long[frame_of(order.seq_perm[i])];
    if (order.long_frames == n_frames) order.long_frames = order.long_tasks = 0;  // (nothing to run beside them)
}
inline uint32_t huf_quad_key(const HufTask *q4)
{
    uint32_t longest = 0, mb = 1;
    for (int k = 0; k < 4; k++) { longest = std::max(longest, q4[k].out_size); mb = std::max(mb, q4[k].max_bits); }
    return (mb << 24) | std::min(longest, 0xFFFFFFu);
}
// the same keys from work lists that were written on the device (k_parse), and the reordering itself
__global__ void k_task_keys(const SeqTask *st, uint32_t ns, const HufTask *ht, uint32_t nq, uint32_t *ks, uint32_t *kh)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ns) ks[i] = min(st[i].n_seq, 0xFFFFFu) | ((uint32_t)st[i].ll_log << 20) | ((uint32_t)st[i].ml_log << 24) | ((uint32_t)st[i].of_log << 28);
    if (i < nq) {
        uint32_t longest = 0, mb = 1;
        for (int k = 0; k < 4; k++) { longest = max(longest, ht[4 * (size_t)i + k].out_size); mb = max(mb, ht[4 * (size_t)i + k].max_bits); }
        kh[i] = (mb << 24) | min(longest, 0xFFFFFFu);
    }
}
template <class T>
__global__ void k_gather(const T *src, T *dst, const uint32_t *perm, uint32_t n, uint32_t group)  // dst[i][k] = src[perm[i]][k]
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * group) dst[i] = src[(size_t)perm[i / group] * group + i % group];
}

const int kMaxSym[3] = {35, 31, 52};  // LL, OF, ML (predefined.go table lengths - 1)
const int kMaxLog[3] = {9, 8, 9};

}  // namespace

extern "C" {

int mzd_abi_version(void) { return MZD_ABI_VERSION; }
#ifndef MZD_BUILD_ID
#define MZD_BUILD_ID "unstamped"
#endif
const char *mzd_build_id(void) { return MZD_BUILD_ID; }
const char *mzd_backend(void) { return "hip-gfx950"; }

const char *mzd_strerror(int code)
{
    switch (code) {
    case MZD_OK: return "ok";
    case MZD_ERR_TRUNCATED: return "unexpected end of input";
    case MZD_ERR_MAGIC: return "Magicnum is not correct";
    case MZD_ERR_BLOCK_TYPE: return "Illegal BlockType. Must be smaller than 3.";
    case MZD_ERR_BLOCK_SIZE: return "Illegal block-size. Must be lower than 128kb";
    case MZD_ERR_FSE_TABLE: return "The probabilities didnt add up to the expected total sum";
    case MZD_ERR_HUF_WEIGHTS: return "The weights didnt leave a power of two for the last weight";
    case MZD_ERR_NO_PREV_TABLE: return "No previous table available to carry over";
    case MZD_ERR_BAD_PADDING: return "The padding at the end of the stream was more than a byte";
    case MZD_ERR_HUF_BITS: return "Didnt read all bits to decode huffman stream";
    case MZD_ERR_HUF_LENGTH: return "Huffstream did not decode to the correct length";
    case MZD_ERR_SEQ_BITS: return "Did not read all bits to decode sequences";
    case MZD_ERR_CORRUPT_SIZES: return "The sizes of literal and sequence section did not add up to blocksize";
    case MZD_ERR_LITERALS: return "Not enough bytes read to execute literals copy";
    case MZD_ERR_OFFSET: return "You cant repeat bytes from before the first one";
    case MZD_ERR_DST_FULL: return "frame output does not fit its slab / content size mismatch";
    case MZD_ERR_UNSUPPORTED: return "outside the device path's limits";
    case MZD_ERR_OUT_OF_BLOCKS: return "No blocks left in frame";
    case MZD_ERR_CHECKSUM: return "Content checksum mismatch";
    case MZD_ERR_DEVICE: return "HIP runtime error";
    case MZD_ERR_INVALID_ARG: return "invalid argument";
    case MZD_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    default: return "unknown error";
    }
}

int mzd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

mzd_ctx *mzd_create(int device, const mzd_options *opt, int *err)
{
    int n = mzd_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        if (err) *err = n <= 0 ? MZD_ERR_NO_DEVICE : MZD_ERR_INVALID_ARG;
        return nullptr;
    }
    mzd_ctx *c = new mzd_ctx();
    c->device = device;
    if (opt) c->opt = *opt;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_head_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_head_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_init_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_huf_done, hipEventDisableTiming) != hipSuccess) {
        if (err) *err = MZD_ERR_DEVICE;
        delete c;
        return nullptr;
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->num_cus = prop.multiProcessorCount;
    }
    if (err) *err = MZD_OK;
    return c;
}

void mzd_destroy(mzd_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->ev_head_ready) (void)hipEventDestroy(ctx->ev_head_ready);
    if (ctx->ev_head_done) (void)hipEventDestroy(ctx->ev_head_done);
    if (ctx->ev_init_done) (void)hipEventDestroy(ctx->ev_init_done);
    if (ctx->ev_huf_done) (void)hipEventDestroy(ctx->ev_huf_done);
    delete ctx;
}

const char *mzd_last_error(mzd_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

void mzd_batch_free(mzd_ctx *ctx, mzd_dbatch *db)
{
    if (!db) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipFree(db->d_in_alloc);
    if (db->own_out) (void)hipFree(db->d_out);
    (void)hipFree(db->d_frames);
    (void)hipFree(db->d_blocks);
    (void)hipFree(db->d_sums);
    (void)hipFree(db->d_huf_tasks);
    (void)hipFree(db->d_seq_tasks);
    (void)hipFree(db->d_fse_entries);
    (void)hipFree(db->d_huf_entries);
    (void)hipFree(db->d_recs);
    (void)hipFree(db->d_tiles);
    (void)hipFree(db->d_litbuf);
    (void)hipFree(db->d_status);
    (void)hipFree(db->d_out_len);
    (void)hipFree(db->d_frame_order);
    (void)hipFree(db->d_jobs);
    (void)hipFree(db->d_heads);
    (void)hipFree(db->d_bframes);
    (void)hipFree(db->d_fixdone);
    (void)hipFree(db->d_walk);
    (void)hipFree(db->d_frame_hist);
    (void)hipFree(db->d_planes);
    (void)hipFree(db->d_pat);
    free_parse_temps(db->tmp);
    delete db;
}

int mzd_batch_upload(mzd_ctx *ctx, const mzd_batch *b, mzd_dbatch **out)
{
    if (!ctx || !b || !out) return MZD_ERR_INVALID_ARG;
    *out = nullptr;
    if (b->abi_version != MZD_ABI_VERSION) {
        ctx->last_error = "abi_version mismatch";
        return MZD_ERR_INVALID_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ---- validate tables once per table
    std::vector<uint8_t> fse_ok(b->n_fse_tables, 1), huf_ok(b->n_huf_tables, 1);
    uint32_t n_fse_build = 0;  // tables that arrive as normalised counts (built on the device below)
    // device layout of the decoding tables: every table gets its full 1 << acc_log cells, whatever form it
    // arrived in (a count-form table occupies only (n_symbols + 1) / 2 cells of the host array)
    std::vector<uint32_t> fse_dev_off(b->n_fse_tables + 1, 0);
    {
        uint64_t at = 0;
        for (uint32_t i = 0; i < b->n_fse_tables; i++) {
            fse_dev_off[i] = (uint32_t)at;
            at += 1ull << std::min<uint32_t>(b->fse_tables[i].acc_log, 9);
            if (at > 0xFFFFFFFFull) {
                ctx->last_error = "more than 2^32 FSE table cells";
                return MZD_ERR_INVALID_ARG;
            }
        }
        fse_dev_off[b->n_fse_tables] = (uint32_t)at;
    }
    for (uint32_t i = 0; i < b->n_fse_tables; i++) {
        const mzd_fse_table_desc &d = b->fse_tables[i];
        const uint64_t n = 1ull << d.acc_log;
        // cells this table occupies in fse_entries[]: all of them, or just the packed counts
        const uint64_t host_cells = (d.build & MZD_FSE_FROM_COUNTS) ? ((d.build & 0xFFu) + 1) / 2 : n;
        if (d.kind > 2 || d.acc_log > kMaxLog[d.kind] || (uint64_t)d.entries_off + host_cells > b->n_fse_entries) {
            fse_ok[i] = 0;
            continue;
        }
        if (d.build & MZD_FSE_FROM_COUNTS) {
            // normalised counts: they must add up to the table size (then the spread of fse.go:160-190
            // visits every cell exactly once) and name no symbol beyond the kind's alphabet
            const uint32_t nsym = d.build & 0xFF;
            int64_t sum = 0;
            bool ok = (d.build & ~(MZD_FSE_FROM_COUNTS | 0xFFu)) == 0 && nsym >= 1 && nsym <= (uint32_t)kMaxSym[d.kind] + 1 &&
                      d.acc_log >= 5;
            for (uint32_t sidx = 0; ok && sidx < nsym; sidx++) {
                const mzd_fse_entry &e = b->fse_entries[d.entries_off + (sidx >> 1)];
                const int16_t c = (sidx & 1) ? (int16_t)((uint16_t)e.nbits | ((uint16_t)e.symbol << 8)) : (int16_t)e.baseline;
                if (c < -1) ok = false;
                sum += c < 0 ? 1 : c;
            }
            if (!ok || sum != (int64_t)n) fse_ok[i] = 0;
            else n_fse_build++;
            continue;
        }
        for (uint64_t j = 0; j < n; j++) {
            const mzd_fse_entry &e = b->fse_entries[d.entries_off + j];
            // a decoding-table cell is canonical (fse.go:209-213): baseline + size is a multiple of
            // 2^nbits; the kernels rebuild nbits and baseline from (baseline + size) >> nbits
            if (e.symbol > kMaxSym[d.kind] || e.nbits > d.acc_log || (uint32_t)e.baseline + (1u << e.nbits) > n ||
                (((uint32_t)e.baseline + (uint32_t)n) & ((1u << e.nbits) - 1)) != 0) {
                fse_ok[i] = 0;
                break;
            }
        }
    }
    uint32_t max_huf_bits = 1;
    uint32_t seq_logs[3] = {0, 0, 0};  // largest LL / ML / OF accuracy log among the sequence tasks
    // device layout of the Huffman decode tables: 1 << max_bits cells each, whatever form they arrived in
    std::vector<uint32_t> huf_dev_off(b->n_huf_tables + 1, 0);
    std::vector<uint8_t> huf_bits(b->n_huf_tables, 1);
    {
        uint64_t at = 0;
        for (uint32_t i = 0; i < b->n_huf_tables; i++) {
            const mzd_huf_table_desc &d = b->huf_tables[i];
            const bool from_w = (d.max_bits & MZD_HUF_FROM_WEIGHTS) != 0;
            const uint32_t mb = d.max_bits & 0xFF, nw = from_w ? (d.max_bits >> 8) & 0xFFFF : 0;
            huf_dev_off[i] = (uint32_t)at;
            if (mb < 1 || mb > 11 || (d.entries_off & 1) || (!from_w && (d.max_bits >> 8)) || (from_w && (nw < 1 || nw > 255)) ||
                (uint64_t)d.entries_off + (from_w ? (nw + 1) / 2 : (1ull << mb)) > b->n_huf_entries) {
                huf_ok[i] = 0;
                at += 2;
                continue;
            }
            huf_bits[i] = (uint8_t)mb;
            at += 1ull << mb;
            if (from_w) {
                // huffman.go:112-131: the weights determine MaxBits and leave a power of two for the last symbol
                uint32_t sum = 0;
                for (uint32_t j = 0; j < nw; j++) {
                    const mzd_huf_entry &e = b->huf_entries[d.entries_off + (j >> 1)];
                    const uint32_t w = (j & 1) ? e.nbits : e.symbol;
                    if (w > 11) huf_ok[i] = 0;
                    else if (w) sum += 1u << (w - 1);
                }
                const uint32_t left = (1u << mb) - sum;
                if (!huf_ok[i] || sum == 0 || sum >= (1u << mb) || (sum >> (mb - 1)) == 0 || (left & (left - 1))) huf_ok[i] = 0;
            } else {
                const uint32_t n = 1u << mb;
                for (uint32_t j = 0; j < n; j++) {
                    const mzd_huf_entry &e = b->huf_entries[d.entries_off + j];
                    if (e.nbits < 1 || e.nbits > mb) {
                        huf_ok[i] = 0;
                        break;
                    }
                }
            }
            if (huf_ok[i]) max_huf_bits = std::max(max_huf_bits, mb);
        }
        if (at > 0xFFFFFFFFull) {
            ctx->last_error = "more than 2^32 Huffman table cells";
            return MZD_ERR_INVALID_ARG;
        }
        huf_dev_off[b->n_huf_tables] = (uint32_t)at;
    }

    // ---- derive the device work lists
    std::vector<DFrame> frames(b->n_frames);
    std::vector<DBlock> blocks(b->n_blocks);
    std::vector<HufTask> huf_tasks;
    std::vector<SeqTask> seq_tasks;
    uint64_t rec_total = 0, tile_total = 0, lit_total = 0, huf_out_total = 0;
    mzd_batch_stats st{};
    auto in_range = [&](uint64_t off, uint64_t n) { return off <= b->in_size && n <= b->in_size - off; };
    std::vector<uint32_t> frame_seq_task(b->n_frames + 1, 0);
    bool has_chunks = false;
    std::vector<uint64_t> frame_in_lo(b->n_frames, ~0ull), frame_in_hi(b->n_frames, 0);
    for (uint32_t f = 0; f < b->n_frames; f++) {
        frame_seq_task[f] = (uint32_t)seq_tasks.size();
        const mzd_frame_desc &fd = b->frames[f];
        DFrame &df = frames[f];
        df.out_offset = fd.out_offset;
        df.out_capacity = fd.out_capacity;
        df.content_size = fd.content_size;
        df.first_block = fd.first_block;
        df.n_blocks = fd.n_blocks;
        df.plan_status = MZD_OK;
        df.checksum = fd.checksum;
        df.has_checksum = (fd.flags & MZD_FRAME_HAS_CHECKSUM) ? 1 : 0;
        // a chunk of a frame (ABI 9): its slab begins with `start` bytes of the frame's window
        const bool continues = (fd.flags & MZD_FRAME_CONTINUES) != 0;
        df.continues = continues ? 1u : 0u;
        df.start = continues ? fd.start : 0;
        df.hist[0] = continues ? fd.hist[0] : 1;
        df.hist[1] = continues ? fd.hist[1] : 4;
        df.hist[2] = continues ? fd.hist[2] : 8;
        df.pad = 0;
        if (continues) {
            has_chunks = true;
            // (positions in a slab are 32-bit for the kernels that take chunks; a history of positive offsets is what every block leaves)
            if (fd.start > fd.out_capacity || fd.out_capacity >= (1ull << 32) - 65536 || fd.hist[0] <= 0 || fd.hist[1] <= 0 || fd.hist[2] <= 0) {
                df.plan_status = MZD_ERR_INVALID_ARG;
                df.n_blocks = 0;
                continue;
            }
        }
        if ((uint64_t)fd.first_block + fd.n_blocks > b->n_blocks || (fd.out_offset & 15) ||
            fd.out_offset > b->out_size || fd.out_capacity > b->out_size - fd.out_offset) {
            df.plan_status = MZD_ERR_INVALID_ARG;
            df.n_blocks = 0;
            continue;
        }
        if (const uint32_t ps = (fd.flags >> MZD_FRAME_PLAN_STATUS_SHIFT) & 0xFFu) {  // the planner gave this frame up
            df.plan_status = (int32_t)ps;
            df.n_blocks = 0;
            continue;
        }
        st.out_capacity_bytes += fd.out_capacity;
        bool seen_seq = continues;  // (a chunk's first block with sequences starts from the history it was handed, not from {1, 4, 8})
        uint64_t in_lo = ~0ull, in_hi = 0;
        // where the block's output starts in the frame, as long as every earlier block's regenerated size is known
        // without decoding (Raw / RLE blocks, compressed blocks without sequences: their output IS their literals)
        bool pos_known = true;
        uint64_t out_pos = df.start;
        for (uint32_t k = 0; k < fd.n_blocks && df.plan_status == MZD_OK; k++) {
            const uint32_t bi = fd.first_block + k;
            const mzd_block_desc &bd = b->blocks[bi];
            DBlock &d = blocks[bi];
            memset(&d, 0, sizeof d);
            d.type = bd.type;
            d.size = bd.size;
            if (bd.type == MZD_BLOCK_RAW || bd.type == MZD_BLOCK_RLE) {
                const uint64_t need = bd.type == MZD_BLOCK_RAW ? bd.size : 1;
                if (bd.size > kBlockMax || !in_range(bd.src_off, need)) { df.plan_status = MZD_ERR_TRUNCATED; break; }
                d.src_off = bd.src_off;
                st.compressed_bytes += need;
                st.n_blocks[bd.type]++;
                out_pos += bd.size;
                continue;
            }
            if (bd.type != MZD_BLOCK_COMPRESSED) { df.plan_status = MZD_ERR_BLOCK_TYPE; break; }
            st.n_blocks[2]++;
            d.lit_type = bd.lit_type;
            d.lit_regen = bd.lit_regen;
            if (bd.lit_regen > kBlockMax) { df.plan_status = MZD_ERR_CORRUPT_SIZES; break; }
            if (bd.lit_type == MZD_LIT_RAW) {
                if (!in_range(bd.lit_off, bd.lit_regen)) { df.plan_status = MZD_ERR_TRUNCATED; break; }
                d.lit_src = bd.lit_off;
                st.compressed_bytes += bd.lit_regen;
            } else if (bd.lit_type == MZD_LIT_RLE) {
                if (!in_range(bd.lit_off, 1)) { df.plan_status = MZD_ERR_TRUNCATED; break; }
                d.lit_src = bd.lit_off;
                st.compressed_bytes += 1;
            } else if (bd.lit_type == MZD_LIT_HUF) {
                if (bd.huf_table >= b->n_huf_tables) { df.plan_status = MZD_ERR_NO_PREV_TABLE; break; }
                if (!huf_ok[bd.huf_table]) { df.plan_status = MZD_ERR_HUF_WEIGHTS; break; }
                const int ns = bd.lit_streams == 4 ? 4 : 1;
                uint64_t csum = 0;
                for (int s = 0; s < ns; s++) csum += bd.lit_stream_size[s];
                if (!in_range(bd.lit_off, csum)) { df.plan_status = MZD_ERR_TRUNCATED; break; }
                const uint32_t normal = ns == 4 ? (bd.lit_regen + 3) / 4 : bd.lit_regen;  // literals.go:306-307
                if (ns == 4 && 3ull * normal > bd.lit_regen) { df.plan_status = MZD_ERR_HUF_LENGTH; break; }
                // A block without sequences regenerates exactly its literals (sequence_execution.go:55-59).  When its
                // place in the frame is known beforehand the Huffman stage writes them THERE and the execution stage has
                // nothing to copy (BASELINE configs[2]: the whole 512 MiB of output once instead of twice).
                const bool in_place = bd.n_seq == 0 && pos_known && out_pos + bd.lit_regen <= fd.out_capacity;
                const uint64_t lit_base = in_place ? fd.out_offset + out_pos : lit_total;
                d.lit_src = lit_base;
                d.pad[0] = in_place ? 1 : 0;
                uint64_t ioff = bd.lit_off;
                for (int s = 0; s < 4; s++) {
                    HufTask t{};
                    t.pad = in_place ? 1u : 0u;  // out_off is relative to the output blob (patched to the literal scratch's base below)
                    if (s < ns) {
                        t.in_off = ioff;
                        t.in_size = bd.lit_stream_size[s];
                        t.out_off = lit_base + (uint64_t)s * normal;
                        t.out_size = ns == 4 ? (s < 3 ? normal : bd.lit_regen - 3 * normal) : bd.lit_regen;
                        ioff += t.in_size;
                        st.n_huf_streams++;
                        huf_out_total += t.out_size;
                    }
                    t.table_off = huf_dev_off[bd.huf_table];
                    t.max_bits = huf_bits[bd.huf_table];
                    t.block = bi;
                    huf_tasks.push_back(t);
                }
                if (!in_place) lit_total += ((uint64_t)bd.lit_regen + 15) & ~15ull;
                st.compressed_bytes += csum;
            } else {
                df.plan_status = MZD_ERR_INVALID_ARG;
                break;
            }
            d.n_seq = bd.n_seq;
            // (a Number_of_Sequences of zero in its two-byte form still has modes, tables and a bitstream, which the reference reads --
            // padding and initial states -- and wants used up, sequences.go:126-208: what that comes to is the planner's to say,
            // mzd_block_desc.seq_status; the execution stage reports it where the sequence stage's own status would stand)
            d.pad[1] = bd.n_seq == 0 ? bd.seq_status : 0;
            if (bd.n_seq == 0) out_pos += bd.lit_regen;
            else pos_known = false;
            if (bd.n_seq > 0) {
                const uint32_t ti[3] = {bd.ll_table, bd.of_table, bd.ml_table};
                bool ok = true;
                for (int kd = 0; kd < 3; kd++)
                    ok = ok && ti[kd] < b->n_fse_tables && fse_ok[ti[kd]] && b->fse_tables[ti[kd]].kind == kd;
                if (!ok) { df.plan_status = MZD_ERR_FSE_TABLE; break; }
                if (bd.seq_size == 0 || bd.seq_size > kBlockMax || !in_range(bd.seq_off, bd.seq_size)) {
                    df.plan_status = MZD_ERR_TRUNCATED;
                    break;
                }
                SeqTask t{};
                t.in_off = bd.seq_off;
                t.in_size = bd.seq_size;
                t.n_seq = bd.n_seq;
                t.rec_off = rec_total;
                t.tile_off = (uint32_t)tile_total;
                t.block = bi;
                t.ll_off = fse_dev_off[bd.ll_table];
                t.of_off = fse_dev_off[bd.of_table];
                t.ml_off = fse_dev_off[bd.ml_table];
                t.ll_log = b->fse_tables[bd.ll_table].acc_log;
                t.of_log = b->fse_tables[bd.of_table].acc_log;
                t.ml_log = b->fse_tables[bd.ml_table].acc_log;
                seq_logs[0] = std::max<uint32_t>(seq_logs[0], t.ll_log);
                seq_logs[1] = std::max<uint32_t>(seq_logs[1], t.ml_log);
                seq_logs[2] = std::max<uint32_t>(seq_logs[2], t.of_log);
                t.hist_known = seen_seq ? 0 : 1;
                seen_seq = true;
                in_lo = std::min<uint64_t>(in_lo, bd.seq_off);
                in_hi = std::max<uint64_t>(in_hi, bd.seq_off + bd.seq_size);
                seq_tasks.push_back(t);
                d.rec_off = rec_total;
                d.tile_off = (uint32_t)tile_total;
                rec_total += bd.n_seq;
                tile_total += (bd.n_seq + 63) / 64;
                if (tile_total > 0xFFFFFFFFull) { df.plan_status = MZD_ERR_UNSUPPORTED; break; }
                st.compressed_bytes += bd.seq_size;
                st.n_sequences += bd.n_seq;
            }
        }
        if (df.plan_status != MZD_OK) df.n_blocks = 0;
        frame_in_lo[f] = in_lo;
        frame_in_hi[f] = in_hi;
    }
    frame_seq_task[b->n_frames] = (uint32_t)seq_tasks.size();
    // ---- heterogeneous work lists are ordered by size (see mzd_dbatch)
    ListOrder order;
    {
        std::vector<uint32_t> ks(seq_tasks.size()), kh(huf_tasks.size() / 4);
        std::vector<uint64_t> caps(b->n_frames);
        for (size_t i = 0; i < ks.size(); i++) ks[i] = seq_key(seq_tasks[i].n_seq, seq_tasks[i].ll_log, seq_tasks[i].ml_log, seq_tasks[i].of_log);
        for (size_t q = 0; q < kh.size(); q++) kh[q] = huf_quad_key(&huf_tasks[4 * q]);
        for (uint32_t f = 0; f < b->n_frames; f++) caps[f] = b->frames[f].out_capacity;
        plan_order(ks.data(), ks.size(), kh.data(), kh.size(), caps.data(), caps.size(), b->in_size, order);
        group_long_frames(ks.data(), ks.size(), frame_seq_task, b->n_frames, ctx->num_cus, order);
        if (!order.seq_perm.empty()) {
            std::vector<SeqTask> sorted(seq_tasks.size());
            for (size_t i = 0; i < sorted.size(); i++) sorted[i] = seq_tasks[order.seq_perm[i]];
            seq_tasks.swap(sorted);
        }
        if (!order.huf_perm.empty()) {
            std::vector<HufTask> sorted(huf_tasks.size());
            for (size_t q = 0; q < order.huf_perm.size(); q++)
                for (int k = 0; k < 4; k++) sorted[4 * q + k] = huf_tasks[4 * (size_t)order.huf_perm[q] + k];
            huf_tasks.swap(sorted);
        }
    }
    const bool seq_sorted = !order.seq_perm.empty(), huf_sorted = !order.huf_perm.empty();
    const uint32_t *huf_class_end = order.huf_class_end;
    std::vector<uint32_t> &frame_order = order.frame_order;
    st.table_bytes = (uint64_t)b->n_fse_entries * 4 + (uint64_t)b->n_huf_entries * 2;
    st.scratch_bytes = rec_total * 8 + tile_total * 8 + lit_total;

    // ---- device memory
    mzd_dbatch *db = new mzd_dbatch();
    db->n_frames = b->n_frames;
    db->n_blocks = b->n_blocks;
    db->n_huf_tasks = (uint32_t)huf_tasks.size();
    db->n_seq_tasks = (uint32_t)seq_tasks.size();
    db->has_chunks = has_chunks;
    for (int k = 0; k < 3; k++) db->seq_cells[k] = 1u << std::min<uint32_t>(seq_logs[k], k == 2 ? 8u : 9u);
    db->huf_slot_cells = 1u << max_huf_bits;
    db->seq_sorted = seq_sorted;
    db->long_frames = order.long_frames;
    db->long_tasks = order.long_tasks;
    for (int c = 0; c < 3; c++) {
        db->seq_class_end[c] = order.seq_class_end[c];
        for (int k = 0; k < 3; k++) db->seq_class_cells[c][k] = 1u << std::min<uint32_t>(order.seq_class_logs[c][k], k == 2 ? 8u : 9u);
    }
    db->huf_sorted = huf_sorted;
    for (int c = 0; c < 3; c++) {
        db->huf_class_end[c] = huf_class_end[c];
        db->huf_class_long[c] = order.huf_class_long[c];
    }
    db->frame_seq_task = std::move(frame_seq_task);
    db->frame_in_lo = std::move(frame_in_lo);
    db->frame_in_hi = std::move(frame_in_hi);
    db->frame_out_off.resize(b->n_frames);
    db->frame_out_cap.resize(b->n_frames);
    db->max_frame_serial_ms = 0;
    db->n_multi = 0;
    for (uint32_t f = 0; f < b->n_frames; f++) {
        db->n_multi += b->frames[f].n_blocks > 1 ? 1u : 0u;
        db->frame_out_off[f] = b->frames[f].out_offset;
        db->frame_out_cap[f] = b->frames[f].out_capacity;
        // what the frame costs a wavefront that walks its blocks in order: 2 us per 64 sequences (a text-like 128 KiB block of
        // 13.4 k sequences: 0.42 ms), 6 us per block, a Raw / RLE block at 8 KiB per us
        const mzd_frame_desc &fd = b->frames[f];
        double us = 0;
        for (uint32_t k = 0; k < fd.n_blocks && fd.first_block + k < b->n_blocks; k++) {
            const mzd_block_desc &bd = b->blocks[fd.first_block + k];
            us += 6.0 + (bd.type == MZD_BLOCK_COMPRESSED ? (double)bd.n_seq * (2.0 / 64.0) + (double)bd.lit_regen / 8192.0 : (double)bd.size / 8192.0);
        }
        db->max_frame_serial_ms = std::max(db->max_frame_serial_ms, us * 1e-3);
    }
    db->out_size = b->out_size;
    db->stats = st;
    int rc = MZD_OK;
    auto fail = [&](int code) {
        mzd_batch_free(ctx, db);
        return code;
    };
#define TRY_OR_FAIL(expr)                 \
    do {                                  \
        rc = (expr);                      \
        if (rc != MZD_OK) return fail(rc); \
    } while (0)
#define HIP_OR_FAIL(expr)                                                                 \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ctx->last_error = std::string(#expr " failed: ") + hipGetErrorString(e_);     \
            return fail(MZD_ERR_DEVICE);                                                  \
        }                                                                                 \
    } while (0)

    if (b->flags & MZD_BATCH_IN_ON_DEVICE) {
        db->d_in = b->in;
        db->in_size = b->in_size;
    } else {
        HIP_OR_FAIL(hipMalloc((void **)&db->d_in_alloc, b->in_size + 2 * MZD_IN_PAD));
        HIP_OR_FAIL(hipMemset(db->d_in_alloc, 0, MZD_IN_PAD));
        HIP_OR_FAIL(hipMemset(db->d_in_alloc + MZD_IN_PAD + b->in_size, 0, MZD_IN_PAD));
        if (b->in_size) HIP_OR_FAIL(hipMemcpy(db->d_in_alloc + MZD_IN_PAD, b->in, b->in_size, hipMemcpyHostToDevice));
        db->d_in = db->d_in_alloc + MZD_IN_PAD;
        db->in_size = b->in_size;
    }
    if ((b->flags & MZD_BATCH_OUT_ON_DEVICE) && b->out) {
        db->d_out = b->out;
    } else {
        HIP_OR_FAIL(hipMalloc((void **)&db->d_out, std::max<uint64_t>(b->out_size, 16)));
        db->own_out = true;
    }
    HIP_OR_FAIL(hipMalloc((void **)&db->d_litbuf, lit_total + 64));
    // (literals that go straight to their place in the output -- HufTask.pad -- keep their offset into the output blob: the
    // Huffman kernels take both bases)
    TRY_OR_FAIL(upload_vec(ctx, frames, &db->d_frames));
    TRY_OR_FAIL(upload_vec(ctx, blocks, &db->d_blocks));
    TRY_OR_FAIL(upload_vec(ctx, huf_tasks, &db->d_huf_tasks));
    TRY_OR_FAIL(upload_vec(ctx, seq_tasks, &db->d_seq_tasks));
    if (!frame_order.empty()) TRY_OR_FAIL(upload_vec(ctx, frame_order, &db->d_frame_order));
    HIP_OR_FAIL(hipMalloc((void **)&db->d_sums, std::max<size_t>(b->n_blocks, 1) * sizeof(BlockSum)));
    // ---- FSE tables: the host array (cells or packed counts) goes up as it is; k_fse_build lays the
    // decoding tables out on the device, copying the ones that came built and building the others
    // (SURVEY 8f #1).  Once per batch.
    {
        const uint32_t n_dev_cells = fse_dev_off[b->n_fse_tables];
        HIP_OR_FAIL(hipMalloc((void **)&db->d_fse_entries, std::max<size_t>(n_dev_cells, 1) * 4));
        db->n_fse_entries = n_dev_cells;
        db->fse_dev_off = fse_dev_off;
        if (b->n_fse_tables) {
            std::vector<FseBuildDesc> tabs(b->n_fse_tables);
            for (uint32_t i = 0; i < b->n_fse_tables; i++) {
                const mzd_fse_table_desc &d = b->fse_tables[i];
                tabs[i].src_off = d.entries_off;
                tabs[i].dst_off = fse_dev_off[i];
                tabs[i].acc_log = (uint8_t)std::min<uint32_t>(d.acc_log, 9);
                tabs[i].n_sym = (uint8_t)((d.build & MZD_FSE_FROM_COUNTS) ? (d.build & 0xFF) : 0);
                tabs[i].ok = fse_ok[i];
            }
            FseBuildDesc *d_tabs = nullptr;
            uint32_t *d_src = nullptr;
            HIP_OR_FAIL(hipMalloc((void **)&d_tabs, tabs.size() * sizeof(FseBuildDesc)));
            hipError_t e0 = hipMalloc((void **)&d_src, std::max<size_t>(b->n_fse_entries, 1) * 4);
            hipError_t e1 = e0 != hipSuccess ? e0 : hipMemcpy(d_tabs, tabs.data(), tabs.size() * sizeof(FseBuildDesc), hipMemcpyHostToDevice);
            if (e1 == hipSuccess && b->n_fse_entries)
                e1 = hipMemcpy(d_src, b->fse_entries, (size_t)b->n_fse_entries * 4, hipMemcpyHostToDevice);
            hipEvent_t t0, t1;
            (void)hipEventCreate(&t0);
            (void)hipEventCreate(&t1);
            (void)hipEventRecord(t0, ctx->stream);
            if (e1 == hipSuccess)
                k_fse_build<<<(b->n_fse_tables + 63) / 64, 64, 0, ctx->stream>>>(d_tabs, b->n_fse_tables, d_src, db->d_fse_entries);
            (void)hipEventRecord(t1, ctx->stream);
            hipError_t e2 = hipStreamSynchronize(ctx->stream);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, t0, t1);
            db->fse_build_ms = ms;
            db->n_fse_built = n_fse_build;
            (void)hipEventDestroy(t0);
            (void)hipEventDestroy(t1);
            (void)hipFree(d_tabs);
            (void)hipFree(d_src);
            if (e1 != hipSuccess || e2 != hipSuccess) {
                ctx->last_error = std::string("k_fse_build failed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
                return fail(MZD_ERR_DEVICE);
            }
        }
    }
    // ---- Huffman tables: same scheme (k_huf_build copies built tables, fills the ones that came as weights)
    {
        const uint32_t n_dev_cells = huf_dev_off[b->n_huf_tables];
        HIP_OR_FAIL(hipMalloc((void **)&db->d_huf_entries, std::max<size_t>(n_dev_cells, 2) * 2 + 8));
        db->huf_dev_off = huf_dev_off;
        if (b->n_huf_tables) {
            std::vector<HufBuildDesc> tabs(b->n_huf_tables);
            for (uint32_t i = 0; i < b->n_huf_tables; i++) {
                const mzd_huf_table_desc &d = b->huf_tables[i];
                tabs[i].src_off = d.entries_off;
                tabs[i].dst_off = huf_dev_off[i];
                tabs[i].max_bits = huf_bits[i];
                tabs[i].n_weights = (uint8_t)((d.max_bits & MZD_HUF_FROM_WEIGHTS) ? (d.max_bits >> 8) & 0xFF : 0);
                tabs[i].ok = huf_ok[i];
                if (tabs[i].ok && tabs[i].n_weights) db->n_huf_built++;
            }
            HufBuildDesc *d_tabs = nullptr;
            uint16_t *d_src = nullptr;
            HIP_OR_FAIL(hipMalloc((void **)&d_tabs, tabs.size() * sizeof(HufBuildDesc)));
            hipError_t e0 = hipMalloc((void **)&d_src, std::max<size_t>(b->n_huf_entries, 2) * 2);
            hipError_t e1 = e0 != hipSuccess ? e0 : hipMemcpy(d_tabs, tabs.data(), tabs.size() * sizeof(HufBuildDesc), hipMemcpyHostToDevice);
            if (e1 == hipSuccess && b->n_huf_entries)
                e1 = hipMemcpy(d_src, b->huf_entries, (size_t)b->n_huf_entries * 2, hipMemcpyHostToDevice);
            if (e1 == hipSuccess)
                k_huf_build<<<(b->n_huf_tables + 63) / 64, 64, 0, ctx->stream>>>(d_tabs, b->n_huf_tables, d_src, db->d_huf_entries);
            hipError_t e2 = hipStreamSynchronize(ctx->stream);
            (void)hipFree(d_tabs);
            (void)hipFree(d_src);
            if (e1 != hipSuccess || e2 != hipSuccess) {
                ctx->last_error = std::string("k_huf_build failed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
                return fail(MZD_ERR_DEVICE);
            }
        }
    }
    HIP_OR_FAIL(hipMalloc((void **)&db->d_recs, std::max<uint64_t>(rec_total, 1) * 8));
    HIP_OR_FAIL(hipMalloc((void **)&db->d_tiles, std::max<uint64_t>(tile_total, 1) * sizeof(TileBase)));
    db->n_recs = rec_total;
    db->n_tiles = tile_total;
    db->lit_bytes = lit_total;
    db->huf_out_bytes = huf_out_total;
    HIP_OR_FAIL(hipMalloc((void **)&db->d_status, std::max<size_t>(b->n_frames, 1) * sizeof(int32_t)));
    HIP_OR_FAIL(hipMalloc((void **)&db->d_out_len, std::max<size_t>(b->n_frames, 1) * sizeof(uint64_t)));
    HIP_OR_FAIL(hipMemset(db->d_status, 0xFF, std::max<size_t>(b->n_frames, 1) * sizeof(int32_t)));
    if (db->has_chunks) {
        // (a batch without sequences launches nothing that writes it: it holds the frames' own history from the start)
        std::vector<int32_t> h(3 * (size_t)b->n_frames);
        for (uint32_t f = 0; f < b->n_frames; f++)
            for (int k = 0; k < 3; k++) h[3 * (size_t)f + k] = frames[f].hist[k];
        TRY_OR_FAIL(upload_vec(ctx, h, &db->d_frame_hist));
    }
    HIP_OR_FAIL(hipMemset(db->d_out_len, 0, std::max<size_t>(b->n_frames, 1) * sizeof(uint64_t)));
#undef TRY_OR_FAIL
#undef HIP_OR_FAIL
    *out = db;
    return MZD_OK;
}

// ---- planning on the device: see mzd_parse.hip.  `reuse` (streaming): a batch whose device
// buffers are recycled -- they only grow -- so that a steady stream of batches allocates nothing; all
// device work goes to `s`, and the host only waits for `s` (other streams keep decoding).
static int upload_frames_impl(mzd_ctx *ctx, const uint8_t *in, uint64_t in_size, uint32_t flags, const uint64_t *frame_off,
                              const uint64_t *frame_len, uint32_t n_frames, uint8_t *out_dev, uint64_t out_dev_size,
                              mzd_dbatch *reuse, hipStream_t s, mzd_dbatch **out)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mzd_dbatch *db = reuse ? reuse : new mzd_dbatch();
    ParseTemps &tp = db->tmp;
    hipEvent_t t0 = nullptr, t1 = nullptr, t2 = nullptr, t3 = nullptr, t4 = nullptr;
    auto cleanup = [&]() {
        for (hipEvent_t e : {t0, t1, t2, t3, t4})
            if (e) (void)hipEventDestroy(e);
        if (!reuse) free_parse_temps(tp);
    };
    auto fail = [&](int code) {
        cleanup();
        if (!reuse) mzd_batch_free(ctx, db);
        return code;
    };
#define HIP_OR_FAIL(expr)                                                             \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            ctx->last_error = std::string(#expr " failed: ") + hipGetErrorString(e_); \
            return fail(MZD_ERR_DEVICE);                                              \
        }                                                                             \
    } while (0)
    // grow-only device buffer: (pointer, capacity in bytes) pairs live in the batch
#define ENSURE(ptr, cap, bytes)                                                        \
    do {                                                                               \
        const size_t need_ = (size_t)(bytes);                                          \
        if ((cap) < need_ || !(ptr)) {                                                 \
            if (ptr) (void)hipFree((void *)(ptr));                                     \
            (ptr) = nullptr;                                                           \
            (cap) = 0;                                                                 \
            const size_t get_ = reuse ? need_ + need_ / 4 + 256 : need_;               \
            HIP_OR_FAIL(hipMalloc((void **)&(ptr), get_));                             \
            (cap) = get_;                                                              \
        }                                                                              \
    } while (0)
    for (hipEvent_t *e : {&t0, &t1, &t2, &t3, &t4}) HIP_OR_FAIL(hipEventCreate(e));
    // ---- the compressed frames
    if (flags & MZD_BATCH_IN_ON_DEVICE) {
        db->d_in = in;
    } else {
        ENSURE(db->d_in_alloc, db->cap.in, in_size + 2 * MZD_IN_PAD);
        HIP_OR_FAIL(hipMemsetAsync(db->d_in_alloc, 0, MZD_IN_PAD, s));
        HIP_OR_FAIL(hipMemsetAsync(db->d_in_alloc + MZD_IN_PAD + in_size, 0, MZD_IN_PAD, s));
        if (in_size) HIP_OR_FAIL(hipMemcpyAsync(db->d_in_alloc + MZD_IN_PAD, in, in_size, hipMemcpyHostToDevice, s));
        db->d_in = db->d_in_alloc + MZD_IN_PAD;
    }
    db->in_size = in_size;
    db->n_frames = n_frames;
    const size_t nf1 = std::max<size_t>(n_frames, 1);
    // ---- the units of the walk (mzd_parse.hip): a frame, or -- a LARGE frame -- each of its blocks.  A lane per frame is right for
    // batches of many frames (65 536 frames in 1 ms) and hopeless for one large frame (0.29 ms per block: 599 ms for 256 MiB):
    // k_parse_index walks the block headers of the frames of kLargeFrame bytes and more (one lane each, the serial part: a
    // dependent load per block), their blocks are then parsed side by side.
    const uint64_t kLargeFrame = ctx->test_large_frame ? ctx->test_large_frame : 1ull << 20;
    std::vector<ParseUnit> units;
    units.reserve(n_frames);
    std::vector<uint32_t> unit0(n_frames + 1, 0);  // first unit of every frame
    {
        std::vector<uint32_t> large;
        for (uint32_t f = 0; f < n_frames; f++)
            if (frame_len[f] >= kLargeFrame && frame_off[f] <= in_size && frame_len[f] <= in_size - frame_off[f]) large.push_back(f);
        std::vector<uint32_t> nfound;
        std::vector<uint64_t> starts;
        std::vector<uint32_t> capoff(large.size() + 1, 0);
        if (!large.empty()) {
            // (room for a block start per KiB of frame -- 0.8 % of the compressed bytes as transient HBM, 8 MiB for a 1 GiB frame; a slot
            // per 64 bytes was 12.5 %, ADVICE r5 -- and a frame with more blocks than that stays one lane's, as does a frame that
            // would take the list beyond 2^28 slots: never a saturated total that the allocation below multiplies by eight.  The
            // test hook's tiny frames, whose blocks are a few bytes each, keep the slot per 64 bytes.)
            const uint64_t slot_bytes = ctx->test_large_frame ? 64 : 1024;
            std::vector<uint64_t> lo, ll;
            uint64_t total = 0;
            size_t kept = 0;
            for (size_t j = 0; j < large.size(); j++) {
                const uint64_t slots = frame_len[large[j]] / slot_bytes + 16;
                if (total + slots > (1ull << 28)) continue;  // (stays one lane's)
                large[kept++] = large[j];
                lo.push_back(frame_off[large[j]]);
                ll.push_back(frame_len[large[j]]);
                total += slots;
                capoff[kept] = (uint32_t)total;
            }
            large.resize(kept);
            capoff.resize(kept + 1);
            if (kept) {
                ENSURE(tp.d_foff, tp.cap_foff, large.size() * 8);
                ENSURE(tp.d_flen, tp.cap_flen, large.size() * 8);
                ENSURE(tp.d_capoff, tp.cap_capoff, capoff.size() * 4);
                ENSURE(tp.d_nfound, tp.cap_nfound, large.size() * 4);
                ENSURE(tp.d_starts, tp.cap_starts, (size_t)total * 8);
                HIP_OR_FAIL(hipMemcpyAsync(tp.d_foff, lo.data(), lo.size() * 8, hipMemcpyHostToDevice, s));
                HIP_OR_FAIL(hipMemcpyAsync(tp.d_flen, ll.data(), ll.size() * 8, hipMemcpyHostToDevice, s));
                HIP_OR_FAIL(hipMemcpyAsync(tp.d_capoff, capoff.data(), capoff.size() * 4, hipMemcpyHostToDevice, s));
                k_parse_index<<<(uint32_t)((large.size() + 63) / 64), 64, 0, s>>>(db->d_in, in_size, tp.d_foff, tp.d_flen, tp.d_capoff, (uint32_t)large.size(),
                                                                                 tp.d_starts, tp.d_nfound);
                nfound.resize(large.size());
                HIP_OR_FAIL(hipMemcpyAsync(nfound.data(), tp.d_nfound, large.size() * 4, hipMemcpyDeviceToHost, s));
                HIP_OR_FAIL(hipStreamSynchronize(s));
                uint64_t used = 0;
                for (size_t j = 0; j < large.size(); j++)
                    if (nfound[j] != 0xFFFFFFFFu) used = std::max<uint64_t>(used, (uint64_t)capoff[j] + nfound[j]);
                starts.resize(used);
                if (used) HIP_OR_FAIL(hipMemcpyAsync(starts.data(), tp.d_starts, used * 8, hipMemcpyDeviceToHost, s));
                HIP_OR_FAIL(hipStreamSynchronize(s));
            }
        }
        size_t j = 0;
        for (uint32_t f = 0; f < n_frames; f++) {
            unit0[f] = (uint32_t)units.size();
            const uint64_t b = frame_off[f], e = frame_off[f] + frame_len[f];  // (out of the blob: k_parse reports it as truncated)
            const bool is_large = j < large.size() && large[j] == f;
            const uint32_t nb = is_large ? nfound[j] : 0u;
            if (!is_large || nb == 0 || nb == 0xFFFFFFFFu) {
                units.push_back(ParseUnit{b, frame_len[f] > ~0ull - b ? ~0ull : e, f, kUnitFirst | kUnitFinal, 0xFFFFFFFFu, 0});
            } else {
                units.push_back(ParseUnit{b, e, f, kUnitFirst, 1u, 0});
                for (uint32_t k = 0; k < nb; k++) units.push_back(ParseUnit{starts[capoff[j] + k], e, f, 0u, 1u, 0});
                units.back().flags |= kUnitFinal;  // (it runs to the frame's end: a frame that stops short of a last block is its to report)
                units.back().max_blocks = 0xFFFFFFFFu;
            }
            if (is_large) j++;
        }
        unit0[n_frames] = (uint32_t)units.size();
    }
    const uint32_t n_units = (uint32_t)units.size();
    const size_t nu1 = std::max<size_t>(n_units, 1);
    ENSURE(tp.d_units, tp.cap_units, nu1 * sizeof(ParseUnit));
    if (n_units) HIP_OR_FAIL(hipMemcpyAsync(tp.d_units, units.data(), (size_t)n_units * sizeof(ParseUnit), hipMemcpyHostToDevice, s));
    // one lane per unit, grid-stride; every lane owns a ParseScratch
    const uint32_t max_wg = (uint32_t)std::max(ctx->num_cus, 1) * 2;
    const uint32_t n_wg = std::max<uint32_t>(1, std::min<uint32_t>((n_units + 63) / 64, max_wg));
    ENSURE(tp.d_scratch, tp.cap_scratch, (size_t)n_wg * 64 * sizeof(ParseScratch));
    ENSURE(tp.d_counts, tp.cap_counts, nu1 * sizeof(FrameCount));
    ENSURE(tp.d_bases, tp.cap_bases, nu1 * sizeof(FrameBase));
    ParseOut po{};
    // ---- pass 0: what does every unit need?
    HIP_OR_FAIL(hipEventRecord(t0, s));
    if (n_units)
        k_parse<0><<<n_wg, 64, 0, s>>>(db->d_in, in_size, tp.d_units, n_units, tp.d_scratch, tp.d_counts, tp.d_bases, po);
    HIP_OR_FAIL(hipEventRecord(t1, s));
    std::vector<FrameCount> counts(n_units);
    if (n_units)
        HIP_OR_FAIL(hipMemcpyAsync(counts.data(), tp.d_counts, (size_t)n_units * sizeof(FrameCount), hipMemcpyDeviceToHost, s));
    HIP_OR_FAIL(hipStreamSynchronize(s));
    // ---- the status of every frame: its first unit (in block order) that failed, or that needs a table no unit before it left
    // (literals.go:247-252, sequences.go:275-366 with nothing to repeat), or a frame whose blocks end without a last one
    std::vector<int32_t> frame_status(n_frames, MZD_OK);
    for (uint32_t f = 0; f < n_frames; f++) {
        uint8_t have[4] = {0, 0, 0, 0};
        for (uint32_t u = unit0[f]; u < unit0[f + 1] && frame_status[f] == MZD_OK; u++) {
            const FrameCount &c = counts[u];
            for (int k = 0; k < 4; k++)
                if (((c.need >> k) & 1) && !have[k]) frame_status[f] = MZD_ERR_NO_PREV_TABLE;
            if (frame_status[f] == MZD_OK && c.status != MZD_OK) frame_status[f] = c.status;
            for (int k = 0; k < 4; k++) have[k] |= c.carry[k].src != 0;
        }
        if (frame_status[f] != MZD_OK)  // (as a frame that fails in one lane: it contributes nothing)
            for (uint32_t u = unit0[f]; u < unit0[f + 1]; u++) {
                FrameCount z{};
                z.status = frame_status[f];
                z.content_size = MZD_UNKNOWN_SIZE;
                counts[u] = z;
            }
    }
    // ---- offsets (exclusive prefix sums).  Tables 0..2 are the predefined ones (predefined.go), kept as
    // their normalised counts and built by k_fse_build like every other table.
    static const int16_t kDef[3][53] = {
        {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1},
        {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1},
        {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1,  1,  1,  1,  1,  1,  1,  1,  1,  1, 1,
         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1}};
    static const int kDefN[3] = {36, 29, 53}, kDefLog[3] = {6, 5, 6};  // by MZD_FSE_*: LL, OF, ML
    std::vector<FrameBase> bases(n_units);
    std::vector<uint32_t> frame_seq_task(n_frames + 1, 0);
    std::vector<DFrame> host_frames;       // the DFrames of the frames of several units (written here, not by a lane)
    std::vector<uint32_t> host_frame_idx;
    db->frame_out_off.assign(n_frames, 0);
    db->frame_out_cap.assign(n_frames, 0);
    db->max_frame_serial_ms = 0;
    db->n_multi = 0;
    db->frame_in_lo.assign(n_frames, ~0ull);
    db->frame_in_hi.assign(n_frames, 0);
    uint64_t n_blocks = 0, n_seq = 0, n_hufb = 0, n_fse_tab = 3, n_fse_src = 0, n_fse_dev = 0, n_huf_tab = 0, n_huf_src = 0, n_huf_dev = 0,
             n_tile = 0, n_rec = 0, lit_total = 0, out_at = 0;
    uint32_t predef_src[3];
    for (int k = 0; k < 3; k++) {
        po.predef_off[k] = (uint32_t)n_fse_dev;
        predef_src[k] = (uint32_t)n_fse_src;
        n_fse_dev += 1u << kDefLog[k];
        n_fse_src += (uint32_t)(kDefN[k] + 1) / 2;
    }
    uint32_t max_huf_bits = 1;
    uint32_t seq_logs[3] = {0, 0, 0};  // largest LL / ML / OF accuracy log among the sequence tasks
    mzd_batch_stats st{};
    for (uint32_t f = 0; f < n_frames; f++) {
        const uint32_t u0 = unit0[f], u1 = unit0[f + 1];
        // the frame: the sums of its units
        uint64_t f_bound = 0, f_blocks = 0, f_seq = 0, f_rec = 0, f_lit = 0;
        for (uint32_t u = u0; u < u1; u++) {
            f_bound += counts[u].out_bound;
            f_blocks += counts[u].n_blocks;
            f_seq += counts[u].n_seq;
            f_rec += counts[u].n_rec;
            f_lit += counts[u].lit_bytes;
        }
        const uint64_t f_content = counts[u0].content_size;
        // what the frame costs a wavefront that walks its blocks in order, as mzd_batch_upload reckons it from the block descriptors: 6 us
        // per block, 2 us per 64 sequences, its bytes at 8 KiB per us (the content size when the header has one, else the Huffman
        // literals: the output BOUND of a frame without one is 128 KiB per block, and real data has frames of hundreds of 1 KiB blocks --
        // by the bound alone 1 GiB of the reference's corpus took block mode, 16.5 ms instead of 8.3)
        if (frame_status[f] == MZD_OK)
            db->max_frame_serial_ms = std::max(db->max_frame_serial_ms,
                                               1e-3 * (6.0 * (double)f_blocks + (double)f_rec * (2.0 / 64.0) +
                                                       (double)(f_content != MZD_UNKNOWN_SIZE ? f_content : f_lit) / 8192.0));
        if (u1 - u0 > 1 && f_content != MZD_UNKNOWN_SIZE) f_bound = std::min(f_bound, f_content);  // (a frame in one unit: its lane did)
        const uint64_t f_cap = frame_status[f] == MZD_OK ? f_bound : 0;
        frame_seq_task[f] = (uint32_t)n_seq;
        db->n_multi += f_blocks > 1 ? 1u : 0u;
        if (f_seq && frame_off[f] <= in_size && frame_len[f] <= in_size - frame_off[f]) {
            db->frame_in_lo[f] = frame_off[f];
            db->frame_in_hi[f] = frame_off[f] + frame_len[f];
        }
        db->frame_out_off[f] = out_at;
        db->frame_out_cap[f] = f_cap;
        st.out_capacity_bytes += f_cap;
        PCarry cur[4] = {};  // what the units so far leave to the next: absolute table references (src 3) or the predefined table (2)
        uint32_t seen_seq = 0;
        const uint32_t frame_block0 = (uint32_t)n_blocks;
        for (uint32_t u = u0; u < u1; u++) {
            const FrameCount &c = counts[u];
            FrameBase &fb = bases[u];
            fb.block0 = (uint32_t)n_blocks;
            fb.seq0 = (uint32_t)n_seq;
            fb.hufb0 = (uint32_t)n_hufb;
            fb.fse_tab0 = (uint32_t)n_fse_tab;
            fb.fse_src0 = (uint32_t)n_fse_src;
            fb.fse_dev0 = (uint32_t)n_fse_dev;
            fb.huf_tab0 = (uint32_t)n_huf_tab;
            fb.huf_src0 = (uint32_t)n_huf_src;
            fb.huf_dev0 = (uint32_t)n_huf_dev;
            fb.tile0 = (uint32_t)n_tile;
            fb.rec0 = n_rec;
            fb.lit0 = lit_total;
            fb.out_off = out_at;
            fb.out_cap = f_cap;
            for (int k = 0; k < 4; k++) fb.in[k] = cur[k];
            fb.seen_seq = seen_seq;
            fb.frame_blocks = (uint32_t)f_blocks;
            fb.frame_block0 = frame_block0;
            fb.pad = (uint32_t)frame_status[f];
            for (int k = 0; k < 4; k++) {
                if (c.carry[k].src == 1) cur[k] = PCarry{(k == 0 ? fb.huf_dev0 : fb.fse_dev0) + c.carry[k].off, c.carry[k].log, 3, {0, 0}};
                else if (c.carry[k].src == 2) cur[k] = PCarry{0, c.carry[k].log, 2, {0, 0}};
            }
            seen_seq |= c.n_seq ? 1u : 0u;
            n_blocks += c.n_blocks;
            n_seq += c.n_seq;
            n_hufb += c.n_hufb;
            n_fse_tab += c.n_fse_tab;
            n_fse_src += c.n_fse_src;
            n_fse_dev += c.n_fse_dev;
            n_huf_tab += c.n_huf_tab;
            n_huf_src += c.n_huf_src;
            n_huf_dev += c.n_huf_dev;
            n_tile += c.n_tile;
            n_rec += c.n_rec;
            lit_total += c.lit_bytes;
            max_huf_bits = std::max(max_huf_bits, c.max_huf_bits);
            for (int k = 0; k < 3; k++) seq_logs[k] = std::max<uint32_t>(seq_logs[k], (c.max_seq_logs >> (8 * k)) & 0xFF);
            st.compressed_bytes += c.comp_bytes;
            st.n_sequences += c.n_rec;
            st.n_huf_streams += c.n_huf_streams;
            st.n_blocks[0] += c.n_raw;
            st.n_blocks[1] += c.n_rle;
            st.n_blocks[2] += c.n_comp;
        }
        if (u1 - u0 > 1 && frame_status[f] == MZD_OK) {
            DFrame df{};
            df.out_offset = out_at;
            df.out_capacity = f_cap;
            df.content_size = f_content;
            df.first_block = frame_block0;
            df.n_blocks = (uint32_t)f_blocks;
            df.plan_status = MZD_OK;
            df.has_checksum = (counts[u0].flags & 0x80000000u) && (counts[u1 - 1].flags & 0x40000000u) ? 1 : 0;  // header flag AND the 4 bytes were there
            df.checksum = df.has_checksum ? counts[u1 - 1].checksum : 0;
            host_frames.push_back(df);
            host_frame_idx.push_back(f);
        }
        out_at += (f_cap + 255) & ~255ull;
    }
    out_at += 256;  // tail slack: the execution kernel's 16-byte source loads may run past the last slab
    frame_seq_task[n_frames] = (uint32_t)n_seq;
    const uint64_t lim32 = 0xFFFFFFFFull;
    if (n_blocks > lim32 || n_seq > lim32 || 4 * n_hufb > lim32 || n_fse_dev > lim32 || n_huf_dev > lim32 || n_tile > lim32) {
        ctx->last_error = "batch too large for 32-bit work-list indices";
        return fail(MZD_ERR_UNSUPPORTED);
    }
    st.table_bytes = n_fse_src * 4 + n_huf_src * 2;
    st.scratch_bytes = n_rec * 8 + n_tile * 8 + lit_total;
    db->n_blocks = (uint32_t)n_blocks;
    db->n_huf_tasks = (uint32_t)(4 * n_hufb);
    db->n_seq_tasks = (uint32_t)n_seq;
    for (int k = 0; k < 3; k++) db->seq_cells[k] = 1u << std::min<uint32_t>(seq_logs[k], k == 2 ? 8u : 9u);
    db->huf_slot_cells = 1u << max_huf_bits;
    db->frame_seq_task = std::move(frame_seq_task);
    db->out_size = out_at;
    db->stats = st;
    // ---- everything the hot path needs, sized by the counts
    if (flags & MZD_BATCH_OUT_ON_DEVICE) {
        if (out_dev_size < out_at) {
            ctx->last_error = "output blob too small: the frames need " + std::to_string(out_at) + " bytes";
            return fail(MZD_ERR_DST_FULL);
        }
        db->d_out = out_dev;
    } else {
        if (!db->own_out) { db->d_out = nullptr; db->cap.out = 0; }
        ENSURE(db->d_out, db->cap.out, std::max<uint64_t>(out_at, 16));
        db->own_out = true;
    }
    ENSURE(db->d_frames, db->cap.frames, nf1 * sizeof(DFrame));
    ENSURE(db->d_blocks, db->cap.blocks, std::max<uint64_t>(n_blocks, 1) * sizeof(DBlock));
    ENSURE(db->d_sums, db->cap.sums, std::max<uint64_t>(n_blocks, 1) * sizeof(BlockSum));
    ENSURE(db->d_huf_tasks, db->cap.huf_tasks, std::max<uint64_t>(4 * n_hufb, 1) * sizeof(HufTask));
    ENSURE(db->d_seq_tasks, db->cap.seq_tasks, std::max<uint64_t>(n_seq, 1) * sizeof(SeqTask));
    ENSURE(db->d_fse_entries, db->cap.fse, std::max<uint64_t>(n_fse_dev, 1) * 4);
    ENSURE(db->d_huf_entries, db->cap.huf, std::max<uint64_t>(n_huf_dev, 2) * 2 + 8);
    ENSURE(db->d_recs, db->cap.recs, std::max<uint64_t>(n_rec, 1) * 8);
    ENSURE(db->d_tiles, db->cap.tiles, std::max<uint64_t>(n_tile, 1) * sizeof(TileBase));
    ENSURE(db->d_litbuf, db->cap.lit, lit_total + 64);
    db->n_recs = n_rec;
    db->n_tiles = n_tile;
    db->lit_bytes = lit_total;
    db->huf_out_bytes = lit_total;
    ENSURE(db->d_status, db->cap.status, nf1 * sizeof(int32_t));
    ENSURE(db->d_out_len, db->cap.out_len, nf1 * sizeof(uint64_t));
    HIP_OR_FAIL(hipMemsetAsync(db->d_status, 0xFF, nf1 * sizeof(int32_t), s));
    HIP_OR_FAIL(hipMemsetAsync(db->d_out_len, 0, nf1 * sizeof(uint64_t), s));
    ENSURE(tp.d_fse_tabs, tp.cap_fse_tabs, n_fse_tab * sizeof(FseBuildDesc));
    ENSURE(tp.d_fse_src, tp.cap_fse_src, n_fse_src * 4);
    ENSURE(tp.d_huf_tabs, tp.cap_huf_tabs, std::max<uint64_t>(n_huf_tab, 1) * sizeof(HufBuildDesc));
    ENSURE(tp.d_huf_src, tp.cap_huf_src, std::max<uint64_t>(n_huf_src, 1) * 2);
    // the predefined tables' counts (host arrays must outlive the async copies: synchronised below)
    FseBuildDesc pd[3];
    std::vector<uint32_t> psrc(predef_src[2] + (kDefN[2] + 1) / 2, 0);
    for (int k = 0; k < 3; k++) {
        pd[k] = FseBuildDesc{predef_src[k], po.predef_off[k], (uint8_t)kDefLog[k], (uint8_t)kDefN[k], 1, 0};
        for (int i = 0; i < kDefN[k]; i++) psrc[predef_src[k] + i / 2] |= (uint32_t)(uint16_t)kDef[k][i] << (16 * (i & 1));
    }
    HIP_OR_FAIL(hipMemcpyAsync(tp.d_fse_tabs, pd, sizeof pd, hipMemcpyHostToDevice, s));
    HIP_OR_FAIL(hipMemcpyAsync(tp.d_fse_src, psrc.data(), psrc.size() * 4, hipMemcpyHostToDevice, s));
    if (n_units) HIP_OR_FAIL(hipMemcpyAsync(tp.d_bases, bases.data(), (size_t)n_units * sizeof(FrameBase), hipMemcpyHostToDevice, s));
    po.frames = db->d_frames;
    po.blocks = db->d_blocks;
    po.huf_tasks = db->d_huf_tasks;
    po.seq_tasks = db->d_seq_tasks;
    po.fse_tabs = tp.d_fse_tabs;
    po.fse_src = tp.d_fse_src;
    po.huf_tabs = tp.d_huf_tabs;
    po.huf_src = tp.d_huf_src;
    // ---- pass 1: write the work lists and the table build descriptors; then build the tables
    HIP_OR_FAIL(hipEventRecord(t2, s));
    if (n_units)
        k_parse<1><<<n_wg, 64, 0, s>>>(db->d_in, in_size, tp.d_units, n_units, tp.d_scratch, tp.d_counts, tp.d_bases, po);
    for (size_t i = 0; i < host_frames.size(); i++)  // (the frames of several units; host_frames lives until the synchronisation below)
        HIP_OR_FAIL(hipMemcpyAsync(db->d_frames + host_frame_idx[i], &host_frames[i], sizeof(DFrame), hipMemcpyHostToDevice, s));
    HIP_OR_FAIL(hipEventRecord(t3, s));
    k_fse_build<<<(uint32_t)((n_fse_tab + 63) / 64), 64, 0, s>>>(tp.d_fse_tabs, (uint32_t)n_fse_tab, tp.d_fse_src, db->d_fse_entries);
    HIP_OR_FAIL(hipEventRecord(t4, s));
    if (n_huf_tab)
        k_huf_build<<<(uint32_t)((n_huf_tab + 63) / 64), 64, 0, s>>>(tp.d_huf_tabs, (uint32_t)n_huf_tab, tp.d_huf_src, db->d_huf_entries);
    HIP_OR_FAIL(hipGetLastError());
    // ---- heterogeneous work lists are ordered by size (see mzd_dbatch): the keys come back from the device, the lists are
    // gathered there
    db->seq_sorted = db->huf_sorted = false;
    bool have_order = false;
    db->long_frames = db->long_tasks = 0;
    if (n_seq > 64 || n_hufb > 64 || n_frames > 64) {
        const uint32_t ns = (uint32_t)n_seq, nq = (uint32_t)n_hufb, nk = std::max(ns, nq);
        std::vector<uint32_t> keys((size_t)ns + nq);
        if (nk) {
            ENSURE(tp.d_keys, tp.cap_keys, ((size_t)ns + nq) * 4);
            k_task_keys<<<(nk + 255) / 256, 256, 0, s>>>(db->d_seq_tasks, ns, db->d_huf_tasks, nq, tp.d_keys, tp.d_keys + ns);
            HIP_OR_FAIL(hipMemcpyAsync(keys.data(), tp.d_keys, keys.size() * 4, hipMemcpyDeviceToHost, s));
            HIP_OR_FAIL(hipStreamSynchronize(s));
        }
        ListOrder order;
        plan_order(keys.data(), ns, keys.data() + ns, nq, db->frame_out_cap.data(), n_frames, in_size, order);
        group_long_frames(keys.data(), ns, db->frame_seq_task, n_frames, ctx->num_cus, order);
        ENSURE(tp.d_perm, tp.cap_perm, (size_t)std::max<uint32_t>(nk, 1) * 4);
        if (!order.seq_perm.empty()) {
            ENSURE(tp.d_sorted, tp.cap_sorted, (size_t)ns * sizeof(SeqTask));
            HIP_OR_FAIL(hipMemcpyAsync(tp.d_perm, order.seq_perm.data(), (size_t)ns * 4, hipMemcpyHostToDevice, s));
            k_gather<SeqTask><<<(ns + 255) / 256, 256, 0, s>>>(db->d_seq_tasks, (SeqTask *)tp.d_sorted, tp.d_perm, ns, 1u);
            HIP_OR_FAIL(hipMemcpyAsync(db->d_seq_tasks, tp.d_sorted, (size_t)ns * sizeof(SeqTask), hipMemcpyDeviceToDevice, s));
            HIP_OR_FAIL(hipStreamSynchronize(s));  // (the host permutation array is reused below)
            db->seq_sorted = true;
            for (int c = 0; c < 3; c++) {
                db->seq_class_end[c] = order.seq_class_end[c];
                for (int k = 0; k < 3; k++) db->seq_class_cells[c][k] = 1u << std::min<uint32_t>(order.seq_class_logs[c][k], k == 2 ? 8u : 9u);
            }
        }
        if (!order.huf_perm.empty()) {
            ENSURE(tp.d_sorted, tp.cap_sorted, (size_t)nq * 4 * sizeof(HufTask));
            HIP_OR_FAIL(hipMemcpyAsync(tp.d_perm, order.huf_perm.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s));
            k_gather<HufTask><<<(4 * nq + 255) / 256, 256, 0, s>>>(db->d_huf_tasks, (HufTask *)tp.d_sorted, tp.d_perm, nq, 4u);
            HIP_OR_FAIL(hipMemcpyAsync(db->d_huf_tasks, tp.d_sorted, (size_t)nq * 4 * sizeof(HufTask), hipMemcpyDeviceToDevice, s));
            HIP_OR_FAIL(hipStreamSynchronize(s));
            db->huf_sorted = true;
            for (int c = 0; c < 3; c++) {
                db->huf_class_end[c] = order.huf_class_end[c];
                db->huf_class_long[c] = order.huf_class_long[c];
            }
        }
        if (!order.frame_order.empty()) {
            ENSURE(db->d_frame_order, db->cap.frame_order, (size_t)n_frames * 4);
            HIP_OR_FAIL(hipMemcpyAsync(db->d_frame_order, order.frame_order.data(), (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
            HIP_OR_FAIL(hipStreamSynchronize(s));
            have_order = true;
            if (db->seq_sorted) {
                db->long_frames = order.long_frames;
                db->long_tasks = order.long_tasks;
            }
        }
    }
    if (!have_order && db->d_frame_order) {  // a recycled slot whose previous batch had an order
        (void)hipFree(db->d_frame_order);
        db->d_frame_order = nullptr;
        db->cap.frame_order = 0;
    }
    HIP_OR_FAIL(hipStreamSynchronize(s));
    float a = 0, b2 = 0, c2 = 0;
    (void)hipEventElapsedTime(&a, t0, t1);
    (void)hipEventElapsedTime(&b2, t2, t3);
    (void)hipEventElapsedTime(&c2, t3, t4);
    db->parse_ms = a + b2;
    db->fse_build_ms = c2;
    db->n_fse_built = (uint32_t)n_fse_tab;
    db->n_huf_built = (uint32_t)n_huf_tab;
    db->n_fse_entries = (uint32_t)n_fse_dev;
    db->fse_dev_off.clear();  // table read-back (mzd_batch_read_*_table) is a host-planned-batch facility
    db->huf_dev_off.clear();
#undef HIP_OR_FAIL
#undef ENSURE
    cleanup();
    *out = db;
    return MZD_OK;
}

int mzd_batch_upload_frames(mzd_ctx *ctx, const uint8_t *in, uint64_t in_size, uint32_t flags, const uint64_t *frame_off,
                            const uint64_t *frame_len, uint32_t n_frames, uint8_t *out_dev, uint64_t out_dev_size, mzd_dbatch **out)
{
    if (!ctx || !out || (!in && in_size) || ((!frame_off || !frame_len) && n_frames) ||
        (flags & ~(uint32_t)(MZD_BATCH_IN_ON_DEVICE | MZD_BATCH_OUT_ON_DEVICE)) || ((flags & MZD_BATCH_OUT_ON_DEVICE) && !out_dev))
        return MZD_ERR_INVALID_ARG;
    *out = nullptr;
    // (round 5: a large frame's blocks are parsed side by side on the device -- the units of mzd_parse.hip --, so a batch with such a
    // frame no longer goes back to the host planner, and a blob that is resident on the device stays there)
    return upload_frames_impl(ctx, in, in_size, flags, frame_off, frame_len, n_frames, out_dev, out_dev_size, nullptr, ctx->stream, out);
}

// ---- streaming: batches of frames pipelined through `depth` recycled device slots

struct mzd_stream {
    mzd_ctx *ctx = nullptr;
    struct Slot {
        mzd_dbatch *db = nullptr;
        hipEvent_t ready = nullptr, run_done = nullptr, done = nullptr;
        int32_t *h_status = nullptr;  // pinned
        uint64_t *h_out_len = nullptr;
        size_t h_cap = 0;
        uint64_t ticket = 0;  // 0 = free
        uint32_t n_frames = 0;
        int error = MZD_OK;
    };
    std::vector<Slot> slots;
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    uint64_t next_ticket = 1;

};

mzd_stream *mzd_stream_create(mzd_ctx *ctx, uint32_t depth, int *err)
{
    if (!ctx || depth < 1 || depth > 8) {
        if (err) *err = MZD_ERR_INVALID_ARG;
        return nullptr;
    }
    (void)hipSetDevice(ctx->device);
    mzd_stream *st = new mzd_stream();
    st->ctx = ctx;
    st->slots.resize(depth);

    // The runtime multiplexes HIP streams onto a few hardware queues (4 by default: GPU_MAX_HW_QUEUES) and two
    // streams on one queue serialise: the decode runs on the context's own stream (+ its helper stream), so a
    // streaming context uses exactly four.
    st->s_run = ctx->stream;
    bool ok = hipStreamCreateWithFlags(&st->s_in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&st->s_out, hipStreamNonBlocking) == hipSuccess;
    for (auto &sl : st->slots) {
        sl.db = new mzd_dbatch();
        ok = ok && hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.run_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        if (err) *err = MZD_ERR_DEVICE;
        mzd_stream_destroy(st);
        return nullptr;
    }
    if (err) *err = MZD_OK;
    return st;
}

void mzd_stream_destroy(mzd_stream *st)
{
    if (!st) return;
    (void)hipSetDevice(st->ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &sl : st->slots) {
        if (sl.db) mzd_batch_free(st->ctx, sl.db);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.run_done) (void)hipEventDestroy(sl.run_done);
        if (sl.done) (void)hipEventDestroy(sl.done);
        (void)hipHostFree(sl.h_status);
        (void)hipHostFree(sl.h_out_len);
    }
    if (st->s_in) (void)hipStreamDestroy(st->s_in);
    if (st->s_out) (void)hipStreamDestroy(st->s_out);
    delete st;
}

int mzd_stream_submit(mzd_stream *st, const uint8_t *in, uint64_t in_size, const uint64_t *frame_off, const uint64_t *frame_len,
                      uint32_t n_frames, uint8_t *out_host, uint64_t out_cap, uint64_t *ticket)
{
    if (!st || !ticket || (!in && in_size) || ((!frame_off || !frame_len) && n_frames) || (!out_host && out_cap)) return MZD_ERR_INVALID_ARG;
    mzd_ctx *ctx = st->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mzd_stream::Slot &sl = st->slots[(st->next_ticket - 1) % st->slots.size()];
    if (sl.ticket != 0) {
        ctx->last_error = "mzd_stream_submit: all slots in flight; collect the oldest ticket with mzd_stream_wait first";
        return MZD_ERR_INVALID_ARG;
    }
    if (sl.h_cap < n_frames) {
        (void)hipHostFree(sl.h_status);
        (void)hipHostFree(sl.h_out_len);
        sl.h_status = nullptr;
        sl.h_out_len = nullptr;
        sl.h_cap = 0;
        const size_t n = (size_t)n_frames + n_frames / 4 + 64;
        HIP_TRY(ctx, hipHostMalloc((void **)&sl.h_status, n * sizeof(int32_t), hipHostMallocDefault));
        HIP_TRY(ctx, hipHostMalloc((void **)&sl.h_out_len, n * sizeof(uint64_t), hipHostMallocDefault));
        sl.h_cap = n;
    }
    // copy-in + planning on s_in: the host waits for s_in only; s_run keeps decoding the previous batches
    mzd_dbatch *db = nullptr;
    int rc = upload_frames_impl(ctx, in, in_size, 0, frame_off, frame_len, n_frames, nullptr, 0, sl.db, st->s_in, &db);
    if (rc != MZD_OK) return rc;
    if (db->out_size > out_cap + 256) {  // the 256 bytes of tail slack are not copied back
        ctx->last_error = "mzd_stream_submit: out_host too small: the frames need " + std::to_string(db->out_size - 256) + " bytes";
        return MZD_ERR_DST_FULL;
    }
    HIP_TRY(ctx, hipEventRecord(sl.ready, st->s_in));
    HIP_TRY(ctx, hipStreamWaitEvent(st->s_run, sl.ready, 0));
    rc = mzd_batch_run(ctx, db, st->s_run);
    if (rc != MZD_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(sl.run_done, st->s_run));
    // copy-out on s_out: overlaps the next batch's decode and copy-in
    HIP_TRY(ctx, hipStreamWaitEvent(st->s_out, sl.run_done, 0));
    // (a shader copy kernel into mapped host memory was tried instead of the DMA engine: 25.3 vs 20.9 ms per batch)
    if (db->out_size > 256) HIP_TRY(ctx, hipMemcpyAsync(out_host, db->d_out, db->out_size - 256, hipMemcpyDeviceToHost, st->s_out));
    if (n_frames) {
        HIP_TRY(ctx, hipMemcpyAsync(sl.h_status, db->d_status, (size_t)n_frames * sizeof(int32_t), hipMemcpyDeviceToHost, st->s_out));
        HIP_TRY(ctx, hipMemcpyAsync(sl.h_out_len, db->d_out_len, (size_t)n_frames * sizeof(uint64_t), hipMemcpyDeviceToHost, st->s_out));
    }
    HIP_TRY(ctx, hipEventRecord(sl.done, st->s_out));
    sl.n_frames = n_frames;
    sl.ticket = st->next_ticket++;
    *ticket = sl.ticket;
    return MZD_OK;
}

int mzd_stream_wait(mzd_stream *st, uint64_t ticket, int32_t *status, uint64_t *out_len, uint64_t *out_offset)
{
    if (!st || ticket == 0) return MZD_ERR_INVALID_ARG;
    mzd_ctx *ctx = st->ctx;
    mzd_stream::Slot &sl = st->slots[(ticket - 1) % st->slots.size()];
    if (sl.ticket != ticket) {
        ctx->last_error = "mzd_stream_wait: unknown or already collected ticket";
        return MZD_ERR_INVALID_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(sl.done));
    for (uint32_t f = 0; f < sl.n_frames; f++) {
        if (status) status[f] = sl.h_status[f];
        if (out_len) out_len[f] = sl.h_out_len[f];
        if (out_offset) out_offset[f] = sl.db->frame_out_off[f];
    }
    sl.ticket = 0;
    return MZD_OK;
}

void *mzd_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, std::max<uint64_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void mzd_host_free(void *p) { (void)hipHostFree(p); }

uint64_t mzd_batch_out_size(mzd_dbatch *db) { return db ? db->out_size : 0; }

int mzd_batch_frame_layout(mzd_dbatch *db, uint64_t *out_offset, uint64_t *out_capacity)
{
    if (!db) return MZD_ERR_INVALID_ARG;
    for (uint32_t f = 0; f < db->n_frames && f < db->frame_out_off.size(); f++) {
        if (out_offset) out_offset[f] = db->frame_out_off[f];
        if (out_capacity) out_capacity[f] = db->frame_out_cap[f];
    }
    return MZD_OK;
}

int mzd_batch_run(mzd_ctx *ctx, mzd_dbatch *db, void *stream_)
{
    if (!ctx || !db) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (the launch checks below read the thread's LAST error: one that somebody else's call left there -- another library in the
    // process probing a device that does not exist: "invalid device ordinal" in a pass that launched fine, round 6 -- is not this
    // pass's)
    (void)hipGetLastError();
    hipStream_t s = stream_ ? (hipStream_t)stream_ : ctx->stream;
    if (db->trimmed) {
        ctx->last_error = "mzd_batch_run on a batch that mzd_batch_trim has reduced to its output";
        return MZD_ERR_INVALID_ARG;
    }
    db->run_stream = s;
    db->run_pending = true;
    const uint32_t exec_threads = ctx->opt.exec_threads ? ctx->opt.exec_threads : 128;
    if (exec_threads % 64 || exec_threads > MZD_EXEC_MAX_THREADS) return MZD_ERR_INVALID_ARG;
    // LDS chunk of the execution kernel: default 8 KiB (up to 16 workgroups per CU); multiple of 1024
    uint32_t exec_cap = ctx->opt.exec_chunk ? ctx->opt.exec_chunk : 8192;
    exec_cap = std::min<uint32_t>(std::max<uint32_t>(exec_cap & ~1023u, 4096), kBlockMax);
    const size_t seq_lds = (size_t)kSeqChains16 * kSeqCellsPerChain * 2 + kSeqExtraLds16;
    (void)seq_lds;  // (k_seq's: libmzd_test.so)
    size_t exec_lds = (size_t)exec_cap + 32 + (exec_cap / 32 + 4) * 4 + 16;
    if (const char *e = exp_env("MZD_EXEC_MIN_LDS")) exec_lds = std::max<size_t>(exec_lds, (size_t)atoi(e));  // experiment: residency cap
    // k_huf residency cap: with every stream resident at once the active cache lines (one per lane)
    // overflow L2 and every refill goes to MALL/HBM; a minimum LDS request per workgroup limits the
    // number of resident wavefronts (opt.huf_min_lds bytes; 0, the default, = no cap: the same value on the sorted class launches below)
    const size_t huf_lds = std::max<size_t>((size_t)kHufQuads * db->huf_slot_cells * 2, ctx->opt.huf_min_lds);
    if (!ctx->attr_set) {
#ifdef MZD_TEST_KERNELS
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_seq, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)((size_t)kSeqChains16 * kSeqCellsPerChain * 2 + kSeqExtraLds16)));
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_seq_pipe, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kPipeFixedLds + kPipeMaxChains * kSeqCellsPerChain * 2));
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_huf_seg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_seq_q4, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         kQ4FixedLds + kQ4MaxChains * kSeqCellsPerChain * 2));
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_exec, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(kBlockMax + 32 + (kBlockMax / 32 + 4) * 4 + 16)));
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_huf, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 2048 * 2 + 16 + kHufTStageBytes));
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_huf_w, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 + 4 * kHwWaveBytes));
        ctx->attr_set = true;
    }
    // ---- split point: k_seq runs ceil(tasks / (chains per CU * CUs)) rounds of one chain latency each and
    // the last, partial round leaves most CUs idle.  The batch is cut at the frame where the full rounds
    // end: k_seq(tail) then runs on the caller's stream while k_exec(head) fills the idle CUs from a
    // second stream (frames are independent, so the two never touch the same data).
    // seq_variant 0 (default) and 2: k_seq_q4; 1: k_seq, the two-wavefront kernel; 3: k_seq_pipe.  k_seq_q4 and
    // k_seq_pipe address the bitstreams with 32-bit offsets from a window of the blob (larger blobs: window by window).
    if (ctx->opt.seq_variant > 3 || ctx->opt.huf_variant > 4 || ctx->opt.exec_variant > 5) return MZD_ERR_INVALID_ARG;
#ifndef MZD_TEST_KERNELS
    // The release library has ONE kernel per stage (+ k_exec for frames of 4 GiB and more): the second implementations the parity tests
    // force -- k_seq (seq_variant 1), k_seq_pipe (3), k_huf_seg (huf_variant 2), k_exec_b (exec_variant 2, 3) -- are compiled into
    // libmzd_test.so (-DMZD_TEST_KERNELS) only
    if (ctx->opt.seq_variant == 1 || ctx->opt.seq_variant == 3 || ctx->opt.huf_variant == 2 || ctx->opt.exec_variant == 2 || ctx->opt.exec_variant == 3) {
        ctx->last_error = "this kernel variant is a second implementation for the parity tests: it is in libmzd_test.so, not in the release library";
        return MZD_ERR_UNSUPPORTED;
    }
#endif
    const uint32_t sv = ctx->opt.seq_variant ? ctx->opt.seq_variant : 2u;
    const bool pipe = sv != 1;  // the kernels that address a window of the blob
    const bool q4 = sv == 2;
    // k_seq_pipe: two chains fewer than fit, so that ~6 KiB of every CU's LDS stay free and the small
    // k_huf workgroups run in k_seq's shadow instead of queueing for whole CUs (measured with 54 chains of 56:
    // 28.6 ms per step; 55: 29.5; 56: 32.2; 53: 29.2; 51: 30.9)
    // The lane-per-stream Huffman kernel beside the sequence stage cost that stage 2 ms of the pass (both fill the CU's
    // address unit).  With sequences to decode and many streams it runs FIRST, alone, with its transposed bulk phase (global
    // memory in 64-byte runs through an LDS staging area that does not fit beside the sequence stage).
    // (huf_variant 3 forces it; by default it takes tables of at most 32 cells -- the bulk phase is for MaxBits <= 5 --
    // and enough streams to fill the chip)
    // Heterogeneous batches (work lists ordered by size at upload, frames executed largest first) run their stages one after
    // the other on the caller's stream: the head / tail split and k_huf in the sequence stage's shadow are for batches whose
    // lists are in frame order.
    // Which execution kernel: k_exec (a workgroup of two wavefronts per frame, a lane per sequence) for batches of frames of one
    // size, pure Raw / RLE / literal-only batches (wide copies) and frames of 4 GiB and more (k_exec_b keeps frame positions in
    // 32 bits); k_exec_b (a wavefront per frame, a lane per output byte) for heterogeneous batches, where a large frame is a
    // long serial job and twice as many frames are in flight.
    // measured (round 3): the 65 536 text-like 128 KiB frames k_exec 10.9 ms, k_exec_b 11.7 ms; the reference's corpus
    // replicated to 4 GiB (frames of 0 to 1 MiB, executed largest first) k_exec 10.1 ms, k_exec_b 9.3 ms
    // (and batches of small frames: 131 072 frames of 4 KiB k_exec 0.98 ms, k_exec_b 0.71 ms)
    // Round 4: k_exec_c (mzd_exec_c.hip: k_exec_b's method, two bytes per lane and pass, fixed-point passes, a lean setup) is what
    // exec_variant 0 takes for every batch with sequences whose frames are below 4 GiB (k_exec_b and k_exec_c keep frame
    // positions in 32 bits) -- measured against k_exec / k_exec_b, whole pass (profiles/r4_exec_matrix.txt): 65 536 x 128 KiB
    // 20.0 against 21.05 / 21.3 ms; 32 768 / 16 384 / 8 192 frames 10.76 / 6.26 / 3.25 against 11.14 / 6.52 / 3.58 and 11.38 / 6.75 /
    // 3.60; 8 192 x 1 MiB 25.7 against 27.8 / 28.0; 131 072 x 4 KiB 1.92 against 2.38 / 2.06; the corpus 17.3 against 18.6 / 17.2.
    // k_exec stays for frames of 4 GiB and more and (variant 1) for the parity tests; k_exec_b for block mode and (2) the tests.
    bool exec_c = ctx->opt.exec_variant == 5 || (ctx->opt.exec_variant == 0 && db->n_seq_tasks > 0);
    bool exec_b = ctx->opt.exec_variant >= 2 && ctx->opt.exec_variant <= 4;
    // (both keep frame positions in 32 bits: a frame of 4 GiB or more -- also under a forced variant -- takes k_exec)
    for (uint32_t f = 0; (exec_c || exec_b) && f < db->n_frames; f++)
        if (db->frame_out_cap[f] >= (1ull << 32) - 65536) exec_c = exec_b = false;
    if (db->has_chunks) {
        // chunks of frames (ABI 9): k_exec_c and block mode with its passes know what a slab that begins with the frame's window is
        if (ctx->opt.exec_variant != 0 && ctx->opt.exec_variant != 4 && ctx->opt.exec_variant != 5) {
            ctx->last_error = "a batch with a chunk of a frame (MZD_FRAME_CONTINUES) takes exec_variant 0, 4 or 5";
            return MZD_ERR_UNSUPPORTED;
        }
        exec_c = true;  // (also without sequences: blocks of literals only)
    }
    const bool exec_b_serial = exec_b;  // the choice without block mode
    // k_exec_b takes opt.exec_chunk as EXTRA dynamic LDS on top of its 7.7 KiB (a residency cap), k_exec as its LDS chunk: a value that
    // suits k_exec (up to 128 KiB) must not make the k_exec_b launch fail -- clamped to what the default 64 KiB limit leaves
    const uint32_t xb_extra_lds = std::min<uint32_t>(ctx->opt.exec_chunk, 64u * 1024u - (uint32_t)sizeof(XbLds) - 256u);
    (void)xb_extra_lds;
    // Block mode (mzd_exec_blk.hip): every block its own job, NP passes and a fix-up walk -- for batches whose largest frame is a
    // longer serial job than NP passes over everything.  The model: a frame's workgroup alone makes a 128 KiB block in ~0.42 ms,
    // the chip 5 120 of them in ~0.85 ms; a fix-up step is ~6 us.  (exec_variant 3 forces it: the parity tests.)
    bool blk = false;
    uint32_t blk_np = 0;
    uint64_t blk_maxcap = 0;
    // A job is a SEGMENT of blk_gs consecutive blocks, executed in order: a fix-up step per segment instead of per block (one
    // 1 GiB frame: 8 192 steps of 1.7 us were 14 of its 22.7 ms), as long as there are jobs enough to fill the chip.
    // (exec_variant 3 forces block mode with segments of one block, 4 with segments of four: the parity tests run both.)
    uint32_t blk_gs = 1;
    // (the frames that count for the walk's policy are the ones of more than one block: 2 000 frames of 128 KiB and two of 64 MiB
    // are a batch of TWO frames for it -- it used to get one fix-up workgroup per frame: 78.6 ms, `profiles/r4_mixed_batches.txt`)
    const uint32_t n_walk = std::max(db->n_multi, 1u);
    if (db->n_seq_tasks > 0 && db->n_frames > 0 && (ctx->opt.exec_variant == 0 || ctx->opt.exec_variant == 3 || ctx->opt.exec_variant == 4)) {
        for (uint32_t f = 0; f < db->n_frames; f++) blk_maxcap = std::max<uint64_t>(blk_maxcap, db->frame_out_cap[f]);
        blk_np = blk_maxcap <= (1u << 23) ? 3u : 4u;
        const double chip = std::max(0.4, (double)db->out_size / kBlockMax / 5120.0 * 0.85);
        // (measured, 8 GiB of output as n frames whose blocks reach back, serial / block mode per pass: 2 048 x 4 MiB 30.2 / 82.1 ms;
        // 512 x 16 MiB 91.8 / 117.0; 256 x 32 MiB 168 / 98.6; 128 x 64 MiB 325 / 97.2 -- a frame's workgroup makes a block in
        // ~0.42 ms; the passes cost ~12 % more than the plain kernel, the fix-up walk ~5 ms per GiB + 6 us per block)
        // (the largest frame's serial walk: from its blocks' sequence counts when the host planned the batch -- a frame's output BOUND
        // says little: the reference's corpus has frames of 70 blocks of a few KiB each, bound 9 MiB, and batches of 0.4-1.5 GiB of
        // it took block mode, four passes, for an execution stage of 7-15 ms where the blocks in order take 6)
        const double t_serial = std::max(db->max_frame_serial_ms > 0 ? db->max_frame_serial_ms : (double)blk_maxcap / kBlockMax * 0.42, chip);
        const double t_blk = blk_np * chip * 1.12 + (double)db->out_size / (1u << 30) * 5.0 + (double)blk_maxcap / kBlockMax * 0.006 + 0.2;
        // (`profiles/r4_corpus_sizes.txt`: 1 GiB of the corpus 19.5 ms in block mode against 11.9 with the blocks in order)
        blk = blk_maxcap < (1ull << 31) - 65536 && (ctx->opt.exec_variant >= 3 || t_blk < 0.85 * t_serial);
        // (a chunk of ONE frame: nothing else is there to fill the chip, and what its blocks cost a single wavefront is not only
        // their sequences -- long matches far back are read from the slab 128 bytes at a time: chunks of 64 MiB of such content,
        // few sequences per block, were walked in order at 0.4 GB/s, 170 ms each, where block mode takes 5; round 6)
        if (db->has_chunks && db->n_frames == 1 && db->n_blocks >= 32 && blk_maxcap < (1ull << 31) - 65536) blk = true;
        if (blk) {
            exec_b = true;
            exec_c = false;
        }
        if (ctx->opt.exec_variant == 4) blk_gs = 4;
        else if (ctx->opt.exec_variant == 0) {
            // (one 1 GiB frame, jobs of 1 / 4 blocks: passes 4 x 1.9 / 2.9 ms, fix-up 12.6 / 5.1 ms; 64 x 128 MiB: 65.4 / 67.3 ms per
            // pass -- with many frames the fix-up walks are short anyway and larger jobs fill the chip's last round worse)
            // (round 4, the walk's workgroups over all XCDs -- see the launch: one 1 GiB frame, jobs of 2 / 3 / 4 / 6 blocks with as many
            // workgroups as give every thread one chunk of a job: 40.2 / 32.3 / 29.3 / 31.2 ms; 2 x 512 MiB jobs of 2 / 4: 29.0 / 27.0;
            // 4 x 256 MiB 24.8 / 26.5; 8 x 256 MiB 38.2 / 34.1; profiles/r4_blk_fixup_spread.txt)
            if (const char *e = exp_env("MZD_EXP_BLK_GS")) blk_gs = (uint32_t)std::max(1, atoi(e));
            // (with every pass of every job in ONE launch -- below -- longer jobs cost the passes little: one 1 GiB frame, jobs of 4 / 8 /
            // 16 blocks: 25.2 / 19.1 / 19.6 ms; 2 x 512 MiB 4 / 8: 23.1 / 16.7; 4 x 256 MiB: 22.3 / 17.0; 8 x 256 MiB: 32.5 / 30.0; one 256 MiB
            // frame 2 / 4 / 8: 11.5 / 8.6 / 9.0; one 64 MiB frame 1 / 2 / 4: 6.4 / 5.2 / 5.4; 16 x 128 MiB 1 / 2 / 4: 30.4 / 29.7 / 28.0;
            // profiles/r4_blk_fused_passes.txt)
            else if (n_walk <= 16) {
                blk_gs = db->n_blocks >= 8192 ? 8u : (db->n_blocks >= 2048 ? 4u : (db->n_blocks >= 512 ? 2u : 1u));
            } else if (n_walk <= 64 && db->n_blocks / 2 <= 20480u) {
                blk_gs = 2u;  // (32 x 128 MiB, jobs of 1 / 2: 49.8 / 47.5 ms; 16 x 128 MiB 4 / 8: 27.8 / 26.5)
            }
        }
    }
    if (blk) {
        auto ensure = [&](auto *&ptr, size_t &cap, size_t bytes) -> hipError_t {
            if (ptr && cap >= bytes) return hipSuccess;
            if (ptr) (void)hipFree((void *)ptr);
            ptr = nullptr;
            cap = 0;
            const hipError_t e = hipMalloc((void **)&ptr, bytes);
            if (e == hipSuccess) cap = bytes;
            return e;
        };
        const uint64_t stride = (db->out_size + 255) & ~(uint64_t)255, pstride = (blk_maxcap + 64 + 255) & ~(uint64_t)255;
        const size_t pat_before = db->cap_pat;
        // (passes - 1 copies of the output layout + the patterns: a batch that leaves no room for them walks its frames' blocks
        // in order instead)
        const bool got = ensure(db->d_jobs, db->cap_jobs, (size_t)std::max<uint32_t>(db->n_blocks, 1) * sizeof(BJob)) == hipSuccess &&
                         ensure(db->d_heads, db->cap_heads, ((size_t)db->n_blocks + 2) * 4) == hipSuccess &&
                         ensure(db->d_bframes, db->cap_bframes, (size_t)db->n_frames * sizeof(BFrame)) == hipSuccess &&
                         ensure(db->d_fixdone, db->cap_fixdone, (size_t)std::min<uint32_t>(n_walk, 1024u) * kFixMaxG * sizeof(uint32_t)) == hipSuccess &&
                         ensure(db->d_walk, db->cap_walk, ((size_t)db->n_frames + 1) * 4) == hipSuccess &&
                         ensure(db->d_planes, db->cap_planes, (size_t)(blk_np - 1) * stride + 256) == hipSuccess &&
                         ensure(db->d_pat, db->cap_pat, (size_t)blk_np * pstride) == hipSuccess;
        if (db->cap_pat != pat_before) db->pat_n = db->pat_np = 0;
        if (!got) {
            (void)hipGetLastError();
            blk = false;
            exec_b = exec_b_serial || ctx->opt.exec_variant >= 3;
            exec_c = ctx->opt.exec_variant == 0 || db->has_chunks;
        }
    }
    const bool serial = db->seq_sorted || db->huf_sorted || db->d_frame_order != nullptr || blk;
    // Heterogeneous batches: the Huffman kernel runs on the second stream BESIDE the sequence stage -- that stage is bound by
    // its longest chain there (real data: 42 k sequences = 5.9 ms of a 7.5 ms kernel with most CUs idle), the Huffman
    // classes with large tables (up to 64 KiB of LDS per wavefront) fill the CUs it leaves.
    // (block mode too, round 4: the frames' literals are not needed before the passes -- one 64 MiB frame 5.18 -> 4.77 ms, 100 frames of
    // the reference's corpus, which take block mode, 12.5 -> 10.2 ms, 64 x 128 MiB 83.3 -> 82.9; profiles/r4_huf_beside_block_mode.txt)
    const bool huf_het_beside = serial && ctx->opt.huf_variant == 0 && db->n_seq_tasks > 0 && !exp_env("MZD_EXP_HET_HUF_FIRST");
    // (no sequences in the batch -- BASELINE config 3: nothing to run the Huffman kernel beside; on the caller's stream the pass is three
    // launches in a row instead of two hand-overs between streams, 10-20 us each of a 0.4 ms pass)
    const bool huf_first = (serial && !huf_het_beside) || ctx->opt.huf_variant == 3 || (ctx->opt.huf_variant == 4 && db->n_seq_tasks > 0) || db->n_seq_tasks == 0 ||
                           (ctx->opt.huf_variant == 0 && db->n_seq_tasks > 0 && db->huf_slot_cells <= 32 &&
                            db->n_huf_tasks >= 64u * (uint32_t)std::max(ctx->num_cus, 1) &&
                            // ... and more chains than one round of the sequence stage: with a single round (the 8 192-frame shard
                            // of configs[4]: 32 chains per CU) LDS is free beside the chains and the Huffman kernel hides in their
                            // one latency -- 3.81 -> 3.55 ms; 16 384 frames (64 chains per CU) 6.52 -> 6.38-6.53, 32 768 frames
                            // 11.11 -> 11.18: not there
                            db->n_seq_tasks > (uint32_t)kQ4Chains * (uint32_t)std::max(ctx->num_cus, 1) && !exp_env("MZD_EXP_HUF_BESIDE"));
    // (heterogeneous batches: the sequence workgroups keep ALL of their CU's LDS, the Huffman wavefronts get the CUs that have
    // none -- beside a long chain they slowed its step: 7.5 -> 10.4 ms for the kernel)
    uint32_t nch = q4 ? (uint32_t)(huf_first || huf_het_beside || db->n_huf_tasks == 0 ? kQ4Chains : kQ4ChainsBeside) : (pipe ? (uint32_t)kPipeMaxChains - 2u : (uint32_t)kSeqChains16);
    if (const char *e = exp_env("MZD_SEQ_NCH")) if (pipe) nch = std::min<uint32_t>(nch, std::max(1, atoi(e)));  // experiment: chains per workgroup
    // k_seq_q4 sizes a chain's LDS slot to the batch's largest tables (less to stage, more LDS left for the Huffman
    // workgroups beside it).  With small tables TWO workgroups share a CU: the kernel holds 94 VGPRs (five wavefronts per
    // SIMD, a workgroup is nine wavefronts), so LDS decides -- 54 chains of up to ~580 cells each.
    const uint32_t q4_slot_cells = db->seq_cells[0] + db->seq_cells[1] + db->seq_cells[2];
    auto q4_lds = [&](uint32_t per_wg) { return (size_t)kQ4FixedLds + (size_t)per_wg * q4_slot_cells * 2; };
    const uint32_t wg_per_cu = q4 && 2 * q4_lds(nch) <= (size_t)160 * 1024 ? 2u : 1u;
    const uint64_t per_round = (uint64_t)nch * wg_per_cu * (uint64_t)(ctx->opt.assume_cus ? ctx->opt.assume_cus : (uint32_t)ctx->num_cus);
    uint32_t fA = db->n_frames;
    if (!ctx->opt.no_split && !serial && db->n_seq_tasks > per_round && db->n_seq_tasks % per_round != 0) {
        const uint64_t lim = (db->n_seq_tasks / per_round) * per_round;
        // last frame boundary at or below the limit
        uint32_t lo = 0, hi = db->n_frames;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) / 2;
            if (db->frame_seq_task[mid] <= lim) lo = mid;
            else hi = mid - 1;
        }
        if (lo > 0 && lo < db->n_frames) {
            fA = lo;
        }
    }
    const bool split = fA < db->n_frames;
    db->last_pass = (blk ? MZD_PASS_BLOCK_MODE : 0u) | (exec_c ? MZD_PASS_EXEC_C : 0u) | (exec_b ? MZD_PASS_EXEC_B : 0u) |
                    (split ? MZD_PASS_SPLIT : 0u);

    hipEvent_t *ev = nullptr;
    if (ctx->timing) {
        if (ctx->runs >= 1024) ctx->runs = 0;  // bounded ring
        while (ctx->ev.size() < (ctx->runs + 1) * kEvPerRun) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->ev.push_back(e);
        }
        if (ctx->run_split.size() < ctx->runs + 1) ctx->run_split.resize(ctx->runs + 1);
        ctx->run_split[ctx->runs] = (uint8_t)((split ? 1 : 0) | (huf_first ? 2 : 0));  // bit 1: k_huf ran first, on the caller's stream
        ev = ctx->ev.data() + ctx->runs * kEvPerRun;
    }
    bool seq_pack = false;  // the launches to come have the execution stage of earlier frames beside them
    hipStream_t seq_stream = s;  // (the stream the next sequence launch goes to)
    auto launch_seq_tasks = [&](uint32_t first, uint32_t count, bool use_pipe, uint64_t base) {
        if (!count) return;
        hipStream_t s = seq_stream;
        if (use_pipe) {
            // chains per workgroup: as few per workgroup as spread the chains over all CUs (a lone chain per CU when the batch is
            // small: the step latency does not depend on the number of lanes) -- unless the execution stage of earlier frames runs
            // beside this launch (the tail of a split batch): then full workgroups, so that a partial round sits on as few CUs as
            // hold it and leaves the others to that stage (same box, spread against packed: 65536 chains 20.03 -> 19.86 ms, 16384
            // chains 6.28 -> 5.60 ms, 32768 chains 10.70 -> 10.02 ms; profiles/r4_seq_tail_packed.txt)
            const uint64_t cus = (uint64_t)std::max(ctx->num_cus, 1) * wg_per_cu;  // workgroups resident at a time
            const uint64_t rounds = (count + cus * nch - 1) / (cus * nch);
            const uint32_t per_wg = rounds > 1 || seq_pack ? nch : (uint32_t)std::min<uint64_t>(nch, (count + cus - 1) / cus);
            if (q4)
                k_seq_q4<<<(count + per_wg - 1) / per_wg, kQ4Threads, q4_lds(per_wg), s>>>(
                    db->d_in, db->d_seq_tasks + first, count, db->d_fse_entries, db->d_recs, db->d_tiles, db->d_sums, per_wg, base,
                    db->seq_cells[0], db->seq_cells[1], db->seq_cells[2]);
#ifdef MZD_TEST_KERNELS
            else
                k_seq_pipe<<<(count + per_wg - 1) / per_wg, 256, kPipeFixedLds + (size_t)per_wg * kSeqCellsPerChain * 2, s>>>(
                    db->d_in, db->d_seq_tasks + first, count, db->d_fse_entries, db->d_recs, db->d_tiles, db->d_sums, per_wg, base);
#endif
        } else {
#ifdef MZD_TEST_KERNELS
            k_seq<<<(count + kSeqChains16 - 1) / kSeqChains16, 128, seq_lds, s>>>(
                db->d_in, db->d_seq_tasks + first, count, db->d_fse_entries, db->d_recs, db->d_tiles, db->d_sums);
#endif
        }
    };
    // Sequence decode of the frames [f0, f1).  k_seq_pipe addresses the bitstreams with 32-bit offsets from the
    // front slack of a WINDOW of the blob: the frames are cut into runs whose bitstreams span < 4 GiB, one launch
    // per run (one run unless the blob is that large); a single frame beyond that takes k_seq.
    auto launch_seq = [&](uint32_t f0, uint32_t f1) {
        if (db->seq_sorted) {
            // the whole list in ONE launch, longest chains first (its bitstreams are within 4 GiB of the blob's start): the
            // longest chain of real data (42 k sequences = 5.9 ms) is a floor the other workgroups fill in behind.  (A launch per
            // length class, each with the LDS slot of its own tables -- seq_class_* -- cost 10.2 ms against 7.5: every launch has
            // its own under-occupied tail, and the class of the long chains is less than one round.)
            launch_seq_tasks(0, db->n_seq_tasks, pipe, 0);
            return;
        }
        if (!pipe) {
            launch_seq_tasks(db->frame_seq_task[f0], db->frame_seq_task[f1] - db->frame_seq_task[f0], false, 0);
            return;
        }
        const uint64_t kWindow = ctx->opt.seq_window_kib ? (uint64_t)ctx->opt.seq_window_kib << 10 : (1ull << 32) - 2 * MZD_IN_PAD - 4096;
        // greedy cut first: how many launches does the range need at least?
        auto cut = [&](uint32_t g, uint32_t limit_tasks, uint64_t &lo_out) -> uint32_t {  // frames [g, e) of one launch
            uint64_t lo = ~0ull, hi = 0;
            uint32_t e = g;
            const uint32_t t0 = db->frame_seq_task[g];
            for (; e < f1; e++) {
                if (e > g && db->frame_seq_task[e] - t0 >= limit_tasks) break;
                if (db->frame_in_lo[e] > db->frame_in_hi[e]) continue;  // no sequences
                const uint64_t nlo = std::min(lo, db->frame_in_lo[e]), nhi = std::max(hi, db->frame_in_hi[e]);
                if (nhi - nlo > kWindow) break;
                lo = nlo;
                hi = nhi;
            }
            lo_out = lo;
            return e;
        };
        uint32_t n_launch = 0;
        for (uint32_t g = f0; g < f1;) {
            uint64_t lo;
            const uint32_t e = cut(g, 0xFFFFFFFFu, lo);
            g = e == g ? g + 1 : e;
            n_launch++;
        }
        // several launches: give them equal numbers of whole rounds rather than a full window and a remainder
        uint32_t limit = 0xFFFFFFFFu;
        if (n_launch > 1) {
            const uint64_t total = db->frame_seq_task[f1] - db->frame_seq_task[f0];
            const uint64_t round = (uint64_t)nch * wg_per_cu * (uint64_t)std::max(ctx->num_cus, 1);
            limit = (uint32_t)std::min<uint64_t>(((total + n_launch - 1) / n_launch + round - 1) / round * round, 0xFFFFFFFFu);
        }
        uint32_t g = f0;
        while (g < f1) {
            uint64_t lo;
            uint32_t e = cut(g, limit, lo);
            if (e == g) {  // one frame wider than the window: k_seq_q4 with a window per WORKGROUP (round 6; k_seq_pipe has none: k_seq)
                launch_seq_tasks(db->frame_seq_task[g], db->frame_seq_task[g + 1] - db->frame_seq_task[g], q4, q4 ? ~0ull : 0);
                e = g + 1;
            } else {
                launch_seq_tasks(db->frame_seq_task[g], db->frame_seq_task[e] - db->frame_seq_task[g], true, lo == ~0ull ? 0 : lo);
            }
            g = e;
        }
    };
    const bool no_exec = exp_env("MZD_DEBUG_SEQ_ONLY") != nullptr;  // debugging: entropy stages only (records via mzd_batch_debug_read)
    auto launch_exec = [&](hipStream_t st, uint32_t first, uint32_t count) {
        if (!count || no_exec) return;
        // frames in the order of d_frame_order when the batch has one (largest first), else in batch order
        if (blk) {
            // (the whole batch: block mode never splits)
            const uint64_t stride = (db->out_size + 255) & ~(uint64_t)255, pstride = (blk_maxcap + 64 + 255) & ~(uint64_t)255;
            (void)hipMemsetAsync(db->d_heads, 0, 4, st);
            (void)hipMemsetAsync(db->d_walk, 0, 4, st);
            k_blk_scan<<<db->n_frames, 64, 0, st>>>(db->d_frames, db->d_blocks, db->d_sums, db->d_jobs, db->d_bframes, blk_gs, db->d_heads, db->d_walk,
                                                    db->d_frame_hist);
            if (db->pat_n != (uint32_t)(blk_maxcap + 64) || db->pat_np != blk_np) {
                k_blk_pattern<<<(uint32_t)((blk_maxcap + 64 + 1023) / 1024), 256, 0, st>>>(db->d_pat, pstride, (uint32_t)(blk_maxcap + 64), blk_np);
                db->pat_n = (uint32_t)(blk_maxcap + 64);
                db->pat_np = blk_np;
            }
            // the passes: k_exec_c's method (exec_variant 0 and 4), k_exec_b's (3: the parity tests keep both alive)
            bool blk_xc = ctx->opt.exec_variant != 3;
            if (exp_env("MZD_EXP_BLK_XB")) blk_xc = false;  // experiment: the round-3 passes
            // ... all of them in ONE launch, a job's passes in neighbouring wavefronts: they are independent of each other, few large
            // frames have fewer jobs than the chip holds wavefronts (one 1 GiB frame: 2 048 jobs of four blocks, a pass 2.4 ms each
            // whatever runs beside it), and the passes of a job read the same records and literals
            // (16 x 128 MiB, 16 384 jobs: 31.1 -> 30.4 ms; 64 x 128 MiB, 65 536 jobs: 83.3 -> 85.5 -- a full chip gains nothing from the
            // passes side by side and loses by their interleaving: a launch per pass from 20 480 jobs on)
            bool fused = blk_xc && db->n_blocks / blk_gs <= 20480u && !exp_env("MZD_EXP_BLK_SERIAL_PASSES");
            if (fused) {
                const XbBlk bk{db->d_jobs, db->d_heads, db->d_bframes, db->d_pat, 0u, blk_np, pstride, stride, db->d_planes};
                k_exec_c<true, 8192><<<db->n_blocks * blk_np, 64, 0, st>>>(db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_sums, db->d_recs,
                                                                  db->d_litbuf, db->d_status, db->d_out_len, nullptr, 0u, bk, nullptr);
            }
            for (uint32_t p = 0; p < (fused ? 0u : blk_np); p++) {
                uint8_t *plane = p == 0 ? db->d_out : db->d_planes + (size_t)(p - 1) * stride;
                const XbBlk bk{db->d_jobs, db->d_heads, db->d_bframes, db->d_pat + (size_t)p * pstride, p, 0u, 0ull, 0ull, nullptr};
                // (as many wavefronts as blocks: the ones beyond the job list exit)
                if (blk_xc)
                    k_exec_c<true, 8192><<<db->n_blocks, 64, 0, st>>>(db->d_in, plane, db->d_frames, db->d_blocks, db->d_sums, db->d_recs, db->d_litbuf,
                                                             db->d_status, db->d_out_len, nullptr, 0u, bk, nullptr);
#ifdef MZD_TEST_KERNELS
                else
                    k_exec_b<true><<<db->n_blocks, 64, xb_extra_lds, st>>>(db->d_in, plane, db->d_frames, db->d_blocks, db->d_sums, db->d_recs,
                                                                        db->d_litbuf, db->d_status, db->d_out_len, nullptr, 0u, bk);
#endif
            }
            // fix-up workgroups per frame: all of a frame's must be resident together (they wait for each other)
            // (frames whose blocks reach back -- 64 x 128 MiB, 8 / 16 / 32 per frame: 113.8 / 103.7 / 114.8 ms per pass; one frame of
            // 1 GiB, 32 / 64 with jobs of two blocks: 49.4 / 44.4 ms, 32 / 64 / 128 with jobs of one: 56.0 / 67.4 / 92.2 ms -- every
            // workgroup is a poller of its frame's counter)
            uint32_t G = n_walk >= 1024 ? 1u : std::min<uint32_t>(64u, 1024u / n_walk);
            // Few frames (up to eight): a frame's workgroups on ALL XCDs instead of on one -- one XCD walking a 1 GiB frame moves its
            // 5 GiB through one L2 (54.8 ms per pass, whatever the number of steps); 128 workgroups over the chip, jobs of four blocks:
            // 29.3 ms.  As many workgroups as give every thread ONE 16-byte chunk of a job (32 per block of the job: a thread's second
            // chunk would be loaded behind the wait, idle workgroups are pollers: 96 / 128 / 160 at jobs of four 40.6 / 29.3 / 39.5 ms),
            // 256 at most over the frames.
            // Up to four frames 128 workgroups in all, beyond that 256, sixteen per frame at least (2 x 512 MiB, 64 / 128 per frame:
            // 27.0 / 28.9 ms; 4 x 256 MiB 32 / 64: 24.8 / 32.8; 8 x 256 MiB 16 / 32: 43.1 / 35.0; 16 x 128 MiB 8 / 16 spread / 64 on one XCD
            // each: 41.3 / 32.6 / 51.4; 32 x 128 MiB 8 / 16 spread / 32 on one XCD: 55.9 / 51.7 / 57.5; 64 x 128 MiB 2 / 4 / 8 spread / 16
            // on one XCD each: 141.6 / 103.3 / 89.8 / 88.9 -- every workgroup is a poller of its frame's counter).
            bool spread = n_walk <= 64;
            // (round 4, later, with the passes in one launch and longer jobs: 256 workgroups in all up to four frames -- one 1 GiB frame, jobs
            // of eight, 128 / 192 / 256: 23.1 / 26.2 / 19.1 ms; 2 x 512 MiB 64 / 128 per frame: 22.4 / 16.7; 4 x 256 MiB 32 / 64: 21.6 / 17.0 --,
            // 512 up to sixteen -- 8 x 256 MiB 32 / 64: 31.1 / 30.0; 16 x 128 MiB, jobs of four, 32: 28.0)
            if (spread)
                G = std::min<uint32_t>(32u * blk_gs, n_walk <= 4 ? 256u / n_walk : (n_walk <= 16 ? 512u / n_walk : std::max<uint32_t>(256u / n_walk, 16u)));
            if (const char *e = exp_env("MZD_EXP_BLK_G")) G = (uint32_t)std::min(256, std::max(1, atoi(e)));  // experiment
            // (G > 1: the workgroups of a frame wait for each other.  Should some of them not be resident -- another stream or
            // process on the GPU --, the waiters give up after a bounded wait and a second launch, one workgroup per such frame,
            // finishes the frame's walk from what `d_fixdone` says each workgroup got done: slower, never wrong, never a hang)
            if (const char *e = exp_env("MZD_EXP_BLK_SPREAD")) spread = atoi(e) != 0;
            spread = spread && G > 1;
            const uint32_t spread_arg = spread ? 1u : 0u, fix_wgs = n_walk * G * (G > 1 && !spread ? 8u : 1u);
            if (G > 1) (void)hipMemsetAsync(db->d_fixdone, 0, (size_t)n_walk * kFixMaxG * sizeof(uint32_t), st);
            const uint32_t tb = ctx->test_fixup_bail;
            if (blk_np == 3) {
                k_blk_fixup<3, false><<<fix_wgs, 256, 0, st>>>(db->d_out, db->d_planes, db->d_planes + stride, nullptr, db->d_frames,
                                                                      db->d_jobs, db->d_bframes, G, spread_arg, db->d_fixdone, tb, db->d_walk);
                if (G > 1)
                    k_blk_fixup<3, true><<<n_walk, 256, 0, st>>>(db->d_out, db->d_planes, db->d_planes + stride, nullptr, db->d_frames, db->d_jobs,
                                                                       db->d_bframes, G, spread_arg, db->d_fixdone, 0u, db->d_walk);
            } else {
                k_blk_fixup<4, false><<<fix_wgs, 256, 0, st>>>(db->d_out, db->d_planes, db->d_planes + stride,
                                                                      db->d_planes + 2 * stride, db->d_frames, db->d_jobs, db->d_bframes, G, spread_arg,
                                                                      db->d_fixdone, tb, db->d_walk);
                if (G > 1)
                    k_blk_fixup<4, true><<<n_walk, 256, 0, st>>>(db->d_out, db->d_planes, db->d_planes + stride, db->d_planes + 2 * stride,
                                                                       db->d_frames, db->d_jobs, db->d_bframes, G, spread_arg, db->d_fixdone, 0u, db->d_walk);
            }
            k_blk_final<<<(db->n_frames + 255) / 256, 256, 0, st>>>(db->d_frames, db->d_jobs, db->d_bframes, db->d_status, db->d_out_len, db->n_frames);
            return;
        }
        if (exec_c) {
            size_t xc_extra_lds = 0;  // experiment: extra dynamic LDS per frame = fewer frames in flight per CU
            if (const char *e = exp_env("MZD_EXP_XC_LDS")) xc_extra_lds = (size_t)std::max(0, atoi(e));
            // Which ring (round 5, `profiles/r5_exec_ring.txt`): 8 KiB at 16 frames per CU serves 64 % of text's matches from LDS instead of
            // 52 % and wins where the frames are all alike and long enough to have far matches (config 4: the execution stage of the
            // split pass 13.9 -> 13.2 ms, the pass 18.27 -> 17.95; 8 192 x 1 MiB 23.6 -> 23.4; block mode, always: 1 x 1 GiB 18.8 -> 18.0);
            // 4 KiB at 20 frames per CU where the frames in flight count -- heterogeneous batches, whose largest frames are serial jobs
            // (real data 1 GiB 10.8 -> 12.0 ms with the large ring), and small frames (131 072 x 4 KiB 1.51 -> 1.59)
            bool win8 = db->d_frame_order == nullptr && !db->seq_sorted && db->n_frames > 0 && db->out_size / db->n_frames >= 32768;
            if (const char *e = exp_env("MZD_EXP_XC_WIN")) win8 = atoi(e) == 8192;
            if (win8)
                k_exec_c<false, 8192><<<count, 64, xc_extra_lds, st>>>(db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_sums, db->d_recs, db->d_litbuf,
                                                                     db->d_status, db->d_out_len, db->d_frame_order, first, XbBlk{}, db->d_frame_hist);
            else
                k_exec_c<false, 4096><<<count, 64, xc_extra_lds, st>>>(db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_sums, db->d_recs, db->d_litbuf,
                                                                     db->d_status, db->d_out_len, db->d_frame_order, first, XbBlk{}, db->d_frame_hist);
            return;
        }
#ifdef MZD_TEST_KERNELS
        if (exec_b) {
            // (opt.exec_chunk: extra dynamic LDS per frame = a residency cap; frames in flight vs cache footprint of their slabs)
            k_exec_b<false><<<count, 64, xb_extra_lds, st>>>(db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_sums, db->d_recs,
                                                                   db->d_litbuf, db->d_status, db->d_out_len, db->d_frame_order, first, XbBlk{});
            return;
        }
#endif
        k_exec<<<count, exec_threads, exec_lds, st>>>(db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_sums,
                                                     db->d_recs, db->d_tiles, db->d_litbuf, db->d_status,
                                                     db->d_out_len, exec_cap, db->d_frame_order, first);
    };
    // optional integrity check of the regenerated frames (extension: the reference never verifies it)
    auto launch_verify = [&](hipStream_t st, uint32_t first, uint32_t count) {
        if (!count || !ctx->opt.verify_checksum) return;
        k_xxh64<<<(count + 15) / 16, 64, 0, st>>>(db->d_out, db->d_frames + first, count, db->d_status + first,
                                                  db->d_out_len + first);
    };
    // Stream plan.  k_huf only feeds k_exec, so it runs on the second stream, in the shadow of k_seq
    // (k_seq keeps ~2.7 KiB of LDS free per CU: a k_huf workgroup with small tables is co-resident):
    //   s  : k_init -> k_seq(head) -> k_seq(tail) -> [wait huf] k_exec(tail) -> [wait head done]
    //   s2 : [wait init] k_huf -> [wait k_seq(head)] k_exec(head)
    hipStream_t s2 = ctx->stream2;
    if (db->n_huf_tasks == 0 && db->n_seq_tasks == 0 && db->stats.n_blocks[2] == 0) {
        // Nothing but Raw / RLE blocks (BASELINE configs[1]): the pass IS the copy kernel -- no summaries to reset,
        // no second stream, no cross-stream events in front of it (they cost more than the 0.2 ms copy itself).
        // (two events, not nine: an event record costs the stream 3-5 us, and this pass is 0.2 ms)
        if (ev) {
            HIP_TRY(ctx, hipEventRecord(ev[4], s));
            ctx->run_split[ctx->runs] = 4;  // bit 2: a copy-only pass -- only ev[4], ev[5] (and ev[11] with checksums) were recorded
        }
        // (exec_variant 0: the chunked copy kernel; a forced variant: that kernel's own copy arms, as the parity tests want them)
        if (ctx->opt.exec_variant == 0 && !no_exec && db->n_frames > 0)
            // (a workgroup per chunk: 0.176 ms for the config's 4 096 frames; the same chunks walked with a stride by eight workgroups
            // per CU -- the shape of the kernel the copy ceiling is measured with -- 0.203 ms)
            k_copy_blocks<<<db->n_frames * kCopyChunksPerBlock, 256, 0, s>>>(
                db->d_in, db->d_out, db->d_frames, db->d_blocks, db->d_status, db->d_out_len, db->d_frame_order, 0u, db->n_frames);
        else
            launch_exec(s, 0, db->n_frames);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[5], s));
        launch_verify(s, 0, db->n_frames);
        if (ev) {
            if (ctx->opt.verify_checksum) HIP_TRY(ctx, hipEventRecord(ev[11], s));
            ctx->runs++;
        }
        HIP_TRY(ctx, hipGetLastError());
        return MZD_OK;
    }
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[0], s));
    if (db->n_blocks) k_init<<<(db->n_blocks + 255) / 256, 256, 0, s>>>(db->d_sums, db->n_blocks);
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], s));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_init_done, s));
    auto launch_huf = [&]() {
        if (!db->n_huf_tasks) return;
        // Which Huffman kernel: a lane per stream (k_huf) needs >= 64 streams per wavefront and many wavefronts per CU
        // to hide its ~190-cycle step; when the batch has few, long streams, a wavefront per stream decoding its
        // segments in parallel (k_huf_seg) is the one that fills the chip.
        const uint32_t hv = ctx->opt.huf_variant;
        const uint64_t streams = std::max<uint64_t>(db->stats.n_huf_streams, 1);
        // (... or many long streams under LARGE tables: a lane-per-stream wavefront keeps 16 tables in LDS, with tables of 2 048 cells
        // that is 64 KiB per wavefront and ~24 k streams in flight on the chip however many there are -- 392 GB/s whatever the streams'
        // length, against k_huf_seg's 1.2 TB/s: BASELINE config 3 at 65 536 frames 23.5 -> 7.3 ms, `profiles/r4_huf_crossover.txt`; with
        // tables of up to 32 cells -- config 4 -- the lane-per-stream kernel keeps its transposed bulk phase and many wavefronts per CU;
        // and with sequences to decode the Huffman kernel runs beside that stage, where k_huf_seg costs it more than it saves:
        // 8 192 x 1 MiB 26.3 -> 27.5 ms)
        // (round 5: with sequences in the batch the Huffman kernel runs before or beside the sequence stage, whose round is 1.7 ms
        // whatever the batch: a lane per stream is then one stream's latency, 0.4-0.7 ms from one wavefront per CU on, where the
        // wavefront-per-stream kernel beside that stage took 1.70 ms -- the longer of the two for the 8 192-frame shard of configs[4] --
        // and 0.66 against 0.45 ms in front of it at 16 384 frames; `profiles/r5_shard_huf.txt`)
        const uint64_t seg_below = 64ull * (db->n_seq_tasks ? 1 : 8) * (uint64_t)std::max(ctx->num_cus, 1);
        const bool seg = hv == 2 || hv == 4 || (hv == 0 && db->huf_out_bytes / streams >= 2048 &&
                                     (streams < seg_below || (db->huf_slot_cells > 32 && db->n_seq_tasks == 0)));
        const uint32_t seg_tbl = (uint32_t)(((size_t)db->huf_slot_cells * 2 + 15) & ~(size_t)15);
        (void)seg_tbl;
        // k_huf_w (round 6, mzd_huf_w.hip): k_huf_seg's method with whole lines between the CU and memory -- what `seg` means from
        // now on; huf_variant 2 keeps k_huf_seg itself alive for the parity tests, 4 forces k_huf_w wherever there are streams
        const bool hw = hv != 2;
        const uint32_t hw_tbl = (uint32_t)(((size_t)db->huf_slot_cells * 2 + 63) & ~(size_t)63);
        const size_t hw_lds = (size_t)hw_tbl + 4 * (size_t)kHwWaveBytes;
#ifdef MZD_TEST_KERNELS
        size_t seg_lds = (size_t)seg_tbl + kHufSegStripBytes;
        if (const char *e = exp_env("MZD_HUF_SEG_LDS")) seg_lds = std::max<size_t>(seg_lds, (size_t)atoi(e));  // experiment: residency cap
#endif
        const hipStream_t hs = huf_first ? s : s2;
        if (db->huf_sorted && !seg) {
            // one launch per table-size class: a wavefront's LDS is 16 tables of the CLASS's size, not of the batch's largest
            // (MaxBits 11 next to MaxBits 5: 64 KiB per wavefront for everybody otherwise), and its 64 streams are of similar length
            static const uint32_t kClassCells[3] = {32, 256, 2048};
            uint32_t q0 = 0;
            for (int c = 0; c < 3; c++) {
                const uint32_t q1 = db->huf_class_end[c];
                // the class's long streams first, a wavefront each (round 5, `profiles/r5_het_huf.txt`: the stage alone 3.05 -> 1.12 ms at 1 GiB
                // of real data, 3.9 -> 2.5 at 4 GiB; beside the sequence stage 3.2 -> 1.6 -- which is when the execution of a batch in two
                // groups of frames may start)
                if (q1 > q0 && (hv == 0 || hv == 4)) {
                    const uint32_t nl = hv == 4 ? q1 - q0 : std::min(db->huf_class_long[c], q1 - q0);  // (4: every stream through k_huf_w)
                    if (nl) {
                        const uint32_t cells = std::min(db->huf_slot_cells, kClassCells[c]);
                        const uint32_t tbl = (uint32_t)(((size_t)cells * 2 + 15) & ~(size_t)15);
                        (void)tbl;
                        const uint32_t wtbl = (uint32_t)(((size_t)cells * 2 + 63) & ~(size_t)63);
                        if (hw)
                            k_huf_w<<<nl, 256, (size_t)wtbl + 4 * (size_t)kHwWaveBytes, hs>>>(db->d_in, db->d_huf_tasks + 4 * (size_t)q0, 4 * nl, db->d_huf_entries,
                                                                                          db->d_litbuf, db->d_out, db->d_sums, wtbl);
#ifdef MZD_TEST_KERNELS
                        else
                        k_huf_seg<<<nl, 256, (size_t)tbl + kHufSegStripBytes, hs>>>(db->d_in, db->d_huf_tasks + 4 * (size_t)q0, 4 * nl, db->d_huf_entries,
                                                                                  db->d_litbuf, db->d_out, db->d_sums, tbl);
#endif
                        q0 += nl;
                    }
                }
                if (q1 > q0) {
                    const uint32_t cells = std::min(db->huf_slot_cells, kClassCells[c]), n = 4 * (q1 - q0);
                    const size_t lds = std::max<size_t>((size_t)kHufQuads * cells * 2, ctx->opt.huf_min_lds);
                    if (c == 0 && (hv == 0 || hv == 3) && n >= 64u * (uint32_t)std::max(ctx->num_cus, 1)) {
                        const uint32_t tstage = (uint32_t)((lds + 15) & ~(size_t)15);
                        k_huf<<<(n + 63) / 64, 64, tstage + kHufTStageBytes, hs>>>(db->d_in, db->d_huf_tasks + 4 * (size_t)q0, n, db->d_huf_entries,
                                                                                  db->d_litbuf, db->d_out, db->d_sums, cells, tstage);
                    } else {
                        k_huf<<<(n + 63) / 64, 64, lds, hs>>>(db->d_in, db->d_huf_tasks + 4 * (size_t)q0, n, db->d_huf_entries, db->d_litbuf,
                                                           db->d_out, db->d_sums, cells, 0u);
                    }
                }
                q0 = std::max(q0, q1);
            }
            return;
        }
        if (seg && hw)
            k_huf_w<<<db->n_huf_tasks / 4, 256, hw_lds, huf_first ? s : s2>>>(db->d_in, db->d_huf_tasks, db->n_huf_tasks, db->d_huf_entries,
                                                                          db->d_litbuf, db->d_out, db->d_sums, hw_tbl);
#ifdef MZD_TEST_KERNELS
        else if (seg)
            k_huf_seg<<<db->n_huf_tasks / 4, 256, seg_lds, huf_first ? s : s2>>>(db->d_in, db->d_huf_tasks, db->n_huf_tasks, db->d_huf_entries,
                                                               db->d_litbuf, db->d_out, db->d_sums, seg_tbl);
#endif
        else if (huf_first) {
            const uint32_t tstage = (uint32_t)((huf_lds + 15) & ~(size_t)15);
            k_huf<<<(db->n_huf_tasks + 63) / 64, 64, tstage + kHufTStageBytes, s>>>(db->d_in, db->d_huf_tasks, db->n_huf_tasks,
                                                                                   db->d_huf_entries, db->d_litbuf, db->d_out, db->d_sums,
                                                                                   db->huf_slot_cells, tstage);
        } else
            k_huf<<<(db->n_huf_tasks + 63) / 64, 64, huf_lds, s2>>>(db->d_in, db->d_huf_tasks, db->n_huf_tasks, db->d_huf_entries,
                                                                  db->d_litbuf, db->d_out, db->d_sums, db->huf_slot_cells, 0u);
    };
    // k_seq(head) is submitted FIRST so that its workgroups (nearly all of a CU's LDS each) claim the
    // CUs; k_huf's small workgroups then fill what is left instead of delaying them.  (Tried: k_huf beside the LAST,
    // partial round of the sequence stage instead of the first, k_exec in one launch after both: 22.3-22.5 ms against
    // 21.2-21.8 ms per pass.)
    // Two groups of frames (a heterogeneous batch whose sequence stage is as long as its longest chain; mzd_batch_upload): the frames that
    // hold the long chains are decoded and executed on a third stream, everything else on the caller's -- the execution of the other
    // frames runs beside the long chains instead of behind them (real data at 1 GiB: the sequence stage 4.9 ms for 0.8 ms of work, the
    // execution stage 4.9 ms, one after the other)
    //   s3 : [wait init] k_seq(long chains' frames) -> [wait huf] k_exec(those frames)
    //   s  : k_init -> k_seq(the others) -> [wait huf] k_exec(the others) -> [wait s3]
    //   s2 : [wait init] k_huf (all frames)
    const bool two_groups = db->long_tasks > 0 && db->long_tasks < db->n_seq_tasks && db->long_frames > 0 && db->long_frames < db->n_frames &&
                            db->seq_sorted && db->d_frame_order && q4 && pipe && exec_c && !blk && !split && !huf_first && !no_exec;
    if (two_groups) {
        db->last_pass |= MZD_PASS_TWO_GROUPS;
        if (!ctx->stream3) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
        hipStream_t s3 = ctx->stream3;
        HIP_TRY(ctx, hipStreamWaitEvent(s3, ctx->ev_init_done, 0));
        seq_pack = true;  // (full workgroups: the long chains on as few CUs as hold them)
        seq_stream = s3;
        // (the third stream's launches carry events of their own -- [6], [7] around its sequence kernel, [12], [13] around its execution
        // kernel -- and mzd_last_run_kernel_ms adds them to the stages' figures as it does for the head of a split pass: without them
        // both stages of the real-data workloads were under-reported by the long chains' share)
        if (ev) {
            ctx->run_split[ctx->runs] |= 8;  // bit 3: two groups of frames
            HIP_TRY(ctx, hipEventRecord(ev[6], s3));
        }
        launch_seq_tasks(0, db->long_tasks, true, 0);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[7], s3));
        seq_stream = s;
        seq_pack = false;
        launch_seq_tasks(db->long_tasks, db->n_seq_tasks - db->long_tasks, true, 0);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[3], s));
        HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_init_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[9], s2));
        launch_huf();
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[2], s2));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_huf_done, s2));
        HIP_TRY(ctx, hipStreamWaitEvent(s3, ctx->ev_huf_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[12], s3));
        launch_exec(s3, 0, db->long_frames);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[13], s3));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_head_done, s3));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_huf_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[4], s));
        launch_exec(s, db->long_frames, db->n_frames - db->long_frames);
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_head_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[5], s));
        launch_verify(s, 0, db->n_frames);
        if (ev) {
            HIP_TRY(ctx, hipEventRecord(ev[11], s));
            HIP_TRY(ctx, hipEventRecord(ev[8], s));
            ctx->runs++;
        }
        HIP_TRY(ctx, hipGetLastError());
        return MZD_OK;
    }
    if (huf_first) {
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[9], s));
        launch_huf();
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[2], s));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_huf_done, s));
        launch_seq(0, fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[3], s));
    } else {
        launch_seq(0, fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[3], s));
        HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_init_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[9], s2));
        launch_huf();
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[2], s2));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_huf_done, s2));
    }
    if (split) {
        seq_pack = true;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_head_ready, s));
        HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_head_ready, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[6], s2));
        launch_exec(s2, 0, fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[7], s2));
        launch_verify(s2, 0, fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[10], s2));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_head_done, s2));
        launch_seq(fA, db->n_frames);
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_huf_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[4], s));
        launch_exec(s, fA, db->n_frames - fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[5], s));
        launch_verify(s, fA, db->n_frames - fA);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[11], s));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_head_done, 0));  // the caller's stream sees the whole batch done
    } else {
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_huf_done, 0));
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[4], s));
        launch_exec(s, 0, db->n_frames);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[5], s));
        launch_verify(s, 0, db->n_frames);
        if (ev) HIP_TRY(ctx, hipEventRecord(ev[11], s));
    }
    if (ev) {
        HIP_TRY(ctx, hipEventRecord(ev[8], s));
        ctx->runs++;
    }
    HIP_TRY(ctx, hipGetLastError());
    return MZD_OK;
}

#ifdef MZD_Q4_STATS
extern "C" int mzd_debug_q4_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_q4_stats), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_q4_stats), z, sizeof z); }
    return 0;
}
#endif
#ifdef MZD_HUF_W_STATS
extern "C" int mzd_debug_huf_w_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_huf_w_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_huf_w_stats), z, sizeof z); }
    return 0;
}
#endif
#ifdef MZD_HUF_SEG_STATS
extern "C" int mzd_debug_huf_seg_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_huf_seg_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_huf_seg_stats), z, sizeof z); }
    return 0;
}
#endif
#ifdef MZD_XB_STATS
extern "C" int mzd_debug_xb_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_xb_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_xb_stats), z, sizeof z); }
    return 0;
}
#endif
#ifdef MZD_XC_STATS
extern "C" int mzd_debug_xc_stats(unsigned long long *out, int reset)
{
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_xc_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_xc_stats), z, sizeof z); }
    return 0;
}
#endif
#ifdef MZD_EXEC_STATS
extern "C" int mzd_debug_exec_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_exec_stats), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_exec_stats), z, sizeof z); }
    return 0;
}
#endif

#ifdef MZD_PIPE_STATS
extern "C" int mzd_debug_pipe_stats(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(mzd::g_pipe_stats), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(mzd::g_pipe_stats), z, sizeof z); }
    return 0;
}
#endif

int mzd_batch_read_fse_table(mzd_ctx *ctx, mzd_dbatch *db, uint32_t table, mzd_fse_entry *out, uint32_t cap)
{
    if (!ctx || !db || !out || (size_t)table + 1 >= db->fse_dev_off.size()) return -MZD_ERR_INVALID_ARG;
    if (db->trimmed) {  // (mzd_batch_trim freed the tables: say so instead of handing hipMemcpy a null base)
        ctx->last_error = "mzd_batch_read_fse_table on a batch that mzd_batch_trim has reduced to its output";
        return -MZD_ERR_INVALID_ARG;
    }
    const uint32_t first = db->fse_dev_off[table], n = db->fse_dev_off[table + 1] - first;
    if (n > cap) return -MZD_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return -MZD_ERR_DEVICE;
    if (n && hipMemcpy(out, db->d_fse_entries + first, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -MZD_ERR_DEVICE;
    return (int)n;
}

int mzd_batch_read_huf_table(mzd_ctx *ctx, mzd_dbatch *db, uint32_t table, mzd_huf_entry *out, uint32_t cap)
{
    if (!ctx || !db || !out || (size_t)table + 1 >= db->huf_dev_off.size()) return -MZD_ERR_INVALID_ARG;
    if (db->trimmed) {  // (mzd_batch_trim freed the tables: say so instead of handing hipMemcpy a null base)
        ctx->last_error = "mzd_batch_read_huf_table on a batch that mzd_batch_trim has reduced to its output";
        return -MZD_ERR_INVALID_ARG;
    }
    const uint32_t first = db->huf_dev_off[table], n = db->huf_dev_off[table + 1] - first;
    if (n > cap) return -MZD_ERR_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return -MZD_ERR_DEVICE;
    if (n && hipMemcpy(out, db->d_huf_entries + first, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) return -MZD_ERR_DEVICE;
    return (int)n;
}

int mzd_sync(mzd_ctx *ctx)
{
    if (!ctx) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    return MZD_OK;
}

int mzd_batch_download(mzd_ctx *ctx, mzd_dbatch *db, uint8_t *out_host, int32_t *status, uint64_t *out_len)
{
    if (!ctx || !db) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (out_host && db->out_size) HIP_TRY(ctx, hipMemcpy(out_host, db->d_out, db->out_size, hipMemcpyDeviceToHost));
    if (status && db->n_frames)
        HIP_TRY(ctx, hipMemcpy(status, db->d_status, db->n_frames * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out_len && db->n_frames)
        HIP_TRY(ctx, hipMemcpy(out_len, db->d_out_len, db->n_frames * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MZD_OK;
}

int mzd_batch_read_out(mzd_ctx *ctx, mzd_dbatch *db, uint64_t offset, uint8_t *dst, uint64_t nbytes)
{
    if (!ctx || !db || (nbytes && !dst) || offset > db->out_size || nbytes > db->out_size - offset) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (the batch's own pass, once -- not the device: a reader that drains a frame in many Reads must not stall the other
    // contexts and threads of the process with every one of them; ADVICE r4.  The pass ends on the stream it was given: the
    // library's second stream joins it.)
    if (db->run_pending) {
        HIP_TRY(ctx, hipStreamSynchronize(db->run_stream));
        db->run_pending = false;
    }
    if (nbytes) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, db->d_out + offset, nbytes, hipMemcpyDeviceToHost, ctx->stream2));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
    }
    return MZD_OK;
}

int mzd_batch_trim(mzd_ctx *ctx, mzd_dbatch *db)
{
    if (!ctx || !db) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (db->run_pending) {
        HIP_TRY(ctx, hipStreamSynchronize(db->run_stream));
        db->run_pending = false;
    }
    auto drop = [](auto *&p) {
        (void)hipFree((void *)p);
        p = nullptr;
    };
    drop(db->d_in_alloc);
    db->d_in = nullptr;
    drop(db->d_blocks);
    drop(db->d_sums);
    drop(db->d_huf_tasks);
    drop(db->d_seq_tasks);
    drop(db->d_fse_entries);
    drop(db->d_huf_entries);
    drop(db->d_recs);
    drop(db->d_tiles);
    drop(db->d_litbuf);
    drop(db->d_frame_order);
    drop(db->d_jobs);
    drop(db->d_heads);
    drop(db->d_bframes);
    drop(db->d_fixdone);
    drop(db->d_walk);
    drop(db->d_planes);
    drop(db->d_pat);
    {
        DevCaps keep{};
        keep.out = db->cap.out;
        keep.frames = db->cap.frames;
        keep.status = db->cap.status;
        keep.out_len = db->cap.out_len;
        db->cap = keep;
    }
    db->cap_jobs = db->cap_heads = db->cap_bframes = db->cap_planes = db->cap_pat = db->cap_fixdone = db->cap_walk = 0;
    db->n_recs = db->n_tiles = db->lit_bytes = 0;
    free_parse_temps(db->tmp);
    db->trimmed = true;
    return MZD_OK;
}

void *mzd_batch_device_out(mzd_dbatch *db) { return db ? db->d_out : nullptr; }
void *mzd_batch_device_status(mzd_dbatch *db) { return db ? db->d_status : nullptr; }
void *mzd_batch_device_out_len(mzd_dbatch *db) { return db ? db->d_out_len : nullptr; }

int mzd_decode_batch(mzd_ctx *ctx, const mzd_batch *batch, int32_t *status, uint64_t *out_len)
{
    if (!ctx || !batch) return MZD_ERR_INVALID_ARG;
    mzd_dbatch *db = nullptr;
    int rc = mzd_batch_upload(ctx, batch, &db);
    if (rc) return rc;
    rc = mzd_batch_run(ctx, db, nullptr);
    std::vector<int32_t> st(batch->n_frames);
    if (rc == MZD_OK) {
        uint8_t *host_out = (batch->flags & MZD_BATCH_OUT_ON_DEVICE) ? nullptr : batch->out;
        rc = mzd_batch_download(ctx, db, host_out, st.data(), out_len);
    }
    mzd_batch_free(ctx, db);
    if (rc) return rc;
    int first = MZD_OK;
    for (uint32_t i = 0; i < batch->n_frames; i++) {
        if (status) status[i] = st[i];
        if (st[i] && !first) first = st[i];
    }
    return first;
}

int mzd_last_run_kernel_ms(mzd_ctx *ctx, const char **names, float *ms, int cap)
{
    // k_seq / k_exec: sum of their launches (head + tail of a split batch; those overlap in time);
    // "path": first event to last completion of the whole hot path (what the roofline divides by)
    static const char *kNames[6] = {"k_init", "k_huf", "k_seq", "k_exec", "path", "k_xxh64"};
    if (!ctx || ctx->runs == 0) return 0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    size_t cnt = 0;
    auto el = [](hipEvent_t a, hipEvent_t b) {
        float t = 0;
        return hipEventElapsedTime(&t, a, b) == hipSuccess ? (double)t : 0.0;
    };
    for (size_t r = 0; r < ctx->runs; r++) {
        hipEvent_t *e = ctx->ev.data() + r * kEvPerRun;
        const bool split = ctx->run_split[r] & 1, huf_first = ctx->run_split[r] & 2, two = ctx->run_split[r] & 8;
        if (ctx->run_split[r] & 4) {  // a copy-only pass (Raw / RLE blocks only)
            acc[3] += el(e[4], e[5]);
            acc[4] += el(e[4], ctx->opt.verify_checksum ? e[11] : e[5]);
            if (ctx->opt.verify_checksum) acc[5] += el(e[5], e[11]);
            cnt++;
            continue;
        }
        acc[0] += el(e[0], e[1]);
        acc[1] += el(e[9], e[2]);                                           // k_huf (second stream, or first on the caller's)
        // k_seq head (+ tail, incl. its wait for k_huf); with k_huf first on the same stream the head starts at ITS end
        acc[2] += el(huf_first ? e[2] : e[1], e[3]) + (split ? el(e[3], e[4]) : 0.0) + (two ? el(e[6], e[7]) : 0.0);
        acc[3] += el(e[4], e[5]) + (split ? el(e[6], e[7]) : 0.0) + (two ? el(e[12], e[13]) : 0.0);
        acc[4] += el(e[0], e[8]);
        if (ctx->opt.verify_checksum) acc[5] += el(e[5], e[11]) + (split ? el(e[7], e[10]) : 0.0);
        cnt++;
    }
    int n = 0;
    for (int i = 0; i < (ctx->opt.verify_checksum ? 6 : 5) && n < cap; i++, n++) {
        if (names) names[n] = kNames[i];
        if (ms) ms[n] = (float)(acc[i] / cnt);
    }
    return n;
}

void mzd_timing_reset(mzd_ctx *ctx, int enable)
{
    if (!ctx) return;
    ctx->runs = 0;
    ctx->timing = enable != 0;
}

int mzd_measure_copy(mzd_ctx *ctx, uint64_t read_bytes, uint64_t write_bytes, int iters, float *ms)
{
    if (!ctx || !ms || iters < 1 || write_bytes < 16) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n_read = read_bytes / 16, n_write = write_bytes / 16;  // both streams in full, whichever is longer
    u32x4 *src = nullptr, *dst = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&src, std::max<uint64_t>(n_read, 1) * 16));
    hipError_t e = hipMalloc((void **)&dst, n_write * 16);
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (e == hipSuccess) e = hipMemsetAsync(src, 0x5A, std::max<uint64_t>(n_read, 1) * 16, ctx->stream);
    if (e == hipSuccess) e = hipEventCreate(&t0);
    if (e == hipSuccess) e = hipEventCreate(&t1);
    // grid: 8 workgroups of 256 threads per CU, the rest grid-stride
    const uint32_t grid = (uint32_t)std::min<uint64_t>((std::max(n_read, n_write) + 255) / 256, (uint64_t)std::max(ctx->num_cus, 1) * 8);
    if (e == hipSuccess) {
        k_copy_ceiling<<<grid, 256, 0, ctx->stream>>>(src, dst, n_read, n_write);  // warm-up
        e = hipEventRecord(t0, ctx->stream);
        for (int i = 0; i < iters; i++) k_copy_ceiling<<<grid, 256, 0, ctx->stream>>>(src, dst, n_read, n_write);
        if (e == hipSuccess) e = hipEventRecord(t1, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        float t = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, t0, t1);
        *ms = t / (float)iters;
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    (void)hipFree(src);
    (void)hipFree(dst);
    HIP_TRY(ctx, e);
    return MZD_OK;
}

static_assert(sizeof(mzd_debug_block) == sizeof(DBlock) && offsetof(mzd_debug_block, tile_off) == offsetof(DBlock, tile_off) &&
              offsetof(mzd_debug_block, lit_type) == offsetof(DBlock, lit_type), "mzd_debug_block mirrors DBlock");

int mzd_batch_debug_read(mzd_ctx *ctx, mzd_dbatch *db, int what, uint64_t offset, void *dst, uint64_t bytes)
{
    if (!ctx || !db || (!dst && bytes)) return MZD_ERR_INVALID_ARG;
    const void *base = nullptr;
    uint64_t size = 0;
    switch (what) {
    case MZD_DEBUG_LITERALS: base = db->d_litbuf; size = db->lit_bytes; break;
    case MZD_DEBUG_RECORDS: base = db->d_recs; size = db->n_recs * 8; break;
    case MZD_DEBUG_TILES: base = db->d_tiles; size = db->n_tiles * sizeof(TileBase); break;
    case MZD_DEBUG_BLOCKS: base = db->d_blocks; size = (uint64_t)db->n_blocks * sizeof(DBlock); break;
    default: return MZD_ERR_INVALID_ARG;
    }
    if (db->trimmed || offset > size || bytes > size - offset) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (bytes) HIP_TRY(ctx, hipMemcpy(dst, (const uint8_t *)base + offset, bytes, hipMemcpyDeviceToHost));
    return MZD_OK;
}

uint32_t mzd_batch_last_pass(const mzd_dbatch *db) { return db ? db->last_pass : 0u; }

int mzd_debug_plan_unit_bytes(mzd_ctx *ctx, uint64_t frame_bytes)
{
    if (!ctx) return MZD_ERR_INVALID_ARG;
    ctx->test_large_frame = frame_bytes;
    return MZD_OK;
}

int mzd_debug_force_fixup_bail(mzd_ctx *ctx, uint32_t step)
{
    if (!ctx) return MZD_ERR_INVALID_ARG;
    ctx->test_fixup_bail = step;
    return MZD_OK;
}

int mzd_debug_backbits(mzd_ctx *ctx, const uint8_t *stream, uint32_t len, const uint8_t *nbits, uint32_t n_reads,
                       uint64_t *values, int64_t *bits_still)
{
    if (!ctx || (!stream && len) || !nbits || !values || !bits_still || !n_reads) return MZD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_reads; i++)
        if (nbits[i] > 32) return MZD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t kPad = 16;  // the reader's window loads reach 8 bytes below the stream and 8 above its end
    uint8_t *d = nullptr;
    const size_t sz = (size_t)kPad + len + kPad + n_reads + 8 + (size_t)n_reads * 16;
    HIP_TRY(ctx, hipMalloc((void **)&d, sz));
    uint8_t *d_stream = d + kPad, *d_nb = d_stream + len + kPad;
    uint64_t *d_val = (uint64_t *)(d + ((kPad + len + kPad + n_reads + 7) & ~(size_t)7));
    int64_t *d_left = (int64_t *)(d_val + n_reads);
    hipError_t e = hipMemset(d, 0xEE, sz);  // slack bytes are NOT zero: the reader itself must mask below the start
    if (e == hipSuccess && len) e = hipMemcpy(d_stream, stream, len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_nb, nbits, n_reads, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_test_backbits<<<1, 64, 0, ctx->stream>>>(d_stream, len, d_nb, n_reads, d_val, d_left);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(values, d_val, (size_t)n_reads * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(bits_still, d_left, (size_t)n_reads * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(ctx, e);
    return MZD_OK;
}

int mzd_batch_get_stats(mzd_dbatch *db, mzd_batch_stats *st)
{
    if (!db || !st) return MZD_ERR_INVALID_ARG;
    *st = db->stats;
    st->n_fse_built = db->n_fse_built;
    st->n_huf_built = db->n_huf_built;
    st->fse_build_ms = db->fse_build_ms;
    st->parse_ms = db->parse_ms;
    return MZD_OK;
}

// ---- one frame in chunks (ABI 9; include/mzd.h): the cursor (planner.cpp) + the window and the offset history kept on the device.
// Two slabs of (window + chunk) bytes.  A chunk is decoded into the slab whose beginning holds the bytes the frame regenerated
// before it (`keep` of them: all, until there are more than the window); its own bytes go to the caller; when slab and window no
// longer hold what is there, the last window_size bytes move to the other slab's beginning (ringbuffer.go:36-49 keeps as much).
//
// A call is one step of a two-stage pipeline: it describes chunk i on the host WHILE chunk i - 1 is on the device, waits for that
// one (its length is where chunk i starts, its history what chunk i starts with), launches chunk i, and copies chunk i - 1 out
// while chunk i runs: the bytes a call produces are those of the chunk the call before consumed.
struct mzd_fstream {
    mzd_ctx *ctx = nullptr;
    mzd_cursor *cur = nullptr;
    uint64_t chunk_out = 0;
    uint8_t *slab[2] = {nullptr, nullptr};
    uint64_t slab_bytes = 0;
    int at = 0;         // the slab the next chunk goes to
    uint64_t keep = 0;  // bytes of the frame at its beginning
    uint64_t window = 0;
    int32_t hist[3] = {1, 4, 8};  // framedecompressor.go:48,59
    uint64_t total = 0;
    bool sized = false, planned_last = false, done = false;
    int status = MZD_OK;
    // the chunk on the device
    mzd_dbatch *fly = nullptr;
    uint64_t fly_keep = 0, fly_bound = 0;
    int fly_at = 0;
    bool fly_last = false;
    mzd_dbatch *spent = nullptr;  // the chunk before it: its scratch is freed when the device is idle anyway (hipFree waits for it)
    hipStream_t s_copy = nullptr;
    double ms[4] = {0, 0, 0, 0};  // mzd_fstream_timing: cursor, upload + launch, wait + free, copy-out
};

int mzd_fstream_open(mzd_ctx *ctx, uint64_t chunk_out, mzd_fstream **out)
{
    if (!ctx || !out) return MZD_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mzd_fstream *fs = new mzd_fstream();
    fs->ctx = ctx;
    if (hipStreamCreateWithFlags(&fs->s_copy, hipStreamNonBlocking) != hipSuccess) {
        delete fs;
        ctx->last_error = "mzd_fstream_open: hipStreamCreateWithFlags failed";
        return MZD_ERR_DEVICE;
    }
    fs->cur = mzd_cursor_create();
    fs->chunk_out = std::max<uint64_t>(chunk_out ? chunk_out : (64ull << 20), kBlockMax);
    *out = fs;
    return MZD_OK;
}

void mzd_fstream_close(mzd_fstream *fs)
{
    if (!fs) return;
    if (fs->ctx) (void)hipSetDevice(fs->ctx->device);
    if (fs->fly) {
        (void)hipStreamSynchronize(fs->ctx->stream);
        mzd_batch_free(fs->ctx, fs->fly);
    }
    if (fs->spent) mzd_batch_free(fs->ctx, fs->spent);
    if (fs->s_copy) (void)hipStreamDestroy(fs->s_copy);
    (void)hipFree(fs->slab[0]);
    (void)hipFree(fs->slab[1]);
    mzd_cursor_destroy(fs->cur);
    delete fs;
}

void mzd_fstream_set_threads(mzd_fstream *fs, uint32_t n_threads)
{
    if (fs) mzd_cursor_set_threads(fs->cur, n_threads);
}
uint64_t mzd_fstream_total_out(const mzd_fstream *fs) { return fs ? fs->total : 0; }
const mzd_cursor *mzd_fstream_cursor(const mzd_fstream *fs) { return fs ? fs->cur : nullptr; }

int mzd_fstream_next(mzd_fstream *fs, const uint8_t *src, uint64_t len, uint8_t *dst, uint64_t dst_cap, uint64_t *consumed,
                     uint64_t *produced, int *done)
{
    if (!fs || (!src && len) || !dst || !consumed || !produced || dst_cap < kBlockMax) return MZD_ERR_INVALID_ARG;
    *consumed = *produced = 0;
    if (done) *done = fs->done ? 1 : 0;
    if (fs->status) return fs->status;
    if (fs->done) return MZD_ERR_OUT_OF_BLOCKS;  // framedecompressor.go:196
    if (fs->fly && dst_cap < fs->fly_bound) return MZD_ERR_INVALID_ARG;  // (the chunk on the device was sized for the call before's dst)
    mzd_ctx *ctx = fs->ctx;
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    auto lap = [&](int k) {
        const auto t1 = clk::now();
        fs->ms[k] += std::chrono::duration<double, std::milli>(t1 - t0).count();
        t0 = t1;
    };
    auto fail = [&](int code) {
        fs->status = code;
        return code;
    };
    // ---- the next chunk's description (host), while the chunk before it is on the device.  Where it starts and with which history
    // is known when that one is done: both are put in below.
    const uint64_t max_out = std::min(fs->chunk_out, dst_cap);
    const mzd_batch *chunk = nullptr;
    int last = 0;
    if (!fs->planned_last) {
        const int rc = mzd_cursor_next(fs->cur, src, len, max_out, 0, nullptr, consumed, &chunk, &last);
        if (rc) return fail(rc);
        if (chunk && last) fs->planned_last = true;
    }
    lap(0);
    if (!chunk && !fs->fly) return MZD_OK;  // no whole block in src yet, nothing on the device
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ---- the chunk on the device: wait for it; its length and the history behind it
    uint64_t n = 0;
    const bool had = fs->fly != nullptr;
    if (had) {
        mzd_dbatch *db = fs->fly;
        int32_t st = MZD_OK;
        uint64_t olen = 0;
        hipError_t e = hipMemcpyAsync(&st, db->d_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&olen, db->d_out_len, sizeof(olen), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && db->d_frame_hist) e = hipMemcpyAsync(fs->hist, db->d_frame_hist, sizeof(fs->hist), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        db->run_pending = false;
        if (e != hipSuccess) {
            ctx->last_error = std::string("mzd_fstream_next: waiting for the chunk failed: ") + hipGetErrorString(e);
            return fail(MZD_ERR_DEVICE);
        }
        if (st) return fail(st);
        if (olen < fs->fly_keep || olen - fs->fly_keep > dst_cap) return fail(MZD_ERR_DST_FULL);
        n = olen - fs->fly_keep;
        // what the next chunk finds in front of it
        const uint64_t have = fs->fly_keep + n;
        if (fs->fly_last) {
            // (the frame's last chunk: nothing follows it)
        } else if (have <= fs->window && ((have + 255) & ~255ull) + fs->chunk_out + 1024 <= fs->slab_bytes) {
            fs->keep = have;  // (the frame so far is within its window and the slab has room behind it: it stays where it is)
        } else {
            const uint64_t nk = std::min(fs->window, have);
            if (nk) HIP_TRY(ctx, hipMemcpyAsync(fs->slab[fs->fly_at ^ 1], fs->slab[fs->fly_at] + (have - nk), nk, hipMemcpyDeviceToDevice, ctx->stream));
            fs->at = fs->fly_at ^ 1;
            fs->keep = nk;
        }
        if (fs->spent) {
            mzd_batch_free(ctx, fs->spent);
            fs->spent = nullptr;
        }
    }
    lap(2);
    // ---- the bytes of the chunk that is done start on their way out (its own stream: the link carries both directions at once, and
    // the next chunk's pass runs beside the copy)
    if (had && n) {
        const hipError_t e = hipMemcpyAsync(dst, fs->slab[fs->fly_at] + fs->fly_keep, n, hipMemcpyDeviceToHost, fs->s_copy);
        if (e != hipSuccess) {
            ctx->last_error = std::string("mzd_fstream_next: the chunk's copy-out failed: ") + hipGetErrorString(e);
            return fail(MZD_ERR_DEVICE);
        }
    }
    // ---- the next chunk goes to the device
    mzd_dbatch *next_db = nullptr;
    uint64_t next_bound = 0;
    if (chunk) {
        if (!fs->sized) {
            fs->window = mzd_cursor_window(fs->cur);
            // (a slab's positions are 32-bit and block mode takes slabs below 2 GiB)
            if (fs->window > (1ull << 31) - fs->chunk_out - (1ull << 20)) {
                ctx->last_error = "mzd_fstream: the frame's window and a chunk do not fit a slab of 2 GiB";
                return fail(MZD_ERR_UNSUPPORTED);
            }
            fs->slab_bytes = ((fs->window + 255) & ~255ull) + fs->chunk_out + 1024;
            for (int k = 0; k < 2; k++) {
                const hipError_t e = hipMalloc((void **)&fs->slab[k], fs->slab_bytes);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->last_error = std::string("mzd_fstream: hipMalloc of a slab failed: ") + hipGetErrorString(e);
                    return fail(MZD_ERR_DEVICE);
                }
            }
            fs->sized = true;
        }
        mzd_frame_desc fd = chunk->frames[0];
        next_bound = fd.out_capacity;  // (described from position 0: the blocks' bound)
        fd.start = fs->keep;
        fd.out_capacity = fs->keep + next_bound;
        for (int k = 0; k < 3; k++) fd.hist[k] = fs->hist[k];
        mzd_batch b = *chunk;
        b.frames = &fd;
        b.out = fs->slab[fs->at];
        b.out_size = fs->slab_bytes;
        b.flags |= MZD_BATCH_OUT_ON_DEVICE;
        int rc = mzd_batch_upload(ctx, &b, &next_db);
        if (rc == MZD_OK) rc = mzd_batch_run(ctx, next_db, nullptr);
        if (rc) {
            (void)hipStreamSynchronize(fs->s_copy);
            if (next_db) mzd_batch_free(ctx, next_db);
            return fail(rc);
        }
    }
    lap(1);
    // ---- ... and have arrived
    if (had) {
        const hipError_t e = hipStreamSynchronize(fs->s_copy);
        if (e != hipSuccess) {
            ctx->last_error = std::string("mzd_fstream_next: the chunk's copy-out failed: ") + hipGetErrorString(e);
            if (next_db) {
                (void)hipStreamSynchronize(ctx->stream);
                mzd_batch_free(ctx, next_db);
            }
            return fail(MZD_ERR_DEVICE);
        }
        fs->spent = fs->fly;
        fs->total += n;
        *produced = n;
        if (fs->fly_last) {
            fs->done = true;
            if (done) *done = 1;
            mzd_batch_free(ctx, fs->spent);
            fs->spent = nullptr;
            fs->fly = nullptr;
            const uint64_t content = mzd_cursor_content_size(fs->cur);
            if (content != MZD_UNKNOWN_SIZE && content != fs->total) return fail(MZD_ERR_DST_FULL);
            lap(3);
            return MZD_OK;
        }
    }
    lap(3);
    fs->fly = next_db;
    fs->fly_keep = fs->keep;
    fs->fly_bound = next_bound;
    fs->fly_at = fs->at;
    fs->fly_last = last != 0;
    return MZD_OK;
}

int mzd_fstream_timing(const mzd_fstream *fs, double *ms, int cap)
{
    if (!fs || !ms) return 0;
    const int n = std::max(0, std::min(cap, 4));
    for (int i = 0; i < n; i++) ms[i] = fs->ms[i];
    return n;
}

}  // extern "C"
