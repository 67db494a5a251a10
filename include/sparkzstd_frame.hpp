// sparkzstd_frame.hpp -- C++ host-side mirror of sparkzstd's public API over the mzd C ABI.
//
// Same names, argument meaning and error behaviour as the Go originals:
//   FrameReader        decompression/framereader.go:9-109      (io.Reader over one frame)
//   FrameDecompressor  decompression/framedecompressor.go:14-374 (source -> target pipe)
//   DecodeFrames       the batch entry the cgo shim adds
// The Go versions decode one block per call on the CPU.  Here the host plans the whole frame
// (mzd_plan_*: headers + tables) and ONE device batch regenerates it the first time output is
// needed; what the caller observes (bytes, short reads, EOF, sentinel errors) is the same.
// Header-only; link with libmzd.so.
#pragma once
#include <cstdint>
#include <condition_variable>
#include <cstring>
#include <istream>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "mzd.h"

namespace sparkzstd {

// Mirrors the reference's sentinel errors (errors.New values): code() is the MZD_ERR_* that maps to
// the Go sentinel listed next to it in mzd.h.
class Error : public std::runtime_error {
  public:
    Error(int code, const std::string &where) : std::runtime_error(where + ": " + mzd_strerror(code)), code_(code) {}
    int code() const { return code_; }

  private:
    int code_;
};

inline mzd_ctx *default_context()
{
    static mzd_ctx *ctx = [] {
        int err = 0;
        mzd_ctx *c = mzd_create(0, nullptr, &err);
        if (!c) throw Error(err, "mzd_create");  // no GPU: there is no CPU fallback
        return c;
    }();
    return ctx;
}

// Where the frame / block / section headers are parsed: false (default) = host planner (mzd_plan_*),
// true = on the device (mzd_batch_upload_frames, k_parse).  Same statuses and bytes either way.
inline bool &DevicePlanning()
{
    static bool on = false;
    return on;
}

// DecodeFrames([][]byte) ([][]byte, []error): many independent frames in one device batch.
inline std::vector<std::vector<uint8_t>> DecodeFrames(const std::vector<std::vector<uint8_t>> &frames,
                                                      std::vector<int> *status = nullptr, mzd_ctx *ctx = nullptr)
{
    if (!ctx) ctx = default_context();
    std::vector<std::vector<uint8_t>> res(frames.size());
    std::vector<int> st(frames.size(), MZD_OK);
    if (DevicePlanning()) {
        std::vector<uint8_t> blob;
        std::vector<uint64_t> off(frames.size()), ln(frames.size());
        for (size_t i = 0; i < frames.size(); i++) {
            off[i] = blob.size();
            ln[i] = frames[i].size();
            blob.insert(blob.end(), frames[i].begin(), frames[i].end());
        }
        mzd_dbatch *db = nullptr;
        int rc = mzd_batch_upload_frames(ctx, blob.data(), blob.size(), 0, off.data(), ln.data(), (uint32_t)frames.size(), nullptr, 0, &db);
        if (rc == MZD_OK) rc = mzd_batch_run(ctx, db, nullptr);
        std::vector<uint8_t> out(db ? mzd_batch_out_size(db) : 0);
        std::vector<int32_t> dst(frames.size());
        std::vector<uint64_t> len(frames.size()), slab(frames.size());
        if (rc == MZD_OK) rc = mzd_batch_download(ctx, db, out.data(), dst.data(), len.data());
        if (rc == MZD_OK) rc = mzd_batch_frame_layout(db, slab.data(), nullptr);
        mzd_batch_free(ctx, db);
        if (rc != MZD_OK) throw Error(rc, std::string("mzd_batch_upload_frames (") + mzd_last_error(ctx) + ")");
        for (size_t i = 0; i < frames.size(); i++) {
            st[i] = dst[i];
            if (st[i] == MZD_OK) res[i].assign(out.data() + slab[i], out.data() + slab[i] + len[i]);
        }
        if (status) *status = st;
        return res;
    }
    mzd_plan *plan = mzd_plan_create();
    mzd_plan_set_device_tables(plan, 1);  // FSE tables travel as normalised counts and are built on the device
    for (size_t i = 0; i < frames.size(); i++)
        st[i] = mzd_plan_add_frame(plan, frames[i].data(), frames[i].size(), nullptr);
    const mzd_batch *b = mzd_plan_finalize(plan);
    mzd_batch run = *b;
    std::vector<uint8_t> out(b->out_size);
    run.out = out.data();
    std::vector<int32_t> dst(frames.size());
    std::vector<uint64_t> len(frames.size());
    int rc = mzd_decode_batch(ctx, &run, dst.data(), len.data());
    if (rc >= MZD_ERR_DEVICE) {
        mzd_plan_destroy(plan);
        throw Error(rc, std::string("mzd_decode_batch (") + mzd_last_error(ctx) + ")");
    }
    for (size_t i = 0; i < frames.size(); i++) {
        if (st[i] == MZD_OK) st[i] = dst[i];
        if (st[i] == MZD_OK) {
            const uint8_t *p = out.data() + b->frames[i].out_offset;
            res[i].assign(p, p + len[i]);
        }
    }
    mzd_plan_destroy(plan);
    if (status) *status = st;
    return res;
}

// ---- one batch over several GPUs of a node.  Frames share nothing (tables, offset history and window are per frame:
// framedecompressor.go:42-52), so device r takes a contiguous range of them -- equal counts when the frames cost the same,
// equal C + D (compressed + declared decompressed bytes, frame.go:23-61) otherwise -- on its own context and host thread,
// and the results are stitched in frame order.  No collective, no RCCL.

// context k of `device` (a device listed twice gets two contexts: two independent streams on one GPU)
inline mzd_ctx *device_context(int device, int k = 0)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, mzd_ctx *> pool;
    std::lock_guard<std::mutex> g(mu);
    auto it = pool.find({device, k});
    if (it != pool.end()) return it->second;
    int err = 0;
    mzd_ctx *c = mzd_create(device, nullptr, &err);
    if (!c) throw Error(err, "mzd_create");
    pool[{device, k}] = c;
    return c;
}

// C + D of one frame from its header alone; a frame without a declared content size counts with its window
inline uint64_t DeclaredFrameCost(const std::vector<uint8_t> &f)
{
    const uint64_t n = f.size();
    static const unsigned char magic[4] = {0x28, 0xB5, 0x2F, 0xFD};
    if (n < 6 || std::memcmp(f.data(), magic, 4) != 0) return n;
    const unsigned fhd = f[4], fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    size_t pos = 5;
    uint64_t window = 128 * 1024;
    if (!single) {
        const unsigned wd = f[pos++];
        const uint64_t base = 1ull << (10 + (wd >> 3));
        window = base + (base >> 3) * (wd & 7);
    }
    static const unsigned did_bytes[4] = {0, 1, 2, 4};
    pos += did_bytes[did];
    const unsigned fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (fcs_bytes == 0 || pos + fcs_bytes > n) return n + window;
    uint64_t d = 0;
    for (unsigned i = 0; i < fcs_bytes; i++) d |= (uint64_t)f[pos + i] << (8 * i);
    return n + d + (fcs_bytes == 2 ? 256 : 0);
}

// the frame range [lo, hi) of each of `world` devices
inline std::vector<std::pair<size_t, size_t>> ShardFrames(const std::vector<std::vector<uint8_t>> &frames, size_t world)
{
    std::vector<std::pair<size_t, size_t>> r(world);
    const size_t n = frames.size();
    std::vector<uint64_t> cost(n);
    bool same = true;
    double total = 0;
    for (size_t i = 0; i < n; i++) {
        cost[i] = DeclaredFrameCost(frames[i]);
        same = same && cost[i] == cost[0];
        total += (double)cost[i];
    }
    if (same) {
        const size_t base = world ? n / world : 0, extra = world ? n % world : 0;
        for (size_t k = 0, lo = 0; k < world; k++) {
            const size_t hi = lo + base + (k < extra ? 1 : 0);
            r[k] = {lo, hi};
            lo = hi;
        }
        return r;
    }
    double acc = 0, target = 0;
    size_t lo = 0;
    for (size_t k = 0; k < world; k++) {
        target += total / (double)world;
        size_t hi = lo;
        while (hi < n && (acc + (double)cost[hi] <= target || hi == lo) && (n - hi) > (world - k - 1)) acc += (double)cost[hi++];
        if (k == world - 1) hi = n;
        r[k] = {lo, hi};
        lo = hi;
    }
    return r;
}

// DecodeFramesOn(devices []int, frames [][]byte) ([][]byte, []error)
inline std::vector<std::vector<uint8_t>> DecodeFramesOn(const std::vector<int> &devices, const std::vector<std::vector<uint8_t>> &frames,
                                                        std::vector<int> *status = nullptr)
{
    if (devices.empty()) return DecodeFrames(frames, status, nullptr);
    std::vector<mzd_ctx *> ctxs;
    std::map<int, int> seen;
    for (int d : devices) ctxs.push_back(device_context(d, seen[d]++));
    const auto ranges = ShardFrames(frames, devices.size());
    std::vector<std::vector<uint8_t>> res(frames.size());
    std::vector<int> st(frames.size(), MZD_OK);
    std::vector<std::string> fail(devices.size());
    std::vector<int> failcode(devices.size(), MZD_OK);
    std::vector<std::thread> th;
    for (size_t k = 0; k < devices.size(); k++) {
        const size_t lo = ranges[k].first, hi = ranges[k].second;
        if (hi <= lo) continue;
        th.emplace_back([&, k, lo, hi] {
            try {
                std::vector<std::vector<uint8_t>> part(frames.begin() + (long)lo, frames.begin() + (long)hi);
                std::vector<int> pst;
                auto out = DecodeFrames(part, &pst, ctxs[k]);
                for (size_t i = lo; i < hi; i++) {
                    res[i] = std::move(out[i - lo]);
                    st[i] = pst[i - lo];
                }
            } catch (const Error &e) {
                failcode[k] = e.code();
                fail[k] = e.what();
            }
        });
    }
    for (auto &t : th) t.join();
    for (size_t k = 0; k < devices.size(); k++)
        if (failcode[k] != MZD_OK) throw Error(failcode[k], "DecodeFramesOn device " + std::to_string(devices[k]) + ": " + fail[k]);
    if (status) *status = st;
    return res;
}

// NewFrameReader(source io.Reader) -- framereader.go:17
class FrameReader {
  public:
    // chunk_bytes > 0 (ABI 9): the frame goes through the device in CHUNKS of whole blocks that regenerate up to so many bytes each
    // (mzd_fstream_*): the source is read piece by piece as the chunks need it, Read returns bytes as soon as the chunk that holds
    // them is decoded, and the device keeps the frame's WINDOW between two chunks, not the frame -- the reference's own shape
    // (framereader.go:51-109 over DecodeNextBlock and a ring of the window's size), for frames larger than the device's memory and
    // for sources that arrive slowly.  0: the frame is read and decoded whole at the first Read (faster for a frame that fits).
    explicit FrameReader(std::istream *source = nullptr, mzd_ctx *ctx = nullptr, uint64_t chunk_bytes = 0)
        : ctx_(ctx), chunk_bytes_(chunk_bytes ? std::max<uint64_t>(chunk_bytes, 128 * 1024) : 0)
    {
        if (source) Reset(source);
    }
    FrameReader(const FrameReader &) = delete;
    FrameReader &operator=(const FrameReader &) = delete;
    ~FrameReader()
    {
        release();
        if (cbuf_) mzd_host_free(cbuf_);
    }
    void Reset(std::istream *source)  // framereader.go:35-49: magic number + frame header are checked here
    {
        release();
        pend_.clear();
        pend_lo_ = clo_ = chi_ = 0;
        pos_ = len_ = off_ = 0;
        decoded_ = false;
        source_ = source;
        head_.clear();
        if (source) {
            char m[4];
            source->read(m, 4);
            head_.assign(m, m + source->gcount());
            if (head_.size() < 4) throw Error(MZD_ERR_TRUNCATED, "NewFrameReader");
            static const unsigned char magic[4] = {0x28, 0xB5, 0x2F, 0xFD};
            if (std::memcmp(head_.data(), magic, 4) != 0) throw Error(MZD_ERR_MAGIC, "NewFrameReader");
        }
    }
    // Read(p []byte) (int, error) -- framereader.go:51-109: up to n bytes; 0 == io.EOF (only after the
    // last block has been drained).  The frame is decoded at the first Read and stays in HBM: a Read of 1 MiB or more is ONE
    // copy from there into p (mzd_batch_read_out), smaller ones come out of a 4 MiB window fetched the same way; the device
    // memory is released with the last byte.
    size_t Read(uint8_t *p, size_t n)
    {
        if (chunk_bytes_) {
            if (clo_ >= chi_ && !next_chunk()) return 0;
            n = (size_t)std::min<uint64_t>(n, chi_ - clo_);
            std::memcpy(p, cbuf_ + clo_, n);
            clo_ += n;
            readTotal_ += n;
            return n;
        }
        if (!decoded_) decode();
        n = std::min<uint64_t>(n, len_ - pos_);
        if (n == 0) return 0;
        const bool in_window = pos_ >= win_lo_ && pos_ < win_hi_;
        if (n >= kDirect && !in_window) {
            const int rc = mzd_batch_read_out(ctx_, db_, off_ + pos_, p, n);
            if (rc != MZD_OK) throw Error(rc, std::string("mzd_batch_read_out (") + mzd_last_error(ctx_) + ")");
        } else {
            if (!in_window) {
                window_.resize((size_t)std::min<uint64_t>(kWindow, len_));
                win_lo_ = pos_;
                win_hi_ = std::min<uint64_t>(pos_ + window_.size(), len_);
                const int rc = mzd_batch_read_out(ctx_, db_, off_ + win_lo_, window_.data(), win_hi_ - win_lo_);
                if (rc != MZD_OK) throw Error(rc, std::string("mzd_batch_read_out (") + mzd_last_error(ctx_) + ")");
            }
            n = std::min<uint64_t>(n, win_hi_ - pos_);
            std::memcpy(p, window_.data() + (pos_ - win_lo_), n);
        }
        pos_ += n;
        readTotal_ += n;
        if (pos_ >= len_) release();
        return n;
    }
    // (for FrameDecompressor: a source whose magic number CheckMagicnum has read and checked already)
    void ResetBehindMagic(std::istream *source, const std::vector<char> &head)
    {
        Reset(nullptr);
        source_ = source;
        head_ = head;
    }
    // chunk mode: the frame's last block has been decoded and all of its bytes handed out
    bool Finished() const { return decoded_ && clo_ >= chi_; }
    // chunk mode: the chunk at the reader's position in place -- its bytes not yet read, valid until the next Read / View / Reset
    // (DecodeNextBlock's hand-over without Read's copy); {nullptr, 0} behind the frame's last block
    std::pair<const uint8_t *, size_t> ViewChunk()
    {
        if (!chunk_bytes_) throw Error(MZD_ERR_INVALID_ARG, "ViewChunk: the reader is not in chunk mode");
        if (clo_ >= chi_ && !next_chunk()) return {nullptr, 0};
        const std::pair<const uint8_t *, size_t> v{cbuf_ + clo_, (size_t)(chi_ - clo_)};
        readTotal_ += v.second;
        clo_ = chi_;
        return v;
    }
    bool PrintStatus = false;

  private:
    static constexpr size_t kDirect = (size_t)1 << 20, kWindow = (size_t)4 << 20, kPiece = (size_t)4 << 20;
    // chunk mode: the next chunk's bytes into cbuf_[clo_, chi_); false behind the frame's last block
    bool next_chunk()
    {
        if (decoded_) return false;
        if (!ctx_) ctx_ = default_context();
        if (!fs_) {
            const int rc = mzd_fstream_open(ctx_, chunk_bytes_, &fs_);
            if (rc != MZD_OK) throw Error(rc, "mzd_fstream_open");
            if (!cbuf_) cbuf_ = static_cast<uint8_t *>(mzd_host_alloc(chunk_bytes_));  // pinned: the chunk's copy-out at the link's rate
            if (!cbuf_) throw Error(MZD_ERR_DEVICE, "mzd_host_alloc");
            pend_.assign(head_.begin(), head_.end());  // (the magic number Reset has read)
            pend_lo_ = 0;
        }
        for (;;) {
            uint64_t used = 0, made = 0;
            int done = 0;
            const int rc = mzd_fstream_next(fs_, pend_.data() + pend_lo_, pend_.size() - pend_lo_, cbuf_, chunk_bytes_, &used, &made, &done);
            if (rc != MZD_OK) {
                const std::string why = mzd_last_error(ctx_);
                release();
                throw Error(rc, "Read (" + why + ")");
            }
            pend_lo_ += used;
            if (done) {
                release();  // (window and history leave the device with the frame's last chunk)
                decoded_ = true;
            }
            if (made) {
                clo_ = 0;
                chi_ = made;
                return true;
            }
            if (done) return false;
            if (used == 0) {  // no whole block in what is here: more of the source
                pend_.erase(pend_.begin(), pend_.begin() + (long)pend_lo_);
                pend_lo_ = 0;
                const size_t at = pend_.size();
                pend_.resize(at + kPiece);
                source_->read(reinterpret_cast<char *>(pend_.data() + at), (std::streamsize)kPiece);
                const size_t got = (size_t)source_->gcount();
                pend_.resize(at + got);
                if (got == 0) {
                    release();
                    throw Error(MZD_ERR_TRUNCATED, "Read");  // io.ErrUnexpectedEOF: the source ended inside the frame
                }
            }
        }
    }
    void release()
    {
        if (fs_) mzd_fstream_close(fs_);
        fs_ = nullptr;
        if (db_) mzd_batch_free(ctx_, db_);
        db_ = nullptr;
        win_lo_ = win_hi_ = 0;
        std::vector<uint8_t>().swap(window_);
    }
    void decode()
    {
        if (!ctx_) ctx_ = default_context();
        std::vector<uint8_t> frame(head_.begin(), head_.end());
        frame.insert(frame.end(), std::istreambuf_iterator<char>(*source_), std::istreambuf_iterator<char>());
        int32_t dst = MZD_OK;
        uint64_t len = 0;
        int st = MZD_OK, rc;
        if (DevicePlanning()) {
            const uint64_t o = 0, l = frame.size();
            rc = mzd_batch_upload_frames(ctx_, frame.data(), frame.size(), 0, &o, &l, 1, nullptr, 0, &db_);
            if (rc == MZD_OK) rc = mzd_batch_run(ctx_, db_, nullptr);
            if (rc == MZD_OK) rc = mzd_batch_download(ctx_, db_, nullptr, &dst, &len);
            if (rc == MZD_OK) rc = mzd_batch_frame_layout(db_, &off_, nullptr);
            if (rc == MZD_OK) rc = mzd_batch_trim(ctx_, db_);  // the frame's bytes stay in HBM until they are read, nothing else does
        } else {
            mzd_plan *plan = mzd_plan_create();
            mzd_plan_set_device_tables(plan, 1);
            st = mzd_plan_add_frame(plan, frame.data(), frame.size(), nullptr);
            const mzd_batch *b = mzd_plan_finalize(plan);
            rc = mzd_batch_upload(ctx_, b, &db_);
            if (rc == MZD_OK) rc = mzd_batch_run(ctx_, db_, nullptr);
            if (rc == MZD_OK) rc = mzd_batch_download(ctx_, db_, nullptr, &dst, &len);
            if (rc == MZD_OK) off_ = b->frames[0].out_offset;
            if (rc == MZD_OK) rc = mzd_batch_trim(ctx_, db_);
            mzd_plan_destroy(plan);
        }
        if (rc != MZD_OK) {
            const std::string why = mzd_last_error(ctx_);
            release();
            throw Error(rc, "Read (" + why + ")");
        }
        if (st == MZD_OK) st = dst;
        if (st != MZD_OK) {
            release();
            throw Error(st, "Read");
        }
        len_ = len;
        pos_ = 0;
        decoded_ = true;
        if (len_ == 0) release();
    }
    std::istream *source_ = nullptr;
    mzd_ctx *ctx_ = nullptr;
    mzd_dbatch *db_ = nullptr;
    std::vector<char> head_;
    std::vector<uint8_t> window_;
    uint64_t off_ = 0, len_ = 0, pos_ = 0, win_lo_ = 0, win_hi_ = 0, readTotal_ = 0;
    bool decoded_ = false;
    // chunk mode
    uint64_t chunk_bytes_ = 0, clo_ = 0, chi_ = 0;
    mzd_fstream *fs_ = nullptr;
    uint8_t *cbuf_ = nullptr;
    std::vector<uint8_t> pend_;  // source bytes no chunk has consumed yet, from pend_lo_ on
    size_t pend_lo_ = 0;
};

// NewFrameDecompressor(s io.Reader, t io.Writer) -- framedecompressor.go:55
class FrameDecompressor {
  public:
    // chunk_bytes > 0: DecodeNextBlock decodes the frame's next CHUNK of whole blocks and writes it to the target
    // (framedecompressor.go:198-303 with a chunk for a block; FrameReader's chunk mode underneath)
    FrameDecompressor(std::istream *source, std::ostream *target, mzd_ctx *ctx = nullptr, uint64_t chunk_bytes = 0)
        : ctx_(ctx), chunk_bytes_(chunk_bytes)
    {
        Reset(source, target);
    }
    void Reset(std::istream *newsource, std::ostream *newtarget)  // framedecompressor.go:42-52
    {
        chunked_.reset();
        source_ = newsource;
        target_ = newtarget;
        done_ = false;
        BlockCounter = 0;
        head_.clear();
    }
    void CheckMagicnum()  // framedecompressor.go:130-150
    {
        char m[4];
        source_->read(m, 4);
        head_.assign(m, m + source_->gcount());
        if (head_.size() < 4) throw Error(MZD_ERR_TRUNCATED, "CheckMagicnum");
        static const unsigned char magic[4] = {0x28, 0xB5, 0x2F, 0xFD};
        if (std::memcmp(head_.data(), magic, 4) != 0) throw Error(MZD_ERR_MAGIC, "CheckMagicnum");
    }
    // Decompress decompresses the whole frame and writes the whole output to the target (:153-170)
    void Decompress()
    {
        if (done_) throw Error(MZD_ERR_OUT_OF_BLOCKS, "Decompress");
        if (chunk_bytes_) {
            while (!done_) DecodeNextBlock();
            return;
        }
        std::vector<uint8_t> frame(head_.begin(), head_.end());
        frame.insert(frame.end(), std::istreambuf_iterator<char>(*source_), std::istreambuf_iterator<char>());
        std::vector<int> st;
        auto out = DecodeFrames({frame}, &st, ctx_);
        if (st[0] != MZD_OK) throw Error(st[0], "Decompress");
        target_->write(reinterpret_cast<const char *>(out[0].data()), (std::streamsize)out[0].size());
        done_ = true;
    }
    // DecodeNextBlock (:198-244): all blocks of the frame come out of one device batch
    void DecodeNextBlock()
    {
        if (done_) throw Error(MZD_ERR_OUT_OF_BLOCKS, "DecodeNextBlock");
        if (!chunk_bytes_) return Decompress();
        if (!chunked_) {
            if (head_.empty()) CheckMagicnum();
            chunked_.reset(new FrameReader(nullptr, ctx_, chunk_bytes_));
            chunked_->ResetBehindMagic(source_, head_);
        }
        const auto v = chunked_->ViewChunk();
        if (v.second) {
            target_->write(reinterpret_cast<const char *>(v.first), (std::streamsize)v.second);
            BlockCounter++;
        }
        if (chunked_->Finished()) done_ = true;
    }
    bool Verbose = false;
    int BlockCounter = 0;

  private:
    std::istream *source_ = nullptr;
    std::ostream *target_ = nullptr;
    mzd_ctx *ctx_ = nullptr;
    std::vector<char> head_;
    bool done_ = false;
    uint64_t chunk_bytes_ = 0;
    std::unique_ptr<FrameReader> chunked_;
};

// A FrameReader that BATCHES.  The reference's harness feeds many frames through ONE reader, Reset per frame
// (framereader.go:35, cmd/sparkzstd/main.go:59,126); one frame at a time that is one device batch per frame.  Here the
// sources of the frames to come are known to the reader (Enqueue), it reads `lookahead` of them ahead, decodes them as ONE
// device batch -- on a background thread -- and serves them in order:
//     BatchFrameReader r(256);  for (auto &s : streams) r.Enqueue(&s);
//     while (r.Reset()) { while (size_t n = r.Read(buf, sizeof buf)) consume(buf, n); }
// Errors surface where FrameReader's do: a wrong magic number at Reset, a damaged block at the frame's first Read; the
// frames behind a damaged one are unaffected.
//
// Round 6: on one device the reader sits on the STREAMING path (mzd_stream_submit / mzd_stream_wait): a worker thread gathers
// the frames of a batch into a PINNED input buffer (several threads copy), submits it -- the device plans the frames itself, no
// host planner -- and keeps TWO batches in flight, so that the copy-in of one batch, the decode of the one before and the copy-out
// of the one before that overlap (what mzd_stream_* is for); the regenerated bytes arrive in a pinned output buffer and the reader
// hands them out from there -- Read copies into the caller's buffer like io.Reader does, View() lends the frame's bytes in place
// (valid until the next Reset) for consumers that can do without that copy.  Four slots: one served, two in flight, one being
// gathered.  A source handed to Enqueue belongs to the reader until its frame has been served (streams are read on the worker).
// With a list of devices a batch goes through DecodeFramesOn instead (one context and host thread per device), one at a time.
class BatchFrameReader {
  public:
    explicit BatchFrameReader(size_t lookahead = 256, std::vector<int> devices = {}, mzd_ctx *ctx = nullptr, unsigned gather_threads = 4)
        : lookahead_(lookahead ? lookahead : 1), devices_(std::move(devices)), ctx_(ctx), gather_threads_(gather_threads ? gather_threads : 1)
    {
    }
    ~BatchFrameReader()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        if (stream_) mzd_stream_destroy(stream_);
        for (Slot &sl : slots_) {
            if (sl.in) mzd_host_free(sl.in);
            if (sl.out) mzd_host_free(sl.out);
        }
    }
    BatchFrameReader(const BatchFrameReader &) = delete;
    BatchFrameReader &operator=(const BatchFrameReader &) = delete;
    void Enqueue(std::istream *source) { Push({source, {}, nullptr, 0}); }
    void Enqueue(std::vector<uint8_t> frame) { Push({nullptr, std::move(frame), nullptr, 0}); }
    // the frame's bytes stay the caller's (alive until the frame has been served): nothing is copied before the gather
    void EnqueueView(const uint8_t *p, size_t n) { Push({nullptr, {}, p, n}); }
    // the next frame becomes the current one; false when there is none left
    bool Reset()
    {
        have_ = false;
        status_ = MZD_OK;
        cur_ = nullptr;
        cur_len_ = pos_ = 0;
        while (serving_ < 0 || ready_pos_ == slots_[serving_].n) {
            std::unique_lock<std::mutex> g(mu_);
            if (serving_ >= 0) {  // the slot that has been served goes back to the worker
                free_.push_back(serving_);
                serving_ = -1;
                cv_.notify_all();
            }
            if (!worker_.joinable() && !dead_) worker_ = std::thread([this] { Work(); });
            if (dead_) return false;  // (a device error ended the worker: it was thrown once)
            cv_.wait(g, [this] { return !ready_.empty() || error_ != MZD_OK || dead_ || (pending_.empty() && busy_ == 0); });
            if (error_ != MZD_OK) {
                const int e = error_;
                error_ = MZD_OK;
                throw Error(e, "BatchFrameReader: " + what_);
            }
            if (ready_.empty()) return false;  // nothing queued, nothing in flight
            serving_ = ready_.front();
            ready_.erase(ready_.begin());
            ready_pos_ = 0;
        }
        const Slot &sl = slots_[serving_];
        const size_t i = ready_pos_++;
        have_ = true;
        FramesServed++;
        const int st = sl.status[i];
        if (sl.owned.empty()) {
            cur_ = sl.out + sl.out_off[i];
            cur_len_ = st == MZD_OK ? (size_t)sl.out_len[i] : 0;
        } else {  // (the multi-device path: a vector per frame)
            cur_ = sl.owned[i].data();
            cur_len_ = sl.owned[i].size();
        }
        if (st == MZD_ERR_MAGIC || (st == MZD_ERR_TRUNCATED && sl.len[i] < 4)) throw Error(st, "Reset");
        status_ = st;
        return true;
    }
    size_t Read(uint8_t *p, size_t n)
    {
        if (!have_ && !Reset()) return 0;
        if (status_ != MZD_OK) throw Error(status_, "Read");
        const size_t k = std::min(n, cur_len_ - pos_);
        std::memcpy(p, cur_ + pos_, k);
        pos_ += k;
        return k;
    }
    // the current frame's regenerated bytes where they are (pinned host memory on the streaming path): no copy; valid until the
    // next Reset.  Throws what Read would throw.
    std::pair<const uint8_t *, size_t> View()
    {
        if (!have_ && !Reset()) return {nullptr, 0};
        if (status_ != MZD_OK) throw Error(status_, "Read");
        pos_ = cur_len_;
        return {cur_, cur_len_};
    }
    size_t FramesServed = 0;

  private:
    struct Source {
        std::istream *stream;
        std::vector<uint8_t> bytes;
        const uint8_t *view;
        size_t view_len;
    };
    struct Slot {  // one batch: pinned buffers (grow only) and what mzd_stream_wait reports per frame
        uint8_t *in = nullptr, *out = nullptr;
        size_t in_cap = 0, out_cap = 0, n = 0;
        std::vector<uint64_t> off, len, out_len, out_off;
        std::vector<int32_t> status;
        std::vector<std::vector<uint8_t>> owned;  // (multi-device path only)
        uint64_t ticket = 0;
    };
    static constexpr int kSlots = 4;
    void Push(Source s)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            pending_.push_back(std::move(s));
        }
        cv_.notify_all();
    }
    static void grow(uint8_t *&p, size_t &cap, size_t need)
    {
        if (cap >= need) return;
        if (p) mzd_host_free(p);
        cap = need + need / 4 + 4096;
        p = (uint8_t *)mzd_host_alloc(cap);
        if (!p) {
            cap = 0;
            throw Error(MZD_ERR_DEVICE, "mzd_host_alloc");
        }
    }
    // the worker: gather -> submit, two batches in flight, collect the oldest -> ready
    void Work()
    {
        std::vector<int> inflight;
        try {
            for (;;) {
                std::vector<Source> take;
                int slot = -1;
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return stop_ || !inflight.empty() || (!pending_.empty() && !free_.empty()); });
                    if (stop_) break;
                    const bool can_fill = !pending_.empty() && !free_.empty() && inflight.size() < 2;
                    if (can_fill) {
                        slot = free_.back();
                        free_.pop_back();
                        const size_t n = std::min(lookahead_, pending_.size());
                        take.assign(std::make_move_iterator(pending_.begin()), std::make_move_iterator(pending_.begin() + (long)n));
                        pending_.erase(pending_.begin(), pending_.begin() + (long)n);
                        busy_++;
                    } else if (inflight.empty()) {
                        continue;  // (sources but no free slot: the consumer has to return one)
                    }
                }
                if (slot >= 0) {
                    Fill(slots_[slot], take);
                    inflight.push_back(slot);
                    // one more batch right away if there is room: its copy-in overlaps the decode of this one
                    std::lock_guard<std::mutex> g(mu_);
                    if (inflight.size() < 2 && !pending_.empty() && !free_.empty()) continue;
                }
                if (!inflight.empty()) {
                    const int s0 = inflight.front();
                    inflight.erase(inflight.begin());
                    Collect(slots_[s0]);
                    {
                        std::lock_guard<std::mutex> g(mu_);
                        ready_.push_back(s0);
                        busy_--;
                    }
                    cv_.notify_all();
                }
            }
        } catch (const Error &e) {
            std::lock_guard<std::mutex> g(mu_);
            error_ = e.code();
            what_ = e.what();
            busy_ = 0;
            dead_ = true;
            cv_.notify_all();
        }
    }
    void Fill(Slot &sl, std::vector<Source> &take)
    {
        const size_t n = take.size();
        for (Source &s : take)
            if (s.stream) s.bytes.assign(std::istreambuf_iterator<char>(*s.stream), std::istreambuf_iterator<char>());
        sl.n = n;
        sl.off.resize(n);
        sl.len.resize(n);
        sl.out_len.assign(n, 0);
        sl.out_off.assign(n, 0);
        sl.status.assign(n, MZD_OK);
        sl.owned.clear();
        sl.ticket = 0;
        size_t total = 0;
        for (size_t i = 0; i < n; i++) {
            sl.off[i] = total;
            sl.len[i] = take[i].view ? take[i].view_len : take[i].bytes.size();
            total += sl.len[i];
        }
        if (!devices_.empty()) {  // several devices: DecodeFramesOn, a vector per frame (no streaming path across devices yet)
            std::vector<std::vector<uint8_t>> frames;
            for (Source &s : take) frames.push_back(s.view ? std::vector<uint8_t>(s.view, s.view + s.view_len) : std::move(s.bytes));
            std::vector<int> st;
            sl.owned = DecodeFramesOn(devices_, frames, &st);
            for (size_t i = 0; i < n; i++) sl.status[i] = st[i];
            if (sl.owned.empty()) sl.owned.resize(1);  // (marks the slot as this path's)
            return;
        }
        mzd_ctx *ctx = ctx_ ? ctx_ : default_context();
        if (!stream_) {
            int err = 0;
            stream_ = mzd_stream_create(ctx, 2, &err);
            if (!stream_) throw Error(err, "mzd_stream_create");
        }
        grow(sl.in, sl.in_cap, total + 64);
        // gather + the output the batch may need: every frame's bound (declared content size, capped by what its blocks can
        // regenerate) from a walk over its headers; a frame the walk cannot read gets an empty slab and its status from the device
        std::vector<uint64_t> need(gather_threads_, 0);
        auto part = [&](unsigned t) {
            uint64_t nd = 0;
            for (size_t i = n * t / gather_threads_; i < n * (t + 1) / gather_threads_; i++) {
                if (sl.len[i]) std::memcpy(sl.in + sl.off[i], take[i].view ? take[i].view : take[i].bytes.data(), sl.len[i]);
                uint64_t fo = 0, fl = 0, bound = 0, tot = 0;
                uint32_t found = 0;
                (void)mzd_split_frames(sl.in + sl.off[i], sl.len[i], &fo, &fl, &bound, 1, &found, &tot);
                nd += ((found ? bound : 0) + 255) / 256 * 256 + 256;
            }
            need[t] = nd;
        };
        std::vector<std::thread> helpers;
        for (unsigned t = 1; t < gather_threads_; t++) helpers.emplace_back(part, t);
        part(0);
        for (auto &h : helpers) h.join();
        uint64_t need_all = 512;
        for (uint64_t v : need) need_all += v;
        grow(sl.out, sl.out_cap, need_all);
        const int rc = mzd_stream_submit(stream_, sl.in, total, sl.off.data(), sl.len.data(), (uint32_t)n, sl.out, sl.out_cap, &sl.ticket);
        if (rc != MZD_OK) throw Error(rc, "mzd_stream_submit");
    }
    void Collect(Slot &sl)
    {
        if (!sl.ticket) return;  // (the multi-device path finished in Fill)
        const int rc = mzd_stream_wait(stream_, sl.ticket, sl.status.data(), sl.out_len.data(), sl.out_off.data());
        sl.ticket = 0;
        if (rc >= MZD_ERR_DEVICE) throw Error(rc, "mzd_stream_wait");
    }
    size_t lookahead_;
    std::vector<int> devices_;
    mzd_ctx *ctx_;
    unsigned gather_threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Source> pending_;               // (under mu_)
    std::vector<int> ready_, free_ = {3, 2, 1, 0};  // decoded slots in order / slots the worker may fill (under mu_)
    int busy_ = 0;                              // batches taken from pending_ and not yet in ready_ (under mu_)
    bool stop_ = false, dead_ = false;
    int error_ = MZD_OK;
    std::string what_;
    std::thread worker_;
    mzd_stream *stream_ = nullptr;
    Slot slots_[kSlots];
    int serving_ = -1;
    size_t ready_pos_ = 0;
    const uint8_t *cur_ = nullptr;
    size_t cur_len_ = 0, pos_ = 0;
    bool have_ = false;
    int status_ = MZD_OK;
};

}  // namespace sparkzstd
