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
#include <cstring>
#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
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

// NewFrameDecompressor(s io.Reader, t io.Writer) -- framedecompressor.go:55
class FrameDecompressor {
  public:
    FrameDecompressor(std::istream *source, std::ostream *target, mzd_ctx *ctx = nullptr) : ctx_(ctx) { Reset(source, target); }
    void Reset(std::istream *newsource, std::ostream *newtarget)  // framedecompressor.go:42-52
    {
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
        Decompress();
    }
    bool Verbose = false;
    int BlockCounter = 0;

  private:
    std::istream *source_ = nullptr;
    std::ostream *target_ = nullptr;
    mzd_ctx *ctx_ = nullptr;
    std::vector<char> head_;
    bool done_ = false;
};

// NewFrameReader(source io.Reader) -- framereader.go:17
class FrameReader {
  public:
    explicit FrameReader(std::istream *source = nullptr, mzd_ctx *ctx = nullptr) : ctx_(ctx)
    {
        if (source) Reset(source);
    }
    void Reset(std::istream *source)  // framereader.go:35-49: magic number + frame header are checked here
    {
        buffer_.clear();
        pos_ = 0;
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
    // last block has been drained)
    size_t Read(uint8_t *p, size_t n)
    {
        if (!decoded_) {
            std::vector<uint8_t> frame(head_.begin(), head_.end());
            frame.insert(frame.end(), std::istreambuf_iterator<char>(*source_), std::istreambuf_iterator<char>());
            std::vector<int> st;
            auto out = DecodeFrames({frame}, &st, ctx_);
            if (st[0] != MZD_OK) throw Error(st[0], "Read");
            buffer_ = std::move(out[0]);
            decoded_ = true;
        }
        const size_t k = std::min(n, buffer_.size() - pos_);
        std::memcpy(p, buffer_.data() + pos_, k);
        pos_ += k;
        readTotal_ += k;
        return k;
    }
    bool PrintStatus = false;

  private:
    std::istream *source_ = nullptr;
    mzd_ctx *ctx_ = nullptr;
    std::vector<char> head_;
    std::vector<uint8_t> buffer_;
    size_t pos_ = 0;
    uint64_t readTotal_ = 0;
    bool decoded_ = false;
};

}  // namespace sparkzstd
