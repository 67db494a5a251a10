// sparkzstd_verify -- the reference's test harness (cmd/sparkzstd/main.go:113-195) on the device
// path: every argument is a .zst file; the output of FrameReader is compared byte for byte with the
// file of the same name minus ".zst" (main.go:46-111) and an average speed is printed (:177-191).
// A first argument --device-plan parses the headers on the device too (k_parse) instead of in the host planner.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/sparkzstd_frame.hpp"

int main(int argc, char **argv)
{
    std::vector<std::string> diffs, errs;
    sparkzstd::FrameReader comp;  // one shared reader, Reset per file (main.go:126,59)
    double seconds = 0;
    uint64_t bytes = 0;
    int first = 1;
    if (argc > 1 && std::string(argv[1]) == "--device-plan") {
        sparkzstd::DevicePlanning() = true;
        first = 2;
    }
    if (argc > first + 1 && std::string(argv[first]) == "--devices") {
        // --devices 0,1,...: every file is one frame of ONE batch, split over the listed devices (DecodeFramesOn; a device
        // listed twice = two contexts on it), results compared in frame order
        std::vector<int> devices;
        for (const char *p = argv[first + 1]; *p;) {
            devices.push_back(atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        std::vector<std::vector<uint8_t>> frames, originals;
        std::vector<std::string> names;
        for (int i = first + 2; i < argc; i++) {
            const std::string path = argv[i];
            std::ifstream z(path, std::ios::binary), o(path.substr(0, path.size() - 4), std::ios::binary);
            if (!z || !o) {
                errs.push_back(path + ": cannot open");
                continue;
            }
            frames.emplace_back(std::istreambuf_iterator<char>(z), std::istreambuf_iterator<char>());
            originals.emplace_back(std::istreambuf_iterator<char>(o), std::istreambuf_iterator<char>());
            names.push_back(path);
        }
        const auto t0 = std::chrono::steady_clock::now();
        try {
            std::vector<int> st;
            const auto out = sparkzstd::DecodeFramesOn(devices, frames, &st);
            for (size_t i = 0; i < frames.size(); i++) {
                if (st[i] != MZD_OK) errs.push_back(names[i] + " status " + mzd_strerror(st[i]));
                else if (out[i] != originals[i]) diffs.push_back(names[i]);
                bytes += out[i].size();
            }
        } catch (const sparkzstd::Error &e) {
            errs.push_back(std::string("DecodeFramesOn: ") + e.what());
        }
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        first = argc;
    }
    if (argc > first + 2 && std::string(argv[first]) == "--reader-selftest") {
        // --reader-selftest <lookahead> <x.zst> ...: BatchFrameReader's behaviour beyond the happy path (framereader.go:35-109): every
        // frame equals DecodeFrames' bytes through Read AND through View; a frame cut in half raises at ITS read and the frames behind it
        // are served; a wrong magic number raises at Reset; sources enqueued while the reader is being served are served; a reader that
        // is destroyed with batches in flight shuts down.
        const size_t look = (size_t)atoi(argv[first + 1]);
        std::vector<std::vector<uint8_t>> frames;
        for (int i = first + 2; i < argc; i++) {
            std::ifstream z(argv[i], std::ios::binary);
            frames.emplace_back(std::istreambuf_iterator<char>(z), std::istreambuf_iterator<char>());
        }
        int fails = 0;
        auto check = [&](bool ok, const char *what, size_t i) {
            if (!ok) {
                printf("selftest FAILED: %s (frame %zu)\n", what, i);
                fails++;
            }
        };
        std::vector<int> st;
        const auto want = sparkzstd::DecodeFrames(frames, &st);
        const size_t n = frames.size(), cut_at = n / 3, magic_at = 2 * n / 3;
        {
            sparkzstd::BatchFrameReader br(look);
            for (size_t i = 0; i < n / 2; i++) {
                if (i == cut_at) br.Enqueue(std::vector<uint8_t>(frames[i].begin(), frames[i].begin() + (long)(frames[i].size() / 2)));
                else br.EnqueueView(frames[i].data(), frames[i].size());
            }
            std::vector<uint8_t> buf(50000), got;
            for (size_t i = 0; i < n; i++) {
                if (i == n / 2)  // the second half becomes known while the first is being served
                    for (size_t k = n / 2; k < n; k++) {
                        if (k == magic_at) br.Enqueue(std::vector<uint8_t>{0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
                        else br.Enqueue(frames[k]);
                    }
                bool threw = false;
                try {
                    if (!br.Reset()) {
                        check(false, "reader ran out of frames", i);
                        break;
                    }
                } catch (const sparkzstd::Error &e) {
                    threw = true;
                    check(i == magic_at && e.code() == MZD_ERR_MAGIC, "Reset threw", i);
                }
                if (i == magic_at) {
                    check(threw, "a wrong magic number must raise at Reset", i);
                    continue;
                }
                got.clear();
                try {
                    if (i % 2) {
                        const auto v = br.View();
                        got.assign(v.first, v.first + v.second);
                        check(br.Read(buf.data(), buf.size()) == 0, "EOF behind View", i);
                    } else {
                        while (const size_t k = br.Read(buf.data(), buf.size())) got.insert(got.end(), buf.begin(), buf.begin() + (long)k);
                    }
                    check(i != cut_at, "a frame cut in half must raise at its read", i);
                    check(st[i] == MZD_OK && got == want[i], "bytes differ from DecodeFrames", i);
                } catch (const sparkzstd::Error &e) {
                    check(i == cut_at, "Read threw", i);
                }
            }
            check(!br.Reset(), "EOF: no frame left", n);
            check(br.FramesServed == n, "frames served", br.FramesServed);
        }
        {
            // destroyed with batches in flight: the destructor stops the worker and frees the pinned buffers
            sparkzstd::BatchFrameReader br(look);
            for (size_t rep = 0; rep < 6; rep++)
                for (auto &f : frames) br.EnqueueView(f.data(), f.size());
            check(br.Reset(), "first frame", 0);
        }
        if (fails) printf("reader selftest: %d failure(s)\n", fails);
        else printf("reader selftest ok (%zu frames, lookahead %zu)\n", n, look);
        return fails ? 1 : 0;
    }
    if (argc > first + 3 && std::string(argv[first]) == "--bench") {
        // --bench <file of concatenated frames> <frames to serve> <lookahead> [read|view]: host-to-host throughput of the reader that
        // batches (framereader.go:35-109 consumer shape: one reader, Reset per frame, the frame consumed whole).  The file's frames are
        // served round robin until the count is reached (content repeats, every frame is decoded again); "read" = Read into a 64 KiB
        // buffer like the reference harness (a copy per byte), "view" = the bytes lent in place.  Every frame's regenerated length is
        // checked against the first pass's, its bytes by a running 64-bit sum that must repeat with the file.
        std::ifstream zf(argv[first + 1], std::ios::binary);
        std::vector<uint8_t> blob((std::istreambuf_iterator<char>(zf)), std::istreambuf_iterator<char>());
        const size_t total_frames = (size_t)atoll(argv[first + 2]), look = (size_t)atoi(argv[first + 3]);
        const bool view = argc > first + 4 && std::string(argv[first + 4]) == "view";
        std::vector<uint64_t> off(1 << 20), len(1 << 20), bound(1 << 20);
        uint32_t nf = 0;
        uint64_t tot = 0;
        const int rc = mzd_split_frames(blob.data(), blob.size(), off.data(), len.data(), bound.data(), (uint32_t)off.size(), &nf, &tot);
        if (rc != MZD_OK || nf == 0 || nf > off.size()) {
            printf("bench: cannot split %s (rc %d, %u frames)\n", argv[first + 1], rc, nf);
            return 1;
        }
        sparkzstd::BatchFrameReader br(look);
        std::vector<uint64_t> sums(nf, 0), lens(nf, 0);
        std::vector<uint8_t> buf(1 << 16);
        uint64_t in_bytes = 0;
        bool ok = true;
        // warm-up, untimed: four batches -- every slot of the reader has its pinned buffers (hipHostMalloc of a gigabyte takes a
        // fifth of a second: a reader lives longer than that) and the context its kernels
        const size_t warm = std::min<size_t>(4 * look, total_frames);
        for (size_t i = 0; i < warm; i++) br.EnqueueView(blob.data() + off[i % nf], (size_t)len[i % nf]);
        for (size_t i = 0; i < warm; i++) ok = br.Reset() && ok;
        for (size_t i = 0; i < total_frames; i++) br.EnqueueView(blob.data() + off[i % nf], (size_t)len[i % nf]);
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < total_frames; i++) {
            if (!br.Reset()) { ok = false; break; }
            uint64_t sum = 0, n_out = 0;
            if (view) {
                const auto v = br.View();
                n_out = v.second;
                // (touch what was lent: a word per 4 KiB page -- the consumer's own pass over the bytes is not the reader's cost)
                for (size_t k = 0; k + 8 <= v.second; k += 4096) { uint64_t w; std::memcpy(&w, v.first + k, 8); sum += w; }
            } else {
                while (const size_t n = br.Read(buf.data(), buf.size())) {
                    n_out += n;
                    for (size_t k = 0; k + 8 <= n; k += 4096) { uint64_t w; std::memcpy(&w, buf.data() + k, 8); sum += w; }
                }
            }
            bytes += n_out;
            in_bytes += len[i % nf];
            if (i < nf) { sums[i] = sum; lens[i] = n_out; }
            else if (sums[i % nf] != sum || lens[i % nf] != n_out) ok = false;
        }
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("{\"bench\": \"BatchFrameReader over mzd_stream_*\", \"mode\": \"%s\", \"frames\": %zu, \"distinct_frames\": %u, \"lookahead\": %zu, "
               "\"seconds\": %.4f, \"frames_per_s\": %.0f, \"out_GBs\": %.2f, \"in_GBs\": %.2f, \"out_bytes\": %llu, \"consistent\": %s}\n",
               view ? "view" : "read", total_frames, nf, look, seconds, total_frames / seconds, bytes / seconds / 1e9, in_bytes / seconds / 1e9,
               (unsigned long long)bytes, ok ? "true" : "false");
        return ok ? 0 : 1;
    }
    if (argc > first + 1 && std::string(argv[first]) == "--batch-reader") {
        // --batch-reader N: the harness's own loop (one reader, Reset per file, Read until EOF) over a reader that decodes N
        // frames per device batch and reads ahead
        const size_t look = (size_t)atoi(argv[first + 1]);
        sparkzstd::BatchFrameReader br(look);
        std::vector<std::ifstream> zs;
        std::vector<std::string> names;
        for (int i = first + 2; i < argc; i++) {
            zs.emplace_back(argv[i], std::ios::binary);
            names.push_back(argv[i]);
        }
        for (auto &z : zs) br.Enqueue(&z);
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < names.size(); i++) {
            const std::string original = names[i].substr(0, names[i].size() - 4);
            std::ifstream o(original, std::ios::binary);
            try {
                if (!br.Reset()) {
                    errs.push_back(names[i] + ": reader ran out of frames");
                    break;
                }
                std::vector<uint8_t> got(1 << 16), want(1 << 16);
                bool differ = !o;
                for (;;) {
                    const size_t n = br.Read(got.data(), got.size());
                    o.read(reinterpret_cast<char *>(want.data()), (std::streamsize)std::max<size_t>(n, 1));
                    const size_t m = (size_t)o.gcount();
                    if (n == 0) {
                        differ = differ || m != 0;
                        break;
                    }
                    if (m != n || std::memcmp(got.data(), want.data(), n) != 0) differ = true;
                    bytes += n;
                }
                if (differ) diffs.push_back(original);
            } catch (const sparkzstd::Error &e) {
                errs.push_back(original + " Decompress-Read Err: " + e.what());
            }
        }
        if (br.Reset()) errs.push_back("reader served more frames than it was given");
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        first = argc;
    }
    for (int i = first; i < argc; i++) {
        const std::string path = argv[i];
        const std::string original = path.substr(0, path.size() - 4);
        std::ifstream z(path, std::ios::binary), o(original, std::ios::binary);
        if (!z || !o) {
            errs.push_back(path + ": cannot open");
            continue;
        }
        const auto t0 = std::chrono::steady_clock::now();
        try {
            comp.Reset(&z);
            std::vector<uint8_t> got(1 << 16), want(1 << 16);
            bool differ = false;
            for (;;) {
                const size_t n = comp.Read(got.data(), got.size());
                o.read(reinterpret_cast<char *>(want.data()), (std::streamsize)std::max<size_t>(n, 1));
                const size_t m = (size_t)o.gcount();
                if (n == 0) {  // io.EOF: both must end at the same byte (main.go:70-82)
                    differ = differ || m != 0;
                    break;
                }
                if (m != n || std::memcmp(got.data(), want.data(), n) != 0) differ = true;
                bytes += n;
            }
            if (differ) diffs.push_back(original);
        } catch (const sparkzstd::Error &e) {
            errs.push_back(original + " Decompress-Read Err: " + e.what());
        }
        seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (diffs.empty()) printf("Found no diffs in any files! Good job you!\n");
    else for (auto &d : diffs) printf("Found diffs in file: %s\n", d.c_str());
    if (errs.empty()) printf("Found no unexpected errors in any files! Good job you!\n");
    else for (auto &e : errs) printf("Found unexpected error: %s\n", e.c_str());
    printf("Average detected Speed: %.1f MB/s (%llu bytes, includes host planning + H2D + D2H per file)\n",
           seconds > 0 ? bytes / seconds / 1e6 : 0.0, (unsigned long long)bytes);
    return diffs.empty() && errs.empty() ? 0 : 1;
}
