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
