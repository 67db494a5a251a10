"""CPU only: the host code that parses untrusted bytes, under AddressSanitizer + UndefinedBehaviorSanitizer.
  * the product's planner and mzd_split_frames (sparkzstd_amd/csrc/planner.cpp -- plain C++, built here without
    HIP): the corpus intact, tools/plan_soak.py's mutation mix (truncation, byte flips, header damage), every
    prefix of the small frames, threaded planning, concatenated / skippable streams.  The error model it must
    keep is the reference's: a status, never a fault (structure/literals.go:43-44,206-207, fse/fse.go:133,
    decompression/framedecompressor.go:90);
  * the oracle (the checker) on the same inputs, with exact-size destinations.
GPU AddressSanitizer is not available on the pool; the device kernels' bounds behaviour is covered by the fuzz
tests in tests/test_gpu_corpus.py."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1",
           UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _build():
    subprocess.check_call(["make", "-s", "-C", SAN])


def _corpus_files():
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "decodecorpus", "*.zst")))
    assert len(files) == 100
    return files


def test_planner_and_split_frames_under_asan_ubsan():
    _build()
    r = subprocess.run([os.path.join(SAN, "build", "san_planner"), "24"] + _corpus_files(), env=ENV, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "san_planner ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]


def test_oracle_under_asan_ubsan():
    _build()
    r = subprocess.run([os.path.join(SAN, "build", "san_oracle"), "12"] + _corpus_files(), env=ENV, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "san_oracle ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
