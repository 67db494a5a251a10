#!/usr/bin/env python3
"""bench.py -- decompressed MB/s of the zstd block-decode hot path on MI355X.

One "step" = one pass of the whole hot path (Huffman literal decode -> FSE sequence decode ->
sequence execution) over one resident batch of synthetic frames.  Default workload = BASELINE.json
configs[3]: 65536 independent single-block 128 KiB text-like frames per GPU,
generated deterministically by tools/synth (own zstd-format encoder).  Inputs, descriptors and
tables are resident in HBM before the timed region; outputs stay in HBM.

Multi-GPU (BASELINE configs[4]): ONE 65536-frame batch, split into contiguous frame ranges, one process per GPU
(8192 frames per GPU at N = 8; `scaling: "strong"`; `--weak` keeps 65536 frames PER GPU instead).  Frames are
independent, so there is no data-path collective and no RCCL anywhere: the ranks only meet at a barrier and
exchange their times / figures, over torch.distributed's gloo backend (TCP on 127.0.0.1; `--rendezvous nccl`
for the RCCL equivalent).  `python bench.py --gpus N` with no WORLD_SIZE in the environment launches its own N
ranks (child processes, before anything touches the GPU) and relays rank 0's line; under torchrun
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# the files libmzd.so is made of, in the order sparkzstd_amd/csrc/Makefile hashes them into mzd_build_id()
LIBRARY_SOURCES = ("mzd_api.hip", "mzd_device.h", "mzd_exec.hip", "mzd_exec_b.hip", "mzd_exec_blk.hip", "mzd_exec_c.hip", "mzd_huf.hip", "mzd_huf_w.hip",
                   "mzd_kernels.hip", "mzd_parse.hip", "mzd_seq.hip", "mzd_seq_q4.hip", "mzd_util.hip", "planner.cpp", "../../include/mzd.h")


def library_src_sha16():
    """what `make -C sparkzstd_amd/csrc` would stamp into a library built from the sources in the tree NOW (None: no sources here)"""
    h = hashlib.sha256()
    try:
        for rel in LIBRARY_SOURCES:
            with open(os.path.join(ROOT, "sparkzstd_amd", "csrc", rel), "rb") as f:
                h.update(f.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def kernel_src_sha16():
    """Identity of the device code a counter file was measured on: what the LOADED library says about itself (mzd_build_id: the
    hash of its sources, put in at build time) -- not a hash of the files beside it, which a stale libmzd.so would carry too.  A
    traffic / issue file carries it and is only quoted when it matches the library this run loaded."""
    from sparkzstd_amd import _lib
    return _lib.load().mzd_build_id().decode()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=4, help="SURVEY 8d config: 2 raw/rle, 3 huffman only, 4 full")
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "corpus"],
                    help="corpus: the reference's own 100 decodecorpus frames (tests/golden/decodecorpus: real zstd output, mixed block "
                         "types / table modes / window sizes, multi-block), replicated to --corpus-gib of output at distinct HBM addresses")
    ap.add_argument("--corpus-gib", type=float, default=4.0)
    ap.add_argument("--frames", "--frames-per-gpu", dest="frames_per_gpu", type=int, default=0,
                    help="frames of the batch: split over the GPUs by default (strong scaling), per GPU with --weak; "
                         "default 65536 (config 4) / 4096 (configs 2, 3)")
    ap.add_argument("--frame-bytes", type=int, default=131072, help="regenerated size of every frame (multiple of 256); above 128 KiB a frame has several blocks with cross-block matches and repeat-offset history")
    ap.add_argument("--strong", action="store_true", help="(default) BASELINE configs[4]: ONE batch of --frames-per-gpu x 1 frames "
                    "(65536) split over the ranks in contiguous ranges")
    ap.add_argument("--weak", action="store_true", help="keep the whole batch PER GPU instead (weak scaling)")
    ap.add_argument("--rendezvous", default=os.environ.get("MZD_BENCH_BACKEND", "gloo"), choices=["gloo", "nccl"],
                    help="how the ranks meet for the barrier and the max-over-ranks time (the data path has no collective)")
    ap.add_argument("--exec-variant", type=int, default=0, help="0 auto, 1 k_exec (workgroup per frame), 2 k_exec_b (wavefront per frame, lane per byte), 3 k_exec_b with the blocks of a frame side by side, 4 the same in jobs of four blocks, 5 k_exec_c (two bytes per lane and pass, fixed-point passes: what 0 takes for batches with sequences)")
    ap.add_argument("--seq-variant", type=int, default=0)
    ap.add_argument("--verify-checksum", action="store_true", help="frames carry the zstd content checksum and the device verifies it after the pass (k_xxh64; an extension, off by default like in the reference)")
    ap.add_argument("--device-plan", action="store_true", help="parse the frame / block / section headers on the device too (mzd_batch_upload_frames) instead of in the host planner")
    ap.add_argument("--host-tables", action="store_true", help="build the FSE / Huffman decode tables in the host planner instead of on the device")
    ap.add_argument("--exec-threads", type=int, default=0)
    ap.add_argument("--exec-chunk", type=int, default=0)
    ap.add_argument("--huf-min-lds", type=int, default=0)
    ap.add_argument("--huf-variant", type=int, default=0, help="0 auto, 1 k_huf beside the sequence stage, 2 k_huf_seg (wavefront per stream), 3 k_huf first (transposed bulk phase)")
    ap.add_argument("--no-split", action="store_true", help="do not overlap k_seq(tail) with k_exec(head)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--gen-seconds", type=float, default=60.0,
                    help="host time budget for generating the synthetic batch; if all-distinct frames would "
                         "take longer, fewer distinct frames are generated and physically replicated")
    ap.add_argument("--window-log", type=int, default=0, help="synthetic frames: matches reach back at most 2^N bytes (what zstd's windowLog "
                    "does: 23 at its levels up to 19); 0 = the generator's default, 2^27 -- the frames of BASELINE's configs are unaffected "
                    "(128 KiB), large frames are: block mode spells an origin with three passes instead of four when offsets stay below 8 MiB")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--traffic-from", default="", help="JSON written by tools/profile_round.sh in the same gpurun "
                    "(FETCH_SIZE / WRITE_SIZE per kernel from separate --pmc passes); default: the newest profiles/r<N>_traffic_<workload>.json. "
                    "Quoted only if its kernel_src_sha16 and workload match this run")
    ap.add_argument("--issue-from", default="", help="JSON written by tools/profile_counters.sh (SQ counters per kernel); default: "
                    "the newest profiles/r<N>_issue_<workload>.json; quoted under the same condition")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the measured copy ceiling (mzd_measure_copy)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` object (the other BASELINE configs, real data, the "
                    "8192-frame shard and one large frame, measured in the same process after the headline; only the default headline "
                    "invocation on one GPU carries it)")
    return ap.parse_args()


def self_launch(a):
    """`bench.py --gpus N` outside torchrun: the parent spawns the N ranks as fresh child processes BEFORE it has
    touched the GPU (never re-executes a process that initialised HIP), relays rank 0's JSON line, and exits with
    the worst child's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def usable_cores():
    """Host threads this process can really run: min(cpu_count, affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def cpu_baseline(blob, off, ln, exp_len, budget_s, cks=None):
    """Oracle ("port" of the reference algorithm, plain C) on the host cores, bounded sample of the
    SAME frames.  Reported next to the GPU number; never the thing measured as `value`."""
    from tests.oracle_binding import load_oracle
    orc = load_oracle()
    cores, quota = usable_cores()
    n_total = len(off)
    cap = int(exp_len.max()) if n_total else 0
    csum = np.concatenate([[0], np.cumsum(exp_len.astype(np.float64))])  # output bytes of the first k frames

    def run(first, count, nthreads):
        per = (count + nthreads - 1) // nthreads
        outs = []
        ths = []
        for t in range(nthreads):
            a, b = first + t * per, min(first + count, first + (t + 1) * per)
            if a >= b:
                continue
            n = b - a
            # every frame of a thread regenerates into the same pre-faulted buffer: the baseline is
            # not charged for page faults or for writing 8 GB to DRAM (optimistic for the CPU)
            dst = np.zeros(cap + 64, dtype=np.uint8)
            doff = np.zeros(n, dtype=np.uint64)
            dcap = np.full(n, cap, dtype=np.uint64)
            olen = np.empty(n, dtype=np.uint64)
            st = np.empty(n, dtype=np.int32)
            o = np.ascontiguousarray(off[a:b])
            l = np.ascontiguousarray(ln[a:b])
            outs.append((dst, olen, st, o, l, doff, dcap, n, np.ascontiguousarray(exp_len[a:b])))
        t0 = time.perf_counter()
        for (dst, olen, st, o, l, doff, dcap, n, _) in outs:
            th = threading.Thread(target=orc.lib.orc_decode_frames,
                                  args=(blob.ctypes.data, o.ctypes.data, l.ctypes.data, n, dst.ctypes.data,
                                        doff.ctypes.data, dcap.ctypes.data, olen.ctypes.data, st.ctypes.data))
            th.start()
            ths.append(th)
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        ok = all(int(x[2].max()) == 0 and bool((x[1] == x[8]).all()) for x in outs)
        return dt, ok

    calib = min(n_total, 8 * cores)
    dt, ok = run(0, calib, cores)
    rate = calib / max(dt, 1e-6)  # frames/s on all cores
    sample = int(max(calib, min(n_total, rate * budget_s)))
    dt, ok2 = run(0, sample, cores)
    if sample == n_total and dt < 0.5 * budget_s:  # fast host: repeat the whole batch to fill the budget
        reps = max(1, int(budget_s / max(dt, 1e-3)))
        t0 = time.perf_counter()
        for _ in range(reps):
            _, okr = run(0, sample, cores)
            ok2 = ok2 and okr
        dt = (time.perf_counter() - t0) / reps
    # the same port on ONE thread ("(i) 1 thread, (ii) all host cores")
    n1 = int(max(8, min(n_total, rate / max(cores, 1) * min(3.0, budget_s / 3))))
    dt1, ok1 = run(0, n1, 1)
    res = {"value": round(csum[sample] / dt / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
           "sample": f"first {sample} frames of the same batch, oracle (C restatement of the reference "
                     f"algorithm) on {cores} host threads ({os.cpu_count()} hardware threads, cgroup quota "
                     f"{quota}), {dt:.2f}s per pass", "ok": bool(ok and ok2 and ok1),
           "one_thread": {"value": round(csum[n1] / dt1 / 1e6, 1), "unit": "MB/s", "cores": 1,
                          "sample": f"first {n1} frames, {dt1:.2f}s"}}
    res["libzstd"] = libzstd_line(blob, off, ln, exp_len, n1, cores)
    if cks is not None:
        # The batch pinned to the ORACLE at the sample's full size (untimed, one more pass): the word sum of every frame the oracle
        # regenerates against the generator's figure for the original content -- the same figure the device's output is held to,
        # so device == generator == oracle, frame by frame.  (A position-weighted sum is a detector, not a hash.)
        st, ol, ws = orc.decode_frames_wsum(blob, off[:sample], ln[:sample], cap, threads=cores)
        good = bool((st == 0).all() and (ol == exp_len[:sample]).all() and (ws == np.asarray(cks[:sample], dtype=np.uint64)).all())
        res["oracle_frames_checked"] = int(sample)
        res["oracle_bytes_checked"] = int(csum[sample])
        res["oracle_equals_generator"] = good
        res["ok"] = bool(res["ok"] and good)
    return res


def libzstd_line(blob, off, ln, exp_len, n1, cores):
    """Context only: libzstd through dlopen if the box has one -- third-party, NOT the
    reference and not a port of it; null when absent."""
    import ctypes
    try:
        Z = ctypes.CDLL("libzstd.so.1")
        Z.ZSTD_decompress.restype = ctypes.c_size_t
        Z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        Z.ZSTD_isError.argtypes = [ctypes.c_size_t]
        Z.ZSTD_versionNumber.restype = ctypes.c_uint
    except (OSError, AttributeError):
        return None

    cap = int(exp_len.max())
    csum = np.concatenate([[0], np.cumsum(exp_len.astype(np.float64))])

    def run(first, count, nthreads):
        per = (count + nthreads - 1) // nthreads
        oks = []

        def work(a, b):
            dst = np.zeros(cap + 64, dtype=np.uint8)
            good = True
            for i in range(a, b):
                r = Z.ZSTD_decompress(dst.ctypes.data, dst.size, blob.ctypes.data + int(off[i]), int(ln[i]))
                good = good and r == int(exp_len[i])
            oks.append(good)
        ths = [threading.Thread(target=work, args=(first + t * per, min(first + count, first + (t + 1) * per))) for t in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return time.perf_counter() - t0, all(oks)
    n = min(len(off), max(8, n1))
    dt1, ok1 = run(0, n, 1)
    nall = min(len(off), n * cores)
    dta, oka = run(0, nall, cores)
    return {"version": int(Z.ZSTD_versionNumber()), "one_thread_MBs": round(csum[n] / dt1 / 1e6, 1),
            "all_threads_MBs": round(csum[nall] / dta / 1e6, 1), "cores": cores, "ok": bool(ok1 and oka),
            "note": "context only: third-party libzstd via dlopen, not the reference"}


def load_corpus(gib, world=1):
    """The reference's own golden frames (tests/golden/decodecorpus, committed with their sha256 / length manifest): one REPLICA =
    the 100 frames in order; a batch is as many replicas as give `gib` GiB of output."""
    gdir = os.path.join(ROOT, "tests", "golden", "decodecorpus")
    manifest = json.load(open(os.path.join(gdir, "manifest.json")))
    names = sorted(manifest)
    cframes = [np.frombuffer(open(os.path.join(gdir, n + ".zst"), "rb").read(), dtype=np.uint8) for n in names]
    clens = np.array([manifest[n]["length"] for n in names], dtype=np.uint64)
    reps_total = max(world, int(gib * 2**30 / float(clens.sum()) + 0.5))
    return {"names": names, "sha": [manifest[n]["sha256"] for n in names], "lens": clens, "frames": cframes, "reps": reps_total}


def corpus_batch(corpus, reps):
    """`reps` replicas of the corpus as one host blob -> (blob, off, ln, exp_len)"""
    one = np.concatenate(corpus["frames"])
    o1 = np.concatenate([[0], np.cumsum([f.size for f in corpus["frames"]])[:-1]]).astype(np.uint64)
    l1 = np.array([f.size for f in corpus["frames"]], dtype=np.uint64)
    blob = np.tile(one, reps)
    off = np.concatenate([o1 + np.uint64(r * one.size) for r in range(reps)])
    return blob, off, np.tile(l1, reps), np.tile(corpus["lens"], reps)


def verify_corpus(torch, d_out, rb, corpus, exp_len, n_frames):
    """sha256 against the manifest: the whole first replica and one frame of every other replica (a different one each time),
    read back from their slabs in HBM -> (ok, frames checked)"""
    lay = rb.frame_layout()[0]
    nf = len(corpus["names"])
    sample = list(range(nf)) + [r * nf + (r * 37) % nf for r in range(1, n_frames // nf)]
    ok = True
    for i in sample:
        o, n = int(lay[i]), int(exp_len[i])
        got = d_out[o:o + n].cpu().numpy().tobytes()
        ok = ok and hashlib.sha256(got).hexdigest() == corpus["sha"][i % nf]
    return ok, len(sample)


def verify_synth(torch, d_out, per, frame_bytes, cks):
    """every frame of a synthetic batch against the checksum the generator took from its ORIGINAL content (a weighted sum of its
    64-bit words), computed on the device from the slabs (frames of one size lie back to back)"""
    exp = torch.from_numpy(cks.view(np.int64)).cuda()
    words = frame_bytes // 8
    wts = (2 * torch.arange(words, dtype=torch.int64, device="cuda") + 1)
    o64 = d_out[:per * frame_bytes].view(torch.int64).view(per, words)
    chunk = max(1, min(4096, (1 << 28) // words))
    ok = True
    for c in range(0, per, chunk):
        got = (o64[c:c + chunk] * wts).sum(dim=1)
        ok = ok and bool((got == exp[c:c + chunk]).all())
    return ok


def secondary_workloads(z, sb, torch, device, headline):
    """What the driver's ONE default run also measures, after the headline and outside its timed region: the other BASELINE
    configs at their stated sizes, real data, the shard one of eight GPUs gets, one large frame.  Same process, same library,
    the library's own kernel choices (default options), each bit-exact on a pass into a POISONED output blob, each with the
    copy ceiling measured for its own C and D bytes.  -> {name: {...}}"""
    out = {}
    threads = max(1, usable_cores()[0])

    def measure(name, blob, off, ln, exp_len, check, steps=10, warmup=2, note=None):
        t_all = time.perf_counter()
        ctx = z.Context(device)
        rb = None
        try:
            plan = z.Plan(device_tables=True)
            assert plan.add_frames(blob, off, ln, threads=threads) == 0
            batch = plan.finalize()
            pad = 64
            d_in = torch.zeros(blob.size + 2 * pad, dtype=torch.uint8, device="cuda")
            d_in[pad:pad + blob.size].copy_(torch.from_numpy(blob))
            d_out = torch.zeros(batch.out_size, dtype=torch.uint8, device="cuda")
            rb = ctx.upload(batch, device_in_ptr=d_in.data_ptr() + pad, device_out_ptr=d_out.data_ptr())
            stats = rb.stats()
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(warmup):
                rb.run(stream)
            torch.cuda.synchronize()
            # the steps that are timed carry no per-kernel events (an event record costs its stream 3-5 us, a pass has nine of them:
            # 8 % of a 0.5 ms pass); the per-kernel figures come from a second, instrumented loop of the same length
            ctx.timing_reset(False)
            t0 = time.perf_counter()
            for _ in range(steps):
                rb.run(stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            ctx.timing_reset(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                rb.run(stream)
            torch.cuda.synchronize()
            ms_events = (time.perf_counter() - t0) / steps * 1e3
            kms = ctx.kernel_ms()
            ctx.timing_reset(False)
            d_out.fill_(0xA5)  # the pass that is verified writes into poison
            torch.cuda.synchronize()
            rb.run(stream)
            torch.cuda.synchronize()
            _, status, out_len = rb.download(want_out=False)
            ok = bool((status == 0).all() and (out_len == exp_len).all())
            ok = ok and bool(check(d_out, rb))
            c_bytes, d_bytes = int(stats.compressed_bytes), int(exp_len.sum())
            path_ms = kms.pop("path", None) or sum(v for v in kms.values() if v > 0)
            achieved = (c_bytes + d_bytes) / (path_ms * 1e-3) / 1e9 if path_ms > 0 else None
            ceil = None
            try:
                cms = ctx.measure_copy(c_bytes, d_bytes, 10)
                ceil = round((c_bytes + d_bytes) / (cms * 1e-3) / 1e9, 1)
            except Exception:  # noqa: BLE001
                pass
            out[name] = {"ms_per_step": round(ms, 4), "ms_per_step_with_kernel_events": round(ms_events, 4), "path_ms": round(path_ms, 4), "steps": steps, "frames": int(len(off)),
                         "decompressed_bytes": d_bytes, "compressed_bytes": c_bytes,
                         "decompressed_MBs": round(d_bytes / (ms * 1e-3) / 1e6, 1),
                         "achieved_GBs": round(achieved, 1) if achieved else None,
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "copy_ceiling_GBs": ceil, "frac_of_copy_ceiling": round(achieved / ceil, 4) if achieved and ceil else None,
                         "kernel_ms": {k: round(v, 4) for k, v in kms.items() if v > 0},
                         # (block mode's passes are k_exec_c<true> by default; the flag beside it names the serial executor it would fall back to)
                         "library_chose": [n for n, bit in (("block mode", 2), ("k_exec_c", 4), ("k_exec_b", 8), ("split pass", 16), ("two groups of frames", 32))
                                           if rb.last_pass() & bit and not (bit == 8 and rb.last_pass() & 2)],
                         "bit_exact": ok, "poisoned_output": True, "wall_s": round(time.perf_counter() - t_all, 2)}
            if note:
                out[name]["note"] = note
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if rb is not None:
                rb.free()
            ctx.close()

    def synth(name, config, frames, frame_bytes=131072, **kw):
        t0 = time.perf_counter()
        blob, off, ln, cks, _ = sb.make_batch(config, 0, frames, frame_bytes, threads=threads)
        gen = time.perf_counter() - t0
        measure(name, blob, off, ln, np.full(frames, frame_bytes, dtype=np.uint64),
                lambda d_out, rb: verify_synth(torch, d_out, frames, frame_bytes, cks), **kw)
        out[name]["generate_s"] = round(gen, 2)
        return blob, off, ln, cks

    def in_chunks(name, frame, frame_bytes, ck, window_log=27, chunk_mib=64):
        """HOST TO HOST (PCIe is in it: not a device pass like the entries above): one frame through mzd_fstream_* in chunks of whole
        blocks -- the device keeps the frame's window (here 2^window_log bytes: the single-segment frame re-headed with a window
        descriptor, frame.go:28-36) and the scratch of two chunks, not the frame; pinned buffers; the first pass copies every chunk
        into one buffer and is checked against the generator's checksum, the second is the one timed."""
        t_all = time.perf_counter()
        ctx = z.Context(device)
        src = dst = whole = None
        try:
            fhd = int(frame[4])
            assert fhd & 0x20 and (fhd >> 6) >= 1
            comp = np.concatenate([frame[:4], np.array([fhd & ~0x20 & 0xFF, (window_log - 10) << 3], dtype=np.uint8), frame[5:]])
            src, dst, whole = z.PinnedBuffer(comp.size), z.PinnedBuffer(chunk_mib << 20), z.PinnedBuffer(frame_bytes)
            src.a[:] = comp
            secs, ok, calls, stages = None, False, 0, None
            for timed in (False, True):
                fs = z.FrameStream(ctx, chunk_mib << 20)
                pos = total = calls = 0
                t0 = time.perf_counter()
                while not fs.done:
                    used, made = fs.next(src.a[pos:], dst.a)
                    if not timed:
                        whole.a[total:total + made] = dst.a[:made]
                    pos += used
                    total += made
                    calls += 1
                secs = time.perf_counter() - t0
                stages = fs.timing()
                fs.close()
                if not timed:
                    ok = total == frame_bytes and sb.checksum64(whole.a[:frame_bytes]) == int(ck)
            out[name] = {"host_to_host": True, "ms": round(secs * 1e3, 2), "out_GBs": round(frame_bytes / secs / 1e9, 2), "chunk_MiB": chunk_mib,
                         "window_log": window_log, "calls": calls, "host_ms_per_stage": stages, "bit_exact": bool(ok),
                         "wall_s": round(time.perf_counter() - t_all, 2)}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            for b in (src, dst, whole):
                if b is not None:
                    b.free()
            ctx.close()

    synth("config2_4096_raw_rle_frames", 2, 4096, steps=20)
    synth("config3_4096_huffman_only_frames", 3, 4096, steps=20)
    synth("config3_65536_huffman_only_frames", 3, 65536, steps=5)
    # the 8192-frame shard of configs[4]: what ONE of eight GPUs runs when the headline batch is split (its first 8192 frames)
    blob, off, ln, cks = headline
    n = min(8192, len(off))
    end = int(off[n - 1] + ln[n - 1])
    measure("config4_shard_8192_frames_of_8_gpus", np.ascontiguousarray(blob[:end]), off[:n], ln[:n], np.full(n, 131072, dtype=np.uint64),
            lambda d_out, rb: verify_synth(torch, d_out, n, 131072, cks[:n]), steps=20)
    corpus = load_corpus(1.0)
    cb, co, cl, ce = corpus_batch(corpus, corpus["reps"])
    measure("decodecorpus_1GiB_real_data", cb, co, cl, ce, lambda d_out, rb: verify_corpus(torch, d_out, rb, corpus, ce, len(co))[0], steps=5,
            note=f"the reference's {len(corpus['names'])} golden frames x {corpus['reps']} replicas; sha256 of the first replica and one frame of every other")
    del cb, co, cl, ce
    corpus4 = load_corpus(4.0)  # (the size profiles/r*_corpus_* are measured at: the sequence stage bound by its work, frames in one group)
    cb, co, cl, ce = corpus_batch(corpus4, corpus4["reps"])
    measure("decodecorpus_4GiB_real_data", cb, co, cl, ce, lambda d_out, rb: verify_corpus(torch, d_out, rb, corpus4, ce, len(co))[0], steps=5,
            note=f"the reference's {len(corpus4['names'])} golden frames x {corpus4['reps']} replicas; sha256 of the first replica and one frame of every other")
    del cb, co, cl, ce
    synth("one_frame_256MiB_block_mode", 4, 1, 268435456, steps=3, warmup=1)
    blob, off, ln, cks = synth("one_frame_1GiB_block_mode", 4, 1, 1073741824, steps=3, warmup=1)  # (the reference's own usage: one big frame per reader)
    # ... and the same frame the reference's way (ABI 9): block by block through a window, here chunk by chunk
    in_chunks("one_frame_1GiB_in_chunks_of_64MiB_host_to_host", blob[int(off[0]):int(off[0] + ln[0])], 1073741824, cks[0])
    return out


def rank_share(a, rank, world, corpus_reps=None):
    """What rank `rank` of `world` decodes: -> (first unit, units of this rank, scaling, units of the whole job's base batch).  The unit
    is the frame (the replica of the corpus for --workload corpus).  Default = BASELINE configs[4]: ONE batch of `base` units cut
    into contiguous ranges, the remainder to the first ranks (sparkzstd_amd/sharding.py: the split the library's own multi-GPU
    entry makes; framedecompressor.go:42-52 is why it is legal) -- "strong"; --weak: `base` units on EVERY rank."""
    base = a.frames_per_gpu or (65536 if a.config == 4 else 4096)
    if corpus_reps is not None:
        base = corpus_reps
    if a.weak and not a.strong:
        return rank * base, base, "weak", base
    from sparkzstd_amd.sharding import frame_range
    first, end = frame_range(base, rank, world)
    return first, end - first, "strong", base


def gen_threads_for(a, world, cores):
    """host threads a rank generates / plans its frames with: the ranks of one node share its cores"""
    return a.gen_threads or max(1, cores // max(1, world))


def per_gpu_rows(gathered):
    """the ranks' own figures (rank, device, frames, ms per step, path ms, algorithmic GB/s, C bytes, D bytes -- one 8-vector per
    rank, as all_gather hands them over) -> (rows for the line's `per_gpu`, C bytes of all ranks, D bytes of all ranks)"""
    rows = [{"rank": int(g[0]), "device": int(g[1]), "frames": int(g[2]), "ms_per_step": round(float(g[3]), 4),
             "path_ms": round(float(g[4]), 4), "algorithmic_GBs": round(float(g[5]), 1),
             "hbm_frac": round(float(g[5]) / HBM_PEAK_GBS, 4)} for g in gathered]
    return rows, sum(int(g[6]) for g in gathered), sum(int(g[7]) for g in gathered)


def job_figures(rows, c_bytes_all, d_bytes_all, elapsed, steps, world):
    """whole-job figures from the slowest rank's clock: value = regenerated bytes of ALL ranks / max-over-ranks time"""
    ms = elapsed / steps * 1e3
    return {"value": d_bytes_all / (elapsed / steps) / 1e6, "ms_per_step": ms,
            "aggregate": {"algorithmic_GBs": round((c_bytes_all + d_bytes_all) / (ms * 1e-3) / 1e9, 1),
                          "hbm_frac_of_all_gpus": round((c_bytes_all + d_bytes_all) / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
                          "slowest_rank": max(rows, key=lambda r: r["ms_per_step"])["rank"],
                          "devices": sorted(r["device"] for r in rows),
                          "devices_distinct": len({r["device"] for r in rows}) == len(rows)}}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"WORLD_SIZE {world} != --gpus {a.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    # test hook (tests of the N>1 control flow on a 1-GPU box): MZD_BENCH_DEVICE=0 puts every rank on one GPU
    backend = a.rendezvous
    if "MZD_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MZD_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import sparkzstd_amd as z
    from tools import synth_binding as sb

    frame_bytes = a.frame_bytes
    if a.window_log:
        sb.set_max_offset(1 << a.window_log)
    assert frame_bytes % 256 == 0 and frame_bytes > 0
    corpus = None
    if a.workload == "corpus":
        corpus = load_corpus(a.corpus_gib, world)
        assert not a.frames_per_gpu, "--workload corpus sizes the batch with --corpus-gib"
    first, per, scaling, base = rank_share(a, rank, world, corpus["reps"] if corpus else None)  # (the corpus is split by replicas)

    # ---- synthetic batch (host), planning (host), upload: all outside the timed region
    t0 = time.perf_counter()
    gen_threads = gen_threads_for(a, world, usable_cores()[0])
    if corpus:
        reps = per
        blob, off, ln, exp_len = corpus_batch(corpus, reps)
        cks, nseq, distinct = None, np.zeros(1), len(corpus["names"])
        per = reps * len(corpus["names"])  # frames of this rank from here on
    # calibrate, then generate as many DISTINCT frames as the time budget allows (normally all of them)
    sb.set_content_checksum(a.verify_checksum)
    calib = 0 if corpus else min(per, 8 * gen_threads)
    if not corpus:
        tc = time.perf_counter()
        sb.make_batch(a.config, first, calib, frame_bytes, threads=gen_threads)
        rate = calib / max(time.perf_counter() - tc, 1e-3)  # frames per second on this rank's threads
        distinct = per
        while distinct > 1024 and distinct / rate > a.gen_seconds:
            distinct //= 2
        blob, off, ln, cks, nseq = sb.make_batch(a.config, first, distinct, frame_bytes, threads=gen_threads)
        exp_len = np.full(per, frame_bytes, dtype=np.uint64)
    if not corpus and distinct < per:
        # physical replication: content repeats, HBM addresses do not (SURVEY 8d option iii)
        reps = (per + distinct - 1) // distinct
        blob = np.ascontiguousarray(blob)
        base_len = int(blob.size)
        blob = np.tile(blob, reps)
        off = np.concatenate([off + np.uint64(r * base_len) for r in range(reps)])[:per]
        ln = np.tile(ln, reps)[:per]
        cks = np.tile(cks, reps)[:per]
        nseq = np.tile(nseq, reps)[:per]
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan = z.Plan(device_tables=not a.host_tables)
    rc = plan.add_frames(blob, off, ln, threads=gen_threads)
    assert rc == 0, f"planner failed: {rc}"
    batch = plan.finalize()
    t_plan = time.perf_counter() - t0
    assert batch.n_frames == per and (corpus or batch.out_size == per * frame_bytes + 256)
    my_d_bytes = int(exp_len.sum())  # regenerated bytes of this rank's frames

    t0 = time.perf_counter()
    pad = 64
    d_in = torch.zeros(blob.size + 2 * pad, dtype=torch.uint8, device="cuda")
    d_in[pad:pad + blob.size].copy_(torch.from_numpy(blob))
    d_out = torch.zeros(batch.out_size, dtype=torch.uint8, device="cuda")
    ctx = z.Context(local_rank, seq_variant=a.seq_variant, exec_threads=a.exec_threads, exec_chunk=a.exec_chunk, huf_min_lds=a.huf_min_lds, no_split=a.no_split,
                    verify_checksum=a.verify_checksum, huf_variant=a.huf_variant, exec_variant=a.exec_variant)
    rb = ctx.upload(batch, device_in_ptr=d_in.data_ptr() + pad, device_out_ptr=d_out.data_ptr())
    torch.cuda.synchronize()
    t_upload = time.perf_counter() - t0
    stats = rb.stats()
    t_dplan = None
    if a.device_plan:
        # the same batch planned ON THE DEVICE: the host planner's batch above only serves as the cross-check
        host_stats = stats
        rb.free()
        t0 = time.perf_counter()
        rb = ctx.upload_frames(int(blob.size), off, ln, device_in_ptr=d_in.data_ptr() + pad, device_out_ptr=d_out.data_ptr(),
                               device_out_size=int(d_out.numel()))
        t_dplan = time.perf_counter() - t0
        stats = rb.stats()
        assert rb.out_size == batch.out_size and stats.compressed_bytes == host_stats.compressed_bytes
        assert stats.n_sequences == host_stats.n_sequences
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup
    for _ in range(a.warmup):
        rb.run(stream)
    torch.cuda.synchronize()
    ctx.timing_reset(True)

    # ---- timed region: exactly K steps between barrier + synchronize on both sides
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rb.run(stream)
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0  # this rank's own K steps (before it waits for the others)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms = ctx.kernel_ms()  # HIP events on the launch stream, averaged over the K timed steps

    # ---- verification (outside the timed region): status, lengths, checksum of every frame.  What is verified is the
    # output of ONE MORE pass into a POISONED output blob (every byte 0xA5, statuses and lengths overwritten by the pass
    # itself): a pass that silently wrote nothing after the first one would leave poison behind, not the earlier passes'
    # bytes.  (ctx.timing_reset keeps this extra pass out of the per-kernel averages.)
    ok = True
    if not a.no_verify:
        ctx.timing_reset(False)
        d_out.fill_(0xA5)
        torch.cuda.synchronize()
        rb.run(stream)
        torch.cuda.synchronize()
        _, status, out_len = rb.download(want_out=False)
        ok = bool((status == 0).all() and (out_len == exp_len).all())
    if not a.no_verify and corpus:
        # every frame by status and length (above); by sha256 against the manifest: a sample (verify_corpus)
        okc, corpus["sha_checked"] = verify_corpus(torch, d_out, rb, corpus, exp_len, per)
        ok = ok and okc
    if not a.no_verify and not corpus:
        ok = ok and verify_synth(torch, d_out, per, frame_bytes, cks)
    if world > 1:
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())

    # every rank's own figures (frames, own wall time per step, own path time from its HIP events, algorithmic GB/s)
    kms_all = dict(kms)
    my_path_ms = kms_all.get("path") or sum(v for k, v in kms_all.items() if v > 0)
    my_alg = int(stats.compressed_bytes) + my_d_bytes
    mine = torch.tensor([float(rank), float(torch.cuda.current_device()), float(per), my_elapsed / a.steps * 1e3, my_path_ms,
                         my_alg / (my_path_ms * 1e-3) / 1e9 if my_path_ms > 0 else 0.0, float(int(stats.compressed_bytes)), float(my_d_bytes)],
                        dtype=torch.float64, device=red_dev)
    if world > 1:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    else:
        gathered = [mine]
    per_gpu, c_bytes_all, d_bytes_all = per_gpu_rows([g.tolist() for g in gathered])
    if world > 1 and "MZD_BENCH_DEVICE" not in os.environ:
        # one process per GPU means one GPU per process: two ranks on one device would halve each other and still print a line
        assert len({p["device"] for p in per_gpu}) == world, f"ranks share a device: {[p['device'] for p in per_gpu]}"

    # measured copy ceiling: a plain 16 B/lane streaming kernel that reads C and writes D bytes (this rank's)
    ceiling = None
    if rank == 0 and not a.no_ceiling:
        rb_bytes, wb_bytes = int(stats.compressed_bytes), my_d_bytes
        try:
            cms = ctx.measure_copy(rb_bytes, wb_bytes, 10)
            ceiling = {"GBs": round((rb_bytes + wb_bytes) / (cms * 1e-3) / 1e9, 1), "ms": round(cms, 4),
                       "kernel": "k_copy_ceiling: plain 16 B/lane streaming copy, same C bytes read + D bytes written"}
        except Exception as e:  # noqa: BLE001
            ceiling = {"error": str(e)}

    if rank == 0:
        total_frames = sum(p["frames"] for p in per_gpu)
        d_bytes = d_bytes_all
        job = job_figures(per_gpu, c_bytes_all, d_bytes_all, elapsed, a.steps, world)
        ms_per_step, value = job["ms_per_step"], job["value"]
        # roofline of the dominant kernel set: algorithmic bytes = compressed bytes read once +
        # decompressed bytes written once, per launch (= one pass over this rank's batch)
        c_bytes = int(stats.compressed_bytes)
        alg = c_bytes + my_d_bytes
        path_ms = kms.pop("path", None) or sum(v for k, v in kms.items() if v > 0)
        xxh_ms = kms.pop("k_xxh64", None)  # optional extension, reported on its own below
        dom = max(kms, key=lambda k: kms[k]) if kms else None
        achieved = alg / (path_ms * 1e-3) / 1e9 if path_ms > 0 else None
        # HBM traffic per launch from PMC passes (FETCH_SIZE + WRITE_SIZE, separate --pmc runs, tools/profile_round.sh):
        # quoted only when the file was measured on THIS device code and THIS workload
        wl_tag = "corpus" if corpus else f"cfg{a.config}"

        def stamped(path, what):
            """a counter file of the builder's (tools/profile_round.sh / profile_counters.sh), quoted only when it was measured on
            THIS device code and THIS workload; -> (json or None, note that says whose number it is and from when)"""
            try:
                j = json.load(open(path))
            except FileNotFoundError:
                return None, f"no {what} file for this workload ({os.path.basename(path)})"
            except Exception as e:  # noqa: BLE001
                return None, f"unreadable {what} file: {e}"
            if j.get("kernel_src_sha16") != kernel_src_sha16():
                return None, f"{os.path.basename(path)} was measured on other device code ({j.get('kernel_src_sha16')}): not quoted"
            same = j.get("workload", "synthetic") == a.workload and (
                (corpus and j.get("corpus_gib", 4.0) == a.corpus_gib) or
                (not corpus and j.get("frames_per_gpu") == per and j.get("config") == a.config and j.get("frame_bytes") == frame_bytes))
            if not same:
                return None, f"{os.path.basename(path)} is for another workload: not quoted"
            m = j.get("measured", {})
            return j, (f"{os.path.basename(path)}: measured by {m.get('by', 'the builder')} on {m.get('date', 'an earlier day')}, NOT in this run; "
                       f"same device sources ({j.get('kernel_src_sha16')}) and workload")

        traffic = None
        def newest(kind):
            """profiles/r<N>_<kind>_<workload>.json of the latest round that has one"""
            import glob
            import re
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_{wl_tag}.json")),
                           key=lambda p: int(re.match(r"r(\d+)_", os.path.basename(p)).group(1)) if re.match(r"r(\d+)_", os.path.basename(p)) else -1)
            return found[-1] if found else os.path.join(ROOT, "profiles", f"r5_{kind}_{wl_tag}.json")

        tj, traffic_note = stamped(a.traffic_from or newest("traffic"), "traffic")
        if tj:
            traffic = sum(v["fetch_bytes"] + v["write_bytes"] for k, v in tj["kernels"].items() if k != "k_init")
            traffic_note += "; FETCH_SIZE (raw; gfx950 under-counts wide streaming reads up to 2x) + WRITE_SIZE summed over the pass's kernels"
        # what each kernel keeps busy inside the CU: the path is latency- and issue-bound, an HBM fraction alone does not show progress
        ij, issue_note = stamped(a.issue_from or newest("issue"), "issue")
        issue = {"source": issue_note}
        if ij:
            issue["kernels"] = {k: {f: v[f] for f in ("valu_wave_insts", "valu_frac", "lds_pipe_frac", "ta_busy_frac", "l2_hit_frac", "kernel_cycles") if f in v}
                                for k, v in ij["kernels"].items()}
            issue["note"] = ij.get("note")
        roof = {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                "traffic": traffic, "traffic_source": traffic_note, "issue": issue,
                "copy_ceiling": ceiling,
                "frac_of_copy_ceiling": round(achieved / ceiling["GBs"], 4) if achieved and ceiling and ceiling.get("GBs") else None,
                "kernel": f"hot path = k_huf -> k_seq -> k_exec (k_seq tail round overlaps k_exec of the head frames); "
                          f"achieved = algorithmic bytes / path time; dominant {dom}",
                "path_ms": round(path_ms, 4),
                "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
                "algorithmic_bytes_per_launch": alg,
                "dominant_kernel_alone_GBs": round(alg / (kms[dom] * 1e-3) / 1e9, 1) if dom else None}
        if xxh_ms:
            # k_xxh64 streams the regenerated bytes once: its own HBM roofline (D bytes / its time)
            roof["k_xxh64"] = {"ms": round(xxh_ms, 4), "achieved_GBs": round(my_d_bytes / (xxh_ms * 1e-3) / 1e9, 1),
                               "frac": round(my_d_bytes / (xxh_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        cpu = None
        if a.cpu_seconds > 0:
            cpu = cpu_baseline(blob, off, ln, exp_len, a.cpu_seconds, None if corpus else cks)
        names = {2: "config2 raw/rle single-block 128KiB frames", 3: "config3 4-stream huffman literals, 0 sequences",
                 4: "config4 text-like 128KiB frames: huffman literals + FSE sequences + match copy"}
        line = {
            "metric": "decompressed MB/s + %HBM-peak, 64k-frame batch, 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u8",
            "data": "real: the reference's decodecorpus frames (tests/golden/decodecorpus), replicated" if corpus else "synthetic",
            "config": {"workload": (f"decodecorpus: the reference's {len(corpus['names'])} golden frames x {base} replicas at distinct HBM addresses "
                                    f"(content repeats, addresses do not); {corpus.get('sha_checked', 0)} frames checked by sha256, all by status and length")
                       if corpus else names.get(a.config, str(a.config)), "frames": total_frames, "frames_per_gpu": per,
                       "frame_bytes": None if corpus else frame_bytes, "window_log": a.window_log or None, "decompressed_bytes_all_gpus": d_bytes_all,
                       "compressed_bytes_per_gpu": c_bytes, "compressed_bytes_all_gpus": c_bytes_all,
                       "sequences_per_frame": round(float(stats.n_sequences) / max(per, 1), 1), "distinct_frames_per_gpu": distinct,
                       "parallelism": f"one batch of {total_frames} frames in contiguous ranges over {world} GPU(s), no collective"
                                      if scaling == "strong" else f"{per} frames on each of {world} GPU(s), no collective",
                       "rendezvous": backend if world > 1 else None,
                       "seq_variant": a.seq_variant, "exec_threads": a.exec_threads or 128,
                       "exec_chunk": a.exec_chunk or 8192},
            "roofline": roof, "cpu_baseline": cpu, "bit_exact": ok, "ranks_seen": len(per_gpu), "per_gpu": per_gpu,
            # (all ranks together, on the slowest rank's clock: what SCALE's N = 1 line is compared with BENCH's by, and N > 1 with N = 1)
            "aggregate": job["aggregate"],
            "kernel_src_sha16": kernel_src_sha16(),
            # (the library reports the hash of the sources it was BUILT from; the tree's sources hashed now: a stale library shows)
            "library_matches_tree_sources": (library_src_sha16() == kernel_src_sha16()) if library_src_sha16() else None,
            "hbm_peak_frac_decompressed": round(value / 1e3 / world / HBM_PEAK_GBS, 4),
            "exec_variant": a.exec_variant,
            "setup_s": {"generate": round(t_gen, 2), "plan": round(t_plan, 3), "upload": round(t_upload, 3),
                        "tables": "host" if a.host_tables else "device",
                        "headers": "device" if a.device_plan else "host",
                        "device_plan_s": round(t_dplan, 4) if t_dplan is not None else None,
                        "k_parse_ms": round(float(stats.parse_ms), 3) if a.device_plan else None,
                        "fse_tables_built_on_device": int(stats.n_fse_built),
                        "huf_tables_built_on_device": int(stats.n_huf_built),
                        "k_fse_build_ms": round(float(stats.fse_build_ms), 3)},
        }
        # (the plain default invocation only: the experiment and profiling scripts pass --cpu-seconds 0 --no-ceiling and get the
        # headline alone, so a rocprofv3 run of theirs sees the headline's kernels and nothing else)
        headline_default = (world == 1 and not corpus and a.config == 4 and not a.frames_per_gpu and frame_bytes == 131072 and
                            not a.window_log and not a.verify_checksum and distinct == per and a.cpu_seconds > 0 and not a.no_ceiling and
                            not a.exec_variant and not a.seq_variant and not a.huf_variant and not a.no_split)
        if headline_default and not a.no_secondary:
            # (the headline's batch and CONTEXT go first: the runtime deals a process's streams to four hardware queues, and a fifth
            # stream -- a second context's second stream -- shares one: the Huffman kernel then no longer runs beside the sequence
            # stage, real data 10.9 -> 13.9 ms; `tools/experiments/r5_sec_only.py`)
            rb.free()
            ctx.close()
            t0 = time.perf_counter()
            line["secondary"] = secondary_workloads(z, sb, torch, local_rank, (blob, off, ln, cks))
            line["secondary"]["wall_s_total"] = round(time.perf_counter() - t0, 1)
        print(json.dumps(line), flush=True)
    rb.free()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    # Under rocprofv3 the HIP/torch teardown after the profiler's own finalization can hang: exit
    # normally (so the profiler writes its output) but leave a detached watchdog behind.
    import subprocess
    subprocess.Popen(["sh", "-c", f"sleep 30; kill -9 {os.getpid()} 2>/dev/null"], start_new_session=True,
                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
