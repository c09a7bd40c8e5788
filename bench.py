#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: GiB/s end-to-end (compress+decompress), 4 MiB blocks.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  compress    lz4f_mi355x_dev_compressFrameIndexed(..., LZ4F_MI355X_INBAND): find_matches -> layout -> emit, then the trailer
              (the blocks' positions and the sequence index as a skippable frame behind the LZ4 frame, part of the byte stream)
  decompress  lz4f_mi355x_dev_decompressFrame on those bytes and nothing else: no block table, no side buffer.  It finds the
              trailer, checks it against the frame, parses per index entry, resolves direct matches, copies.
both through the C ABI on torch's current stream.  The same LZ4 frame WITHOUT its trailer - what a frame from liblz4 or the
`lz4` tool looks like to this decoder - is timed after the timed region (`foreign_frame`: seeded size-word walk, stretch-parallel self-index, indexed kernels).
Workload at every N: BASELINE configs[2] per GPU -- 4 GiB of synth50 (~50 % compressible), 4 MiB independent blocks; frame
blocks are independent, so ranks shard the stream with no data-path collective ("weak" scaling: every rank gets its own 4 GiB
with seed 1234+rank).  value = bytes of uncompressed input all ranks processed / max-over-ranks wall time of the K steps.

Extra objects on the JSON line (all outside the timed region):
  roofline       the kernel with the largest share of the step: algorithmic bytes (U + C per direction) /
                 its launch duration measured with HIP events on the launch stream (events of the last timed step)
  kernels        the same for every kernel of the step
  foreign_frame  decode of the bare LZ4 frame (walk included) and its roofline fraction
  block_checksum_on   the step with XXH32 block checksums written and verified on the GPU
  cfg2           BASELINE configs[1]: decompress-only, 1 GiB of text pre-framed by liblz4 at 64 KiB independent blocks, bare frame, walk included
  host_to_host   lz4f_mi355x_compressFrame / decompressFrame on host buffers (SURVEY 8d variant H), pageable and page-locked
  conduit_replay the reference's conduit call pattern (Conduit.hsc:457-533, :598-701: 16 KiB slices, default preferences)
                 through this library's twelve LZ4F_* functions, next to the CPU codec driven the same way on one thread
  cpu_baseline   the same blocks through liblz4 (dlopen, kind "reference") or the oracle port, on the host cores,
                 rank 0 at N=1 only, on a bounded sample
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
HBM_COPY_GBS = 6290.0
PCIE_GBS = 63.0                # MI355X_MICROARCH.md: PCIe Gen5 x16, 63 GB/s spec (53-54 GiB/s measured each way, tools/probe/pcie_rates.py)
PMC_PROFILE = "profiles/round3_pmc_traffic.json"


def pmc_traffic(kernel: str, n_bytes: int, block_size: int):
    """HBM bytes per launch of `kernel` from the committed PMC profile (rocprofv3 cannot run inside the timed process:
    counters are collected in their own passes), or None when that profile was not taken on this workload."""
    try:
        if n_bytes != (4 << 30) or block_size != (4 << 20):
            return None
        with open(os.path.join(ROOT, PMC_PROFILE)) as f:
            prof = json.load(f)
        return int(sum(prof["kernels"][k]["hbm_bytes_corrected"] for k in prof["groups"][kernel]))
    except Exception:
        return None


def cpu_baseline(block_size: int, sample_bytes: int):
    """oracle/orc_cpu_baseline on `sample_bytes` of the canonical (numpy) synth50 stream."""
    import oracle
    from lz4_frame_conduit_amd import datagen
    oracle.build()
    exe = os.path.join(ROOT, "oracle", "orc_cpu_baseline")
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("LZ4F_BENCH_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share is 16 cores
    data = datagen.synth50(sample_bytes, 1234)
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(d, "lz4f_bench_sample_%d.bin" % os.getpid())
    try:
        data.tofile(path)
        out = subprocess.run([exe, path, str(block_size), str(cores), "3"], capture_output=True, text=True, timeout=600)
        j = json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    u = j["bytes"]
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    e2e = u / (j["t_comp"] + j["t_decomp"]) / GIB
    return {"value": round(e2e, 3), "unit": "GiB/s", "cores": j["threads"], "kind": j["kind"],
            "sample": "%d MiB of synth50 (seed 1234), %d KiB independent blocks, block-parallel LZ4_compress_default + LZ4_decompress_safe, "
                      "best of 3 after warm-up; host: %s" % (u >> 20, block_size >> 10, cpu),
            "compress_GiBs": round(u / j["t_comp"] / GIB, 3), "decompress_GiBs": round(u / j["t_decomp"] / GIB, 3),
            "ratio": round(u / j["compressed"], 4), "roundtrip_ok": j["roundtrip_ok"]}


def leg(fn):
    """A side leg must never take the headline down."""
    try:
        return fn()
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=4 << 30, help="uncompressed bytes per GPU per step")
    ap.add_argument("--block-size-id", type=int, default=7, help="4=64KiB 5=256KiB 6=1MiB 7=4MiB")
    ap.add_argument("--block-checksum", type=int, default=0)
    ap.add_argument("--linked", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the side legs: what the rocprofv3 summaries under profiles/ are taken with, so that their per-kernel averages are the headline launches' only")
    ap.add_argument("--foreign", action="store_true", help="the timed decode gets the bare LZ4 frame (no trailer): walk + self-index + indexed kernels")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--legs", default="all", help="comma-separated side legs to run (foreign,bck,cfg2,linked,host,replay); what tools/prof_leg.sh profiles one at a time")
    args = ap.parse_args()

    # N > 1 and not yet under a launcher: start N ranks (one process per GPU) as a CHILD process and relay its output - nothing in
    # this process has touched the GPU yet (torch is not even imported).  Under a launcher WORLD_SIZE must be what --gpus says.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            import socket
            import torch                              # (counting devices does not initialise the GPU; the ranks are children of this process)
            if torch.cuda.device_count() < args.gpus and not os.environ.get("LZ4F_BENCH_SHARE_GPU"):
                sys.exit("bench.py: --gpus %d but %d GPU(s) are visible: one process per GPU, no oversubscription" % (args.gpus, torch.cuda.device_count()))
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
            env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("MASTER_ADDR", "127.0.0.1")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd, env=env).returncode)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]))

    import numpy as np
    import torch
    import torch.distributed as dist

    have = torch.cuda.device_count()              # (counting does not initialise the GPU)
    # LZ4F_BENCH_SHARE_GPU=1 (a test switch, tests/test_distributed_cpu.py): the ranks share the visible GPU(s), rank r on device r mod visible, and
    # meet over gloo (RCCL refuses two ranks on one device) - the launch, the rendezvous, the barrier and the max-over-ranks time run as they do
    # on N GPUs; the number it prints is not a measurement of anything
    share = bool(os.environ.get("LZ4F_BENCH_SHARE_GPU")) and have >= 1
    if have < args.gpus and not share:
        sys.exit("bench.py: --gpus %d but %d GPU(s) are visible: one process per GPU, no oversubscription" % (args.gpus, have))

    from lz4_frame_conduit_amd import _ffi, conduit, datagen, shard
    from lz4_frame_conduit_amd.device import Engine, synth50_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if share: local_rank = local_rank % have
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if share: dist.init_process_group("gloo")
        else: dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    n = args.bytes
    bs = 1 << (8 + 2 * args.block_size_id)
    prefs = conduit.make_preferences(blockSizeID=args.block_size_id, blockMode=0 if args.linked else 1, blockChecksum=args.block_checksum)
    nb = (n + bs - 1) // bs

    src = synth50_device(n, 1234 + rank, dev)
    eng = Engine(local_rank)
    inband = not args.foreign
    frame = torch.empty(eng.frame_bound_inband(n, prefs), dtype=torch.uint8, device=dev)
    back = torch.empty_like(src)
    eng.set_timing(True)

    def step():
        # compress; read the 32-byte result record (the stream's length: a host round trip, inside the step because a caller has
        # it too); decompress those bytes
        eng.compress_async(src, frame, prefs, inband=inband)
        size = int(eng.result().size)
        eng.decompress_frame_async(frame, size, back)

    def barrier():
        shard.barrier_all(torch.device(dev))

    def spoil(t):
        # Between a side leg's repetitions the output of the one before is spoiled - a byte in every 4 KiB, so that a decode that left any
        # block alone is seen by the comparison behind it - instead of zeroed: 4 GiB of fresh zeros are still on their way out of the caches
        # when the timed decode starts, and its time then moves by up to 0.6 ms from run to run (foreign frame: 2.39 / 3.03 / 2.61 ms
        # with zero_() against 2.33 +- 0.02 without).
        t[::4099] = 0xA5

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    back.zero_()
    torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, "cpu" if share else dev)
    kt = {}
    if rank == 0:                                             # per-kernel HIP-event times of the last timed step
        kt = dict(eng.get_timing())
    r2 = eng.result()
    ok = bool(r2.size == n and torch.equal(back, src))
    # the last timed step's own frame decodes to the input as well
    eng.compress_async(src, frame, prefs, inband=inband); rl = eng.result(); t_c = eng.get_timing()
    back.zero_()
    eng.decompress_frame_async(frame, int(rl.size), back); r3 = eng.result(); t_d = eng.get_timing()
    ok = ok and bool(r3.size == n and torch.equal(back, src))
    csize_stream = int(rl.size)
    frame_only = int(r3.consumed)                              # the LZ4 frame without its trailer
    if rank == 0:
        kt = {**{k: t_c[k] for k in ("find_matches", "layout", "emit", "xxh32_write")}, **{k: t_d[k] for k in ("walk", "xxh32_verify", "decode", "finish", "decode_parse", "decode_copy")}}

    side = {}
    legs = set(x.strip() for x in args.legs.split(","))
    def want(name): return "all" in legs or name in legs
    if rank == 0 and world == 1 and not args.headline_only:
        algo = float(n + frame_only)

        def foreign():
            t = []
            good = True
            for _ in range(3):
                spoil(back)
                eng.decompress_frame_async(frame, frame_only, back)          # the bare frame: nothing but LZ4
                rf = eng.result(); tt = eng.get_timing()
                t.append((tt["decompress_total"], tt["walk"], tt["decode"]))
                good = good and bool(rf.size == n and torch.equal(back, src))
            best = min(t)
            return {"what": "the same LZ4 frame without the trailer (as liblz4 / the lz4 tool would have written it), device-resident, no block table: "
                            "seeded size-word walk, then the decoder cuts the blocks into stretches itself (decode_spx.cuh) and runs the indexed kernels", "ms": round(best[0], 4), "walk_ms": round(best[1], 4), "decode_ms": round(best[2], 4),
                    "roundtrip_verified": good,
                    "roofline": {"bound": "hbm", "achieved": round(algo / (best[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(algo / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        if want("foreign"): side["foreign_frame"] = leg(foreign)

        def bck_on():
            p2 = conduit.make_preferences(blockSizeID=args.block_size_id, blockMode=1, blockChecksum=1)
            best = None
            good = True
            for _ in range(2):
                eng.compress_async(src, frame, p2, inband=True); rr = eng.result(); tc = eng.get_timing()
                spoil(back)
                eng.decompress_frame_async(frame, int(rr.size), back); rd = eng.result(); td = eng.get_timing()
                good = good and bool(rd.size == n and torch.equal(back, src))
                ms = tc["compress_total"] + td["decompress_total"]          # (whole calls: the verification runs beside the decode kernels, on a stream of its own)
                if best is None or ms < best[0]:
                    best = (ms, tc["xxh32_write"], td["xxh32_verify"], tc["compress_total"], td["decompress_total"])
            return {"what": "the headline step with an XXH32 behind every block, written and verified on the GPU (a 4 MiB block is one serial chain for one wave: ~2 ms "
                            "whatever else runs; the verification runs beside the decode kernels, the writing can only follow the last block's emission)",
                    "ms_per_step": round(best[0], 3), "xxh32_write_ms": round(best[1], 3), "xxh32_verify_ms": round(best[2], 3),
                    "compress_ms": round(best[3], 3), "decompress_ms": round(best[4], 3),
                    "e2e_GiBs": round(n / (best[0] * 1e-3) / GIB, 1), "roundtrip_verified": good}
        if want("bck"): side["block_checksum_on"] = leg(bck_on)

        def cfg2():
            # BASELINE configs[1]: "1 GiB enwik-style text pre-framed at 64 KiB independent blocks" - pre-framed by liblz4 (the reference's
            # codec): LZ4F_compressFrame of a 64 MiB tile of the text through the installed liblz4.so.1, or - where that is absent - through
            # the oracle port, which is bit-exact with it; the tile's blocks, sixteen times, between one header and one EndMark.
            m, tile_n = 1 << 30, 64 << 20
            tile = datagen.synth_text(tile_n, 99)
            framer = None
            try:
                lz = ctypes.CDLL("liblz4.so.1")
                lz.LZ4F_compressFrameBound.restype = ctypes.c_size_t; lz.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
                lz.LZ4F_compressFrame.restype = ctypes.c_size_t; lz.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
                lp = conduit.make_preferences(blockSizeID=4, blockMode=1)
                cap = lz.LZ4F_compressFrameBound(tile_n, ctypes.byref(lp))
                buf = ctypes.create_string_buffer(cap)
                r_ = lz.LZ4F_compressFrame(buf, cap, tile.ctypes.data_as(ctypes.c_void_p), tile_n, ctypes.byref(lp))
                if r_ < 64 or r_ > cap: raise OSError("LZ4F_compressFrame: %d" % r_)
                one = buf.raw[:r_]; framer = "liblz4.so.1 %d (LZ4F_compressFrame)" % lz.LZ4_versionNumber()
            except (OSError, AttributeError):
                import oracle
                one = oracle.conduit_compress(tile.tobytes(), oracle.mkprefs(bsid=4, indep=1)); framer = "oracle port of liblz4 1.9.3 (bit-exact with it: tests/test_oracle_golden.py)"
            assert one[:4] == b"\x04\x22\x4d\x18" and one[-4:] == bytes(4) and len(one) > 11
            body = np.frombuffer(one[7:-4], dtype=np.uint8)
            reps = m // tile_n
            host = np.empty(7 + len(body) * reps + 4 + 64, dtype=np.uint8)
            host[:7] = np.frombuffer(one[:7], dtype=np.uint8)
            for r_ in range(reps): host[7 + r_ * len(body): 7 + (r_ + 1) * len(body)] = body
            fsize = 7 + len(body) * reps + 4
            host[fsize - 4:] = 0
            f2 = torch.from_numpy(host).to(dev)
            tx = torch.from_numpy(tile).to(dev).repeat(reps)
            b2 = torch.empty(m, dtype=torch.uint8, device=dev)
            best = None
            good = True
            for _ in range(3):
                spoil(b2)
                eng.decompress_frame_async(f2, fsize, b2); rd = eng.result(); td = eng.get_timing()
                good = good and bool(rd.size == m and rd.consumed == fsize and torch.equal(b2, tx))
                ms = td["decompress_total"]
                if best is None or ms < best[0]:
                    best = (ms, td["walk"], td["decode"])
            a2 = float(m + fsize)
            # this library's own encoder on the same text, for the record (not what the leg decodes)
            p2 = conduit.make_preferences(blockSizeID=4, blockMode=1)
            f3 = torch.empty(eng.frame_bound(m, p2), dtype=torch.uint8, device=dev)
            eng.compress_async(tx, f3, p2); rc = eng.result(); tc = eng.get_timing()
            return {"workload": "BASELINE configs[1]: decompress-only, 1 GiB of synthetic text (Zipf words, liblz4 ratio %.3f), pre-framed at 64 KiB independent blocks by %s, bare LZ4 frame, "
                                "device-resident, no block table" % (m / fsize, framer),
                    "decompress_ms": round(best[0], 3), "walk_ms": round(best[1], 3), "decode_ms": round(best[2], 3),
                    "decompress_GiBs": round(m / (best[0] * 1e-3) / GIB, 1), "roundtrip_verified": good,
                    "own_encoder": {"compress_ms": round(tc["compress_total"], 3), "compress_GiBs": round(m / (tc["compress_total"] * 1e-3) / GIB, 1), "ratio": round(m / int(rc.size), 4)},
                    "roofline": {"bound": "hbm", "achieved": round(a2 / (best[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(a2 / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(a2)}}
        if want("cfg2"): side["cfg2"] = leg(cfg2)

        def linked_default():
            m = 1 << 30
            lp = conduit.make_preferences(blockSizeID=4, blockMode=0)
            lf = torch.empty(eng.frame_bound_inband(m, lp), dtype=torch.uint8, device=dev)
            lb = torch.empty(m, dtype=torch.uint8, device=dev)
            best = None
            good = True
            for _ in range(3):
                eng.compress_async(src[:m], lf, lp, inband=True); rc = eng.result(); tc = eng.get_timing()
                spoil(lb)
                eng.decompress_frame_async(lf, int(rc.size), lb); rd = eng.result(); td = eng.get_timing()
                good = good and bool(rd.size == m and torch.equal(lb, src[:m]))
                c_ms = tc["find_matches"] + tc["layout"] + tc["emit"]; d_ms = td["walk"] + td["decode"] + td["finish"]
                if best is None or c_ms + d_ms < best[0] + best[1]:
                    best = (c_ms, d_ms)
            return {"workload": "1 GiB of the same stream, 64 KiB LINKED blocks (Conduit.hsc default preferences), device-resident, decoded from the stream alone",
                    "compress_ms": round(best[0], 3), "decompress_ms": round(best[1], 3), "e2e_GiBs": round(m / ((best[0] + best[1]) * 1e-3) / GIB, 1), "roundtrip_verified": good}
        if want("linked"): side["reference_default_framing"] = leg(linked_default)

        def host_legs():
            L = _ffi.lib()
            m = 1 << 30
            data = src[:m].cpu().numpy()
            hp = conduit.make_preferences(blockSizeID=7, blockMode=1)
            bound = L.lz4f_mi355x_compressFrameBound(m, ctypes.byref(hp))
            out = {"what": "lz4f_mi355x_compressFrame / decompressFrame, 1 GiB of the stream, 4 MiB independent blocks, host buffers in and out; the link: PCIe Gen5 x16, "
                           "%.0f GB/s spec, 53-54 GiB/s measured each way - a round trip moves the uncompressed bytes over it twice, so <= ~26 GiB/s per GPU whatever the kernels do" % PCIE_GBS}

            def ptr(a):
                return a.ctypes.data_as(ctypes.c_void_p)
            for mode in ("pageable", "page_locked"):
                if mode == "pageable":
                    s_, d_, b_ = data, np.empty(bound, dtype=np.uint8), np.empty(m + 8, dtype=np.uint8)
                    free = []
                else:
                    free = [L.lz4f_mi355x_host_alloc(x) for x in (m, bound, m + 8)]
                    s_, d_, b_ = [np.ctypeslib.as_array((ctypes.c_uint8 * x).from_address(p)) for x, p in zip((m, bound, m + 8), free)]
                    s_[:] = data
                bc = bd = 1e9
                good = True
                try:
                    for it in range(3):
                        t0_ = time.perf_counter()
                        rr = L.lz4f_mi355x_compressFrame(ptr(d_), bound, ptr(s_), m, ctypes.byref(hp))
                        t1_ = time.perf_counter()
                        assert not L.LZ4F_isError(rr), L.LZ4F_getErrorName(rr)
                        used = ctypes.c_size_t(0)
                        t2_ = time.perf_counter()
                        r2_ = L.lz4f_mi355x_decompressFrame(ptr(b_), m + 8, ptr(d_), rr, ctypes.byref(used))
                        t3_ = time.perf_counter()
                        assert not L.LZ4F_isError(r2_), L.LZ4F_getErrorName(r2_)
                        if it:
                            bc = min(bc, t1_ - t0_); bd = min(bd, t3_ - t2_)
                    good = bool(r2_ == m and np.array_equal(b_[:m], data))
                finally:
                    for p in free:
                        L.lz4f_mi355x_host_free(p)
                g = m / GIB
                out[mode] = {"compress_GiBs": round(g / bc, 2), "decompress_GiBs": round(g / bd, 2), "round_trip_GiBs": round(g / (bc + bd), 2), "roundtrip_verified": good}
            return out
        if want("host"): side["host_to_host"] = leg(host_legs)

        def conduit_replay():
            import oracle
            m = 32 << 20
            data = src[:m].cpu().numpy().tobytes()
            chunks = [data[i:i + (1 << 20)] for i in range(0, m, 1 << 20)]            # 1 MiB ByteStrings; the conduit slices them to 16 KiB itself
            t0_ = time.perf_counter(); fr = b"".join(conduit.compress(chunks)); t1_ = time.perf_counter()
            fchunks = [fr[i:i + 65536] for i in range(0, len(fr), 65536)]
            t2_ = time.perf_counter(); outb = b"".join(conduit.decompress(fchunks)); t3_ = time.perf_counter()
            good = outb == data
            t4_ = time.perf_counter(); ref = oracle.conduit_compress(data, None, 16384); t5_ = time.perf_counter()
            t6_ = time.perf_counter(); rb, _ = oracle.decompress_frame(ref, cap=m + 64); t7_ = time.perf_counter()
            g = m / GIB
            return {"what": "Baseline B: the reference's call pattern (Conduit.hsc:457-533, :598-701; 16 KiB slices, default preferences = 64 KiB linked blocks) on 32 MiB: "
                            "this library's twelve LZ4F_* functions (every completed block: one DMA up out of page-locked staging, three launches, one DMA back, one synchronisation - ~46 us of kernels for 64 KiB that one CPU core encodes in 60: a synchronous block-at-a-time API is the one shape a GPU cannot win, see DESIGN.md section 7) vs the CPU codec (oracle port, 1 thread) driven the same way",
                    "gpu_library": {"compress_GiBs": round(g / (t1_ - t0_), 3), "decompress_GiBs": round(g / (t3_ - t2_), 3), "roundtrip_verified": good},
                    "cpu_1_thread": {"compress_GiBs": round(g / (t5_ - t4_), 3), "decompress_GiBs": round(g / (t7_ - t6_), 3), "kind": "port", "roundtrip_verified": rb == data}}
        if want("replay"): side["conduit_replay"] = leg(conduit_replay)

    if rank == 0:
        total_u = n * world * args.steps
        value = total_u / dt / GIB
        mean = {k: v for k, v in kt.items() if v and v > 0}
        algo = float(n + frame_only)               # U + C per direction; match copies served on-chip are not counted
        kernels = {k: {"ms": round(ms, 4), "algo_GBs": round(algo / (ms * 1e-3) / 1e9, 1)} for k, ms in mean.items()
                   if k in ("find_matches", "emit", "decode", "decode_parse", "decode_copy", "walk")}
        dom = max((k for k in kernels if k in ("find_matches", "emit", "decode")), key=lambda k: kernels[k]["ms"]) if kernels else None
        t_comp = sum(mean.get(k, 0.0) for k in ("find_matches", "layout", "emit", "xxh32_write"))
        t_dec = sum(mean.get(k, 0.0) for k in ("walk", "xxh32_verify", "decode", "finish"))
        out = {
            "metric": "GiB/s end-to-end (compress+decompress), 4 MiB blocks, 1/2/4/8 GPUs vs liblz4",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synth50 (~50%% compressible), %.0f GiB per GPU per step, %d KiB %s blocks, device-resident (inputs/outputs in HBM), "
                                   "block checksums %s, content checksum off, decode from the byte stream alone (%s)" % (
                                       n / GIB, bs >> 10, "linked" if args.linked else "independent", "on" if args.block_checksum else "off",
                                       "bare LZ4 frame: seeded size-word walk + stretch-parallel self-index + indexed kernels" if args.foreign else
                                       "the compressor's index travels in the stream as a skippable frame behind the LZ4 frame; no block table, no side buffer"),
                       "bytes_per_gpu": n, "block_size": bs, "n_blocks_per_gpu": nb, "generator": "synth50 recipe, torch Philox seed 1234+rank",
                       "sharding": ("BASELINE configs[3] shape: %d ranks x %.0f GiB = %.0f GiB of synth50 per step, every rank its own stream (seed 1234+rank) of %d "
                                    "independent blocks on its own GPU and HIP stream; nothing crosses ranks on the data path (no RCCL collective: blocks of an "
                                    "independent-block frame need nothing from each other, SURVEY 8e); RCCL carries the barrier and the max-over-ranks time only"
                                    % (world, n / GIB, world * n / GIB, nb)) if world > 1 else "single GPU",
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1)},
            "ratio": round(n / frame_only, 4), "compressed_bytes": frame_only, "stream_bytes_with_trailer": csize_stream, "roundtrip_verified": ok,
            "compress_GiBs_per_gpu": round(n / (t_comp * 1e-3) / GIB, 2) if t_comp else None,
            "decompress_GiBs_per_gpu": round(n / (t_dec * 1e-3) / GIB, 2) if t_dec else None,
            "kernels": kernels,
        }
        out.update(side)
        if dom:
            a = kernels[dom]["algo_GBs"]
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                               "traffic": pmc_traffic(dom, n, bs), "traffic_source": PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)",
                               "algorithmic_bytes_per_launch": int(algo), "ms_per_launch": kernels[dom]["ms"],
                               "frac_of_measured_copy_peak": round(a / HBM_COPY_GBS, 4)}
            if "decode" in kernels:
                d = kernels["decode"]["algo_GBs"]
                out["roofline_decode"] = {"bound": "hbm", "achieved": d, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d / HBM_PEAK_GBS, 4),
                                          "traffic": pmc_traffic("decode", n, bs),
                                          "kernels": "k_check_index + k_parse_indexed + k_resolve_direct + k_copy_indexed (+ the trailer's link check in `walk`)" if inband else "k_decode_blocks_fused"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(bs, args.cpu_sample_mib << 20)
            except Exception as e:  # the baseline is a reported reference point, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
