#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: GiB/s end-to-end (compress+decompress), 4 MiB blocks.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  compress    lz4f_mi355x_dev_compressFrameIndexed(..., LZ4F_MI355X_INBAND): find_matches -> layout -> emit, then the trailer
              (the blocks' positions and the sequence index as a skippable frame behind the LZ4 frame, part of the byte stream)
  decompress  lz4f_mi355x_dev_decompressFrame on those bytes and nothing else: no block table, no side buffer.  It finds the
              trailer, checks it against the frame, parses per index entry, resolves direct matches, copies.
both through the C ABI on torch's current stream.  The same LZ4 frame WITHOUT its trailer - what a frame from liblz4 or the
`lz4` tool looks like to this decoder - is timed after the timed region (`foreign_frame`: the stream framed by liblz4.so.1 itself, and beside it this
library's own frame minus its trailer; seeded size-word walk, stretch-parallel self-index, indexed kernels).
Workload: one GPU: BASELINE configs[2] -- 4 GiB of synth50 (~50 % compressible), 4 MiB independent blocks.  N > 1: configs[3]'s shape --
every rank its own 8 GiB stream (seed 1234+rank; 8 ranks = the 64 GiB the config names, and every call crosses 2^32 bytes); frame
blocks are independent, so ranks shard with no data-path collective ("weak" scaling: per-GPU work fixed for N >= 2).
value = bytes of uncompressed input all ranks processed / max-over-ranks wall time of the K steps.
`multi_device_host` is the other multi-GPU number: ONE frame in page-locked host memory whose slabs lz4f_mi355x_use_devices deals over
the N GPUs (rank 0 drives them all after the timed region; one GPU: two logical devices, plumbing only).

Extra objects on the JSON line (all outside the timed region):
  roofline       the kernel with the largest share of the step: algorithmic bytes (U + C per direction) /
                 its launch duration measured with HIP events on the launch stream (events of the last timed step)
  kernels        the same for every kernel of the step
  foreign_frame  decode of the bare LZ4 frame (walk included) and its roofline fraction
  block_checksum_on   the step with XXH32 block checksums written and verified on the GPU
  cfg2           BASELINE configs[1]: decompress-only, 1 GiB of text pre-framed by liblz4 at 64 KiB independent blocks, bare frame, walk included
  host_to_host   lz4f_mi355x_compressFrame / decompressFrame on host buffers (SURVEY 8d variant H), pageable and page-locked
  content_checksum_cap   what a frame's content checksum (one serial XXH32 chain) caps a stream at: device wave, host thread
  multi_device_host      one frame's slabs dealt over N devices, host to host
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
PMC_PROFILE = "profiles/round4_pmc_traffic.json"


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


def liblz4_frame(tile, bsid: int, reps: int):
    """One LZ4 frame of `reps` copies of the numpy byte array `tile`, independent blocks of size id `bsid`, framed by the REFERENCE's codec:
    LZ4F_compressFrame of the tile through the installed liblz4.so.1 (else the oracle port, bit-exact with it), the tile's blocks `reps` times
    between one header and one EndMark (the tile is a whole number of blocks).  -> (numpy frame with 64 spare bytes, frame size, who framed it)"""
    import numpy as np
    from lz4_frame_conduit_amd import conduit
    tile_n = int(tile.size)
    assert tile_n % (1 << (8 + 2 * bsid)) == 0
    try:
        lz = ctypes.CDLL("liblz4.so.1")
        lz.LZ4F_compressFrameBound.restype = ctypes.c_size_t; lz.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        lz.LZ4F_compressFrame.restype = ctypes.c_size_t; lz.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lp = conduit.make_preferences(blockSizeID=bsid, blockMode=1)
        cap = lz.LZ4F_compressFrameBound(tile_n, ctypes.byref(lp))
        buf = np.empty(cap, dtype=np.uint8)
        r_ = lz.LZ4F_compressFrame(buf.ctypes.data_as(ctypes.c_void_p), cap, tile.ctypes.data_as(ctypes.c_void_p), tile_n, ctypes.byref(lp))
        if r_ < 64 or r_ > cap: raise OSError("LZ4F_compressFrame: %d" % r_)
        one = buf[:r_]; framer = "liblz4.so.1 %d (LZ4F_compressFrame)" % lz.LZ4_versionNumber()
    except (OSError, AttributeError):
        import oracle
        one = np.frombuffer(oracle.conduit_compress(tile.tobytes(), oracle.mkprefs(bsid=bsid, indep=1)), dtype=np.uint8)
        framer = "oracle port of liblz4 1.9.3 (bit-exact with it: tests/test_oracle_golden.py)"
    assert bytes(one[:4]) == b"\x04\x22\x4d\x18" and bytes(one[-4:]) == bytes(4) and one.size > 11
    body = one[7:-4]
    host = np.empty(7 + body.size * reps + 4 + 64, dtype=np.uint8)
    host[:7] = one[:7]
    for r_ in range(reps): host[7 + r_ * body.size: 7 + (r_ + 1) * body.size] = body
    fsize = 7 + body.size * reps + 4
    host[fsize - 4:] = 0
    return host, fsize, framer


def where_am_i():
    """CPU affinity and NUMA node of the calling thread (the host legs' numbers move with them)."""
    out = {}
    try:
        aff = sorted(os.sched_getaffinity(0)); out["affinity"] = "%d cpus: %d-%d" % (len(aff), aff[0], aff[-1])
        cpu = None
        with open("/proc/self/stat") as f: cpu = int(f.read().rsplit(")", 1)[1].split()[36])
        out["running_on_cpu"] = cpu
        for nd in sorted(os.listdir("/sys/devices/system/node")):
            if nd.startswith("node") and os.path.exists("/sys/devices/system/node/%s/cpu%d" % (nd, cpu)): out["numa_node"] = int(nd[4:])
        out["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")])
    except Exception as e:      # noqa: BLE001
        out["note"] = repr(e)
    return out


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
    ap.add_argument("--bytes", type=int, default=0, help="uncompressed bytes per GPU per step (default: 4 GiB on one GPU = BASELINE configs[2]; 8 GiB per rank on N > 1, "
                                                        "so that 8 ranks move the 64 GiB of configs[3] - and every call crosses 2^32 bytes)")
    ap.add_argument("--block-size-id", type=int, default=7, help="4=64KiB 5=256KiB 6=1MiB 7=4MiB")
    ap.add_argument("--block-checksum", type=int, default=0)
    ap.add_argument("--linked", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the side legs: what the rocprofv3 summaries under profiles/ are taken with, so that their per-kernel averages are the headline launches' only")
    ap.add_argument("--foreign", action="store_true", help="the timed decode gets the bare LZ4 frame (no trailer): walk + self-index + indexed kernels")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--legs", default="all", help="comma-separated side legs to run (foreign,bck,cfg2,dense,linked,host,cck,replay,multi); what tools/prof_leg.sh profiles one at a time")
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

    n = args.bytes if args.bytes else ((4 << 30) if world == 1 else (8 << 30))
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
            def timed(fr, size, expect):
                t = []; good = True
                for _ in range(3):
                    spoil(back)
                    eng.decompress_frame_async(fr, size, back)
                    rf = eng.result(); tt = eng.get_timing()
                    t.append((tt["decompress_total"], tt["walk"], tt["decode"]))
                    good = good and bool(rf.size == n and rf.consumed == size and torch.equal(back, expect))
                best = min(t)
                a_ = float(n + size)
                return {"ms": round(best[0], 4), "walk_ms": round(best[1], 4), "decode_ms": round(best[2], 4), "frame_bytes": int(size), "roundtrip_verified": good,
                        "roofline": {"bound": "hbm", "achieved": round(a_ / (best[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(a_ / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            # (a) bytes the REFERENCE's codec wrote: LZ4F_compressFrame (liblz4.so.1) of the stream's first 512 MiB, its blocks eight times
            tile_n = min(n, 512 << 20); reps = n // tile_n
            host, fsize, framer = liblz4_frame(src[:tile_n].cpu().numpy(), args.block_size_id, reps)
            f_l = torch.from_numpy(host).to(dev)
            exp = src[:tile_n].repeat(reps) if reps > 1 else src
            out_ = timed(f_l, fsize, exp)
            del f_l, exp
            # (b) this library's own frame without its trailer (what rounds 2-3 reported here)
            own = timed(frame, frame_only, src)
            out_.update({"what": "a bare LZ4 frame (no trailer, no block table), device-resident: seeded size-word walk, then the decoder cuts the blocks into stretches itself "
                                 "(decode_spx.cuh) and runs the indexed kernels", "framer": framer + ", %d MiB tile x %d" % (tile_n >> 20, reps),
                         "own_frame_minus_trailer": own})
            return out_
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
            reps = m // tile_n
            host, fsize, framer = liblz4_frame(tile, 4, reps)
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

        def dense_big():
            # what `lz4 -c` writes of text by default (test/Main.hs:33-36): 4 MiB independent blocks, no index - 256 blocks per GiB, each one chain of ~13-byte sequences
            m, tile_n = 1 << 30, 128 << 20
            tile = datagen.synth_text(tile_n, 99)
            reps = m // tile_n
            host, fsize, framer = liblz4_frame(tile, 7, reps)
            f2 = torch.from_numpy(host).to(dev)
            tx = torch.from_numpy(tile).to(dev).repeat(reps)
            b2 = torch.empty(m, dtype=torch.uint8, device=dev)
            best = None
            good = True
            for _ in range(3):
                spoil(b2)
                eng.decompress_frame_async(f2, fsize, b2); rd = eng.result(); td = eng.get_timing()
                good = good and bool(rd.size == m and rd.consumed == fsize and torch.equal(b2, tx))
                if best is None or td["decompress_total"] < best: best = td["decompress_total"]
            return {"workload": "decompress-only, 1 GiB of synthetic text pre-framed at 4 MiB independent blocks by %s, bare LZ4 frame, device-resident: a workgroup per block "
                                "(decode_relay.cuh)" % framer, "decompress_ms": round(best, 3), "decompress_GiBs": round(m / (best * 1e-3) / GIB, 1), "roundtrip_verified": good}
        if want("dense"): side["text_4mib_independent_blocks"] = leg(dense_big)

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
            out = {"what": "lz4f_mi355x_compressFrame / decompressFrame, 1 GiB of the stream, 4 MiB independent blocks, host buffers in and out, best of 5 after a warm-up pass; the link: PCIe Gen5 x16, "
                           "%.0f GB/s spec, 53-54 GiB/s measured each way - a round trip moves the uncompressed bytes over it twice, so <= ~26 GiB/s per GPU whatever the kernels do" % PCIE_GBS,
                   "calling_thread": where_am_i()}

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
                    for it in range(6):                                  # (first pass: warm-up; best of 5)
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

        def cck_cap():
            # what a frame's content checksum (one XXH32 over the whole stream: one dependent chain) caps a stream at
            L = _ffi.lib()
            m = 256 << 20
            p0 = conduit.make_preferences(blockSizeID=7, blockMode=1); p1 = conduit.make_preferences(blockSizeID=7, blockMode=1, contentChecksum=1)
            f_ = torch.empty(eng.frame_bound(m, p1), dtype=torch.uint8, device=dev)
            def dev_ms(pp):
                best = 1e9
                for _ in range(2):
                    eng.compress_async(src[:m], f_, pp); eng.result(); best = min(best, eng.get_timing()["compress_total"])
                return best
            d0, d1 = dev_ms(p0), dev_ms(p1)
            data = src[:m].cpu().numpy(); bound = L.lz4f_mi355x_compressFrameBound(m, ctypes.byref(p1)); dst_ = np.empty(bound, dtype=np.uint8)
            def host_s(pp):
                best = 1e9
                for _ in range(3):
                    t0_ = time.perf_counter(); rr = L.lz4f_mi355x_compressFrame(dst_.ctypes.data_as(ctypes.c_void_p), bound, data.ctypes.data_as(ctypes.c_void_p), m, ctypes.byref(pp)); t1_ = time.perf_counter()
                    assert not L.LZ4F_isError(rr)
                    best = min(best, t1_ - t0_)
                return best
            h0, h1 = host_s(p0), host_s(p1)
            return {"what": "contentChecksum = 1 on 256 MiB of the stream: XXH32 of a whole stream is ONE chain of four accumulators (no way to combine partial states), so it caps the stream "
                            "whatever the block kernels do: the device path runs it on one wave (k_xxh32_content), the host-pointer path on a host thread beside the transfers",
                    "device_wave_GBs": round(m / max((d1 - d0) * 1e-3, 1e-9) / 1e9, 2), "device_compress_ms_without": round(d0, 3), "device_compress_ms_with": round(d1, 3),
                    "host_call_GiBs_without": round(m / GIB / h0, 2), "host_call_GiBs_with": round(m / GIB / h1, 2)}
        if want("cck"): side["content_checksum_cap"] = leg(cck_cap)

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

    def multi_device_host(n_dev: int, m: int, logical: bool):
        """ONE frame dealt over n_dev GPUs: lz4f_mi355x_use_devices(n_dev) + compressFrame / decompressFrame on page-locked host memory - the path a
        Conduit user's stream takes, and the one that shards a single frame's blocks over the node (pipeline.hip: slabs dealt round-robin, a
        stream per device, only sizes meet on the host)."""
        L = _ffi.lib()
        if logical: os.environ["LZ4F_MI355X_LOGICAL_DEVICES"] = str(n_dev)
        hp = conduit.make_preferences(blockSizeID=7, blockMode=1)
        bound = L.lz4f_mi355x_compressFrameBound(m, ctypes.byref(hp))
        free = []
        try:
            rdev = L.lz4f_mi355x_use_devices(n_dev)
            if L.LZ4F_isError(rdev): raise RuntimeError("use_devices(%d): %s" % (n_dev, L.LZ4F_getErrorName(rdev).decode()))
            for x in (m, bound, m + 8):
                p_ = L.lz4f_mi355x_host_alloc(x)
                if not p_: raise MemoryError("page-locked allocation of %d bytes" % x)
                free.append(p_)
            s_, d_, b_ = [np.ctypeslib.as_array((ctypes.c_uint8 * x).from_address(p_)) for x, p_ in zip((m, bound, m + 8), free)]
            tile = src[:min(m, 1 << 30)].cpu().numpy()
            for a in range(0, m, tile.size): s_[a:a + tile.size] = tile[:min(tile.size, m - a)]
            bc = bd = 1e9; good = True
            for it in range(4):
                t0_ = time.perf_counter()
                rr = L.lz4f_mi355x_compressFrame(d_.ctypes.data_as(ctypes.c_void_p), bound, s_.ctypes.data_as(ctypes.c_void_p), m, ctypes.byref(hp))
                t1_ = time.perf_counter()
                if L.LZ4F_isError(rr): raise RuntimeError(L.LZ4F_getErrorName(rr).decode())
                used = ctypes.c_size_t(0)
                t2_ = time.perf_counter()
                r2_ = L.lz4f_mi355x_decompressFrame(b_.ctypes.data_as(ctypes.c_void_p), m + 8, d_.ctypes.data_as(ctypes.c_void_p), rr, ctypes.byref(used))
                t3_ = time.perf_counter()
                if L.LZ4F_isError(r2_): raise RuntimeError(L.LZ4F_getErrorName(r2_).decode())
                if it: bc = min(bc, t1_ - t0_); bd = min(bd, t3_ - t2_)
            good = bool(r2_ == m and used.value == rr and np.array_equal(b_[:m], s_))
            g = m / GIB
            return {"what": "ONE LZ4 frame of %.0f GiB (4 MiB independent blocks) in page-locked host memory, its slabs dealt over %d %s device(s) by lz4f_mi355x_use_devices, "
                            "compressFrame + decompressFrame host to host, best of 3 after a warm-up pass" % (g, n_dev, "LOGICAL (mapped onto the visible GPU: plumbing, not a measurement of N links)" if logical else "physical"),
                    "devices": n_dev, "logical": logical, "compress_GiBs": round(g / bc, 2), "decompress_GiBs": round(g / bd, 2), "round_trip_GiBs": round(g / (bc + bd), 2),
                    "roundtrip_verified": good, "calling_thread": where_am_i()}
        finally:
            for p_ in free: L.lz4f_mi355x_host_free(p_)
            L.lz4f_mi355x_use_devices(1)
            L.lz4f_mi355x_release_engines()
            if logical: os.environ.pop("LZ4F_MI355X_LOGICAL_DEVICES", None)
    if rank == 0 and not args.headline_only and want("multi"):
        # N ranks: the one-frame host path over all N GPUs of the node (the other ranks idle at the closing barrier); one GPU: two logical devices
        if world > 1 and not share: side["multi_device_host"] = leg(lambda: multi_device_host(world, 8 << 30, False))
        elif world > 1: side["multi_device_host"] = leg(lambda: multi_device_host(world, 512 << 20, True))      # (LZ4F_BENCH_SHARE_GPU: a test switch)
        else: side["multi_device_host"] = leg(lambda: multi_device_host(2, 2 << 30, True))

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
                                          "kernels": "k_check_index + k_copy_selffed (the copy workgroup's first wave parses and resolves; + the trailer's link check in `walk`)" if inband else "k_decode_blocks_fused"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(bs, args.cpu_sample_mib << 20)
            except Exception as e:  # the baseline is a reported reference point, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "GiB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    eng.close()
    if world > 1:
        # (rank 0's one-frame leg drives every GPU of the node from its own process: the other ranks wait for it on the CPU - the
        # rendezvous store - not in a collective that would spin on their GPUs meanwhile)
        import datetime
        try:
            store = dist.distributed_c10d._get_default_store()
            if rank == 0: store.set("lz4f_bench_rank0_done", "1")
            else: store.wait(["lz4f_bench_rank0_done"], datetime.timedelta(minutes=15))
        except Exception:      # noqa: BLE001
            pass
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
