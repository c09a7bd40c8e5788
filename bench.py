#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: GiB/s end-to-end (compress+decompress), 4 MiB blocks.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: compress the stream into one LZ4 frame (find_matches -> layout -> emit kernels), then decompress
that frame (block table and sequence index from the compressor: parse per index entry -> resolve direct matches ->
copier workgroups), all through the C ABI (lz4f_mi355x_dev_compressFrameIndexed / lz4f_mi355x_dev_decompressBlocksIndexed)
on torch's current stream.  `--no-index` runs the path a foreign frame takes (generic fused decoder); the default run
reports that decoder's time too (`decode_without_index_ms`, measured after the timed region).
Workload at every N: BASELINE configs[2] per GPU -- 4 GiB of synth50 (~50 % compressible), 4 MiB
independent blocks; frame blocks are independent, so ranks shard the stream with no data-path
collective ("weak" scaling: every rank gets its own 4 GiB with seed 1234+rank).
value = bytes of uncompressed input all ranks processed / max-over-ranks wall time of the K steps.

Extra objects on the JSON line:
  roofline     : the kernel with the largest share of the step, algorithmic bytes (U + C per direction,
                 SURVEY.md section 8d) / its launch duration measured with HIP events on the launch stream (the K steps of
                 the timed region are enqueued back to back; the events read afterwards are those of the last of them)
  kernels      : the same for every kernel of the step
  cpu_baseline : the same blocks through liblz4 (dlopen, kind "reference") or the oracle port, on the host cores,
                 rank 0 at N=1 only, on a bounded sample
"""
import argparse
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


def pmc_traffic(kernel: str, n_bytes: int, block_size: int, indexed: bool = True):
    """HBM bytes per launch of `kernel` from the committed PMC profile (rocprofv3 cannot run inside the timed process:
    counters are collected in their own passes, see profiles/round1b_pmc_traffic.json), or None when that profile was
    not taken on this workload."""
    try:
        if n_bytes != (4 << 30) or block_size != (4 << 20):
            return None
        if kernel == "decode" and not indexed:
            with open(os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")) as f:
                return int(json.load(f)["kernels"]["k_decode_blocks_fused<lz4f::FzCfg<8> >"]["hbm_bytes_corrected"])
        with open(os.path.join(ROOT, "profiles", "round1b_pmc_traffic.json")) as f:
            prof = json.load(f)
        names = {"find_matches": ["k_find_matches<1>"], "emit": ["k_emit_gather<4>"],
                 "decode": ["k_parse_indexed", "k_resolve_direct", "k_copy_indexed<FzCfg<8> >"]}
        return int(sum(prof["kernels"][k]["hbm_bytes_corrected"] for k in names[kernel]))
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
    ap.add_argument("--headline-only", action="store_true", help="skip the side legs (other framings, dense data): what the rocprofv3 summaries under profiles/ are taken with, so that their per-kernel averages are the headline launches' only")
    ap.add_argument("--no-index", action="store_true", help="decode without the compressor's sequence index (what a foreign frame gets)")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from lz4_frame_conduit_amd import conduit, datagen, shard
    from lz4_frame_conduit_amd.device import Engine, synth50_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    n = args.bytes
    bs = 1 << (8 + 2 * args.block_size_id)
    prefs = conduit.make_preferences(blockSizeID=args.block_size_id, blockMode=0 if args.linked else 1, blockChecksum=args.block_checksum)
    nb = (n + bs - 1) // bs

    src = synth50_device(n, 1234 + rank, dev)
    eng = Engine(local_rank)
    frame = torch.empty(eng.frame_bound(n, prefs), dtype=torch.uint8, device=dev)
    back = torch.empty_like(src)
    table = eng.new_table(nb)
    # (the indexed kernels take independent blocks of 256 KiB and more, and linked frames of any block size)
    index = None if (args.no_index or (not args.linked and bs < (256 << 10))) else eng.new_index(n, prefs)
    eng.set_timing(True)

    def step():
        eng.compress_async(src, frame, prefs, table, index)
        # frame size is known on the device only; the decoder needs just an upper bound for bounds checks
        eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, prefs.frameInfo, index)

    def barrier():
        shard.barrier_all(torch.device(dev))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.compress_async(src, frame, prefs, table, index)
    r = eng.result()
    csize = int(r.size)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                                                # enqueue only: nothing in the timed region waits for the host
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dev)
    kt = {}
    if rank == 0:                                             # per-kernel HIP-event times of the last timed step (the events are recorded
        for k, v in eng.get_timing().items():                 # on the launch stream in every step; reading them is what would synchronise)
            kt.setdefault(k, []).append(v)
    r2 = eng.result()
    ok = bool(r2.size == n and torch.equal(back, src))
    generic_ms = None
    if index is not None and rank == 0 and not args.headline_only:      # the same frame through the generic decoder (no index), outside the timed region
        t = []
        for _ in range(3):
            back.zero_()
            eng.decompress_blocks_async(frame, frame.numel(), back, table, nb, prefs.frameInfo)
            r3 = eng.result()
            t.append(eng.get_timing()["decode"] + eng.get_timing()["finish"])
            ok = ok and bool(r3.size == n and torch.equal(back, src))
        generic_ms = round(min(t), 4)

    linked_leg = dense_leg = None
    if rank == 0 and world == 1 and not args.linked and n >= (1 << 30) and not args.headline_only:
        # the reference's DEFAULT framing (64 KiB linked blocks), outside the timed region: 1 GiB of the same stream, with the index
        try:
            m = 1 << 30
            lp = conduit.make_preferences(blockSizeID=4, blockMode=0)
            lnb = m >> 16
            lframe = torch.empty(eng.frame_bound(m, lp), dtype=torch.uint8, device=dev)
            ltable, lindex = eng.new_table(lnb), eng.new_index(m, lp)
            lback = torch.empty(m, dtype=torch.uint8, device=dev)
            best = None
            for _ in range(3):
                eng.compress_async(src[:m], lframe, lp, ltable, lindex)
                eng.decompress_blocks_async(lframe, lframe.numel(), lback, ltable, lnb, lp.frameInfo, lindex)
                rl = eng.result()
                t = eng.get_timing()
                tc = t["find_matches"] + t["layout"] + t["emit"]
                td = t["decode"] + t["finish"]
                if best is None or tc + td < best[0] + best[1]:
                    best = (tc, td)
            lok = bool(rl.size == m and torch.equal(lback, src[:m]))
            foreign = None
            for _ in range(2):                                    # the same frame as a foreign one would arrive: no index (the decoder makes its own)
                lback.zero_()
                eng.decompress_blocks_async(lframe, lframe.numel(), lback, ltable, lnb, lp.frameInfo)
                rf = eng.result()
                t = eng.get_timing()
                foreign = t["decode"] + t["finish"] if foreign is None else min(foreign, t["decode"] + t["finish"])
                lok = lok and bool(rf.size == m and torch.equal(lback, src[:m]))
            linked_leg = {"workload": "1 GiB of the same stream, 64 KiB LINKED blocks (Conduit.hsc default preferences)",
                          "compress_ms": round(best[0], 3), "decompress_ms_with_index": round(best[1], 3),
                          "decompress_ms_without_index": round(foreign, 3),
                          "e2e_GiBs_with_index": round(m / ((best[0] + best[1]) * 1e-3) / GIB, 1),
                          "e2e_GiBs_without_index": round(m / ((best[0] + foreign) * 1e-3) / GIB, 1), "roundtrip_verified": lok}
            ok = ok and lok
            try:
                # the same framing on DENSE data (text: ~9 output bytes per sequence), where a block - here the whole frame - is one
                # match chain: decoded by pointer doubling over the output bytes
                dm = 256 << 20
                tx = torch.from_numpy(datagen.synth_text(8 << 20, 99)).to(dev).repeat(dm // (8 << 20))
                dnb = dm >> 16
                # (the recommended index size is for one sequence per 64 input bytes; text has one per 9: eight times that)
                dtable, dindex = eng.new_table(dnb), torch.zeros(eng.index_size(dm, lp) * 8, dtype=torch.uint8, device=dev)
                eng.compress_async(tx, lframe, lp, dtable, dindex)
                dsize = int(eng.result().size)
                dbest = None
                for _ in range(3):
                    eng.compress_async(tx, lframe, lp, dtable, dindex)
                    eng.decompress_blocks_async(lframe, lframe.numel(), lback[:dm], dtable, dnb, lp.frameInfo, dindex)
                    rd = eng.result()
                    t = eng.get_timing()
                    tc, td = t["find_matches"] + t["layout"] + t["emit"], t["decode"] + t["finish"]
                    if dbest is None or td < dbest[1]:
                        dbest = (tc, td)
                dok = bool(rd.size == dm and torch.equal(lback[:dm], tx))
                dforeign = None
                for _ in range(2):                                # as a foreign frame: no index, the decoder makes its own
                    lback[:dm].zero_()
                    eng.decompress_blocks_async(lframe, lframe.numel(), lback[:dm], dtable, dnb, lp.frameInfo)
                    rf = eng.result()
                    t = eng.get_timing()
                    dforeign = t["decode"] + t["finish"] if dforeign is None else min(dforeign, t["decode"] + t["finish"])
                    dok = dok and bool(rf.size == dm and torch.equal(lback[:dm], tx))
                dense_leg = {"workload": "256 MiB of synthetic text (Zipf words, ratio %.2f), 64 KiB LINKED blocks" % (dm / max(1, dsize)),
                             "compress_ms": round(dbest[0], 3), "decompress_ms_with_index": round(dbest[1], 3),
                             "decompress_ms_without_index": round(dforeign, 3),
                             "decompress_GiBs_with_index": round(dm / (dbest[1] * 1e-3) / GIB, 2), "decompress_GiBs_without_index": round(dm / (dforeign * 1e-3) / GIB, 2),
                             "roundtrip_verified": dok}
                ok = ok and dok
                del tx, dtable, dindex
            except Exception as e:      # noqa: BLE001 - a bench leg must not take the headline down
                dense_leg = {"error": repr(e)}
            del lframe, lback, ltable, lindex
        except Exception as e:
            linked_leg = {"error": repr(e)}
    if rank == 0:
        total_u = n * world * args.steps
        value = total_u / dt / GIB
        mean = {k: (sum(v) / len(v)) for k, v in kt.items() if v and sum(v) > 0}
        algo = float(n + csize)               # U + C per direction; match copies served on-chip are not counted
        kernels = {k: {"ms": round(ms, 4), "algo_GBs": round(algo / (ms * 1e-3) / 1e9, 1)} for k, ms in mean.items()
                   if k in ("find_matches", "emit", "decode", "decode_parse", "decode_copy")}
        dom = max((k for k in kernels if not k.startswith("decode_")), key=lambda k: kernels[k]["ms"]) if kernels else None
        t_comp = sum(mean.get(k, 0.0) for k in ("find_matches", "layout", "emit", "xxh32_write"))
        t_dec = sum(mean.get(k, 0.0) for k in ("walk", "xxh32_verify", "decode", "finish"))
        out = {
            "metric": "GiB/s end-to-end (compress+decompress), 4 MiB blocks, 1/2/4/8 GPUs vs liblz4",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synth50 (~50%% compressible), %.0f GiB per GPU per step, %d KiB %s blocks, device-resident (inputs/outputs in HBM), "
                                   "block checksums %s, content checksum off, decode %s" % (n / GIB, bs >> 10, "linked" if args.linked else "independent",
                                                                                "on" if args.block_checksum else "off",
                                                                                "with the compressor's sequence index" if index is not None else "without index"),
                       "bytes_per_gpu": n, "block_size": bs, "n_blocks_per_gpu": nb, "generator": "synth50 recipe, torch Philox seed 1234+rank",
                       "sharding": "one 4 GiB stream per rank, no collective" if world > 1 else "single GPU"},
            "ratio": round(n / csize, 4), "compressed_bytes": csize, "roundtrip_verified": ok,
            "compress_GiBs_per_gpu": round(n / (t_comp * 1e-3) / GIB, 2) if t_comp else None,
            "decompress_GiBs_per_gpu": round(n / (t_dec * 1e-3) / GIB, 2) if t_dec else None,
            "kernels": kernels,
            "decode_without_index_ms": generic_ms,
            "reference_default_framing": linked_leg,
            "dense_default_framing": dense_leg,
        }
        if dom:
            a = kernels[dom]["algo_GBs"]
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                               "traffic": pmc_traffic(dom, n, bs, index is not None), "traffic_source": "profiles/round1b_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)",
                               "algorithmic_bytes_per_launch": int(algo), "ms_per_launch": kernels[dom]["ms"],
                               "frac_of_measured_copy_peak": round(a / HBM_COPY_GBS, 4)}
            if "decode" in kernels:
                d = kernels["decode"]["algo_GBs"]
                out["roofline_decode"] = {"bound": "hbm", "achieved": d, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d / HBM_PEAK_GBS, 4),
                                          "traffic": pmc_traffic("decode", n, bs, index is not None),
                                          "kernels": "k_parse_indexed + k_resolve_direct + k_copy_indexed + k_finish_decode" if index is not None else "k_decode_blocks_fused + k_finish_decode"}
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
