"""N>1 path on CPU: world_size 2, gloo.  Blocks of one stream are dealt round-robin to the ranks
(SURVEY.md 8e); each rank encodes only its own blocks (the oracle stands in for the GPU codec, which is
what tests may use it for), ranks exchange the block size words, rank 0 emits the frame in block order,
and the result must decode to the input.  Also covers the barrier + max-over-ranks timing helper of bench.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank: int, world: int, port: int, outq):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from lz4_frame_conduit_amd import datagen, shard
        data = datagen.synth50(1 << 20, 1234).tobytes() + datagen.ints_100000()[:70000]      # 17 blocks of 64 KiB, last one short
        bs = 65536
        n_blocks = (len(data) + bs - 1) // bs
        mine = list(shard.my_blocks(n_blocks, rank, world))
        assert all(shard.block_owner(i, world) == rank for i in mine)
        local_words, local_payloads = {}, {}
        for i in mine:
            blk = data[i * bs:(i + 1) * bs]
            c = oracle.compress_block(blk, dst_cap=len(blk) - 1)
            if c:
                local_words[i], local_payloads[i] = len(c), c
            else:                                                       # does not shrink: stored block
                local_words[i], local_payloads[i] = len(blk) | 0x80000000, blk
        words = shard.exchange_block_words(local_words, n_blocks)       # the only cross-rank data of the compress path
        assert all(w != 0 for w in words)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(local_payloads, gathered, dst=0)
        shard.barrier_all()
        t = shard.max_over_ranks(1.0 + rank)
        assert t == float(world)
        assert shard.sum_over_ranks(1.0) == float(world)
        if rank == 0:
            payloads = {}
            for d in gathered:
                payloads.update(d)
            header = oracle.header_bytes(oracle.mkprefs(bsid=4, indep=1))
            frame = shard.assemble_frame(header, words, [payloads[i] for i in range(n_blocks)])
            offs = shard.frame_offsets(len(header), words, False)
            assert offs[-1] + 4 == len(frame)
            out, used = oracle.decompress_frame(frame, cap=len(data) + 64)
            assert used == len(frame) and out == data
            # same bytes as the single-process frame (independent blocks: sharding is invisible in the output)
            assert frame == oracle.conduit_compress(data, oracle.mkprefs(bsid=4, indep=1))
        outq.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        outq.put((rank, "FAIL %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_round_robin_blocks_world2_gloo():
    import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _hip_worker(rank: int, world: int, port: int, outq):
    """One process per GPU: rank r encodes its contiguous run of blocks on device r through the C ABI (the HIP path, no oracle in
    the product), the ranks exchange nothing but the bodies' sizes and bytes, rank 0 puts header | body 0 | body 1 | EndMark
    together.  liblz4 (the oracle, as the checker) and the library itself must decode that to the input."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        import oracle
        from lz4_frame_conduit_amd import conduit, datagen, shard
        from lz4_frame_conduit_amd.device import Engine
        bs = 4 << 20
        data = datagen.synth50(64 << 20, 99)
        n_blocks = len(data) // bs
        lo, hi = shard.contiguous_blocks(n_blocks, rank, world)
        dev_i = rank % torch.cuda.device_count()                         # (a one-GPU box: both ranks on device 0, still one process per rank)
        torch.cuda.set_device(dev_i)
        eng = Engine(dev_i)
        prefs = conduit.make_preferences(blockSizeID=7, blockMode=1)
        src = torch.from_numpy(data[lo * bs:hi * bs].copy()).cuda(dev_i)
        frame = torch.empty(eng.frame_bound(src.numel(), prefs), dtype=torch.uint8, device="cuda:%d" % dev_i)
        eng.compress_async(src, frame, prefs)
        r = eng.result()
        body = frame[7:r.size - 4].cpu().numpy().tobytes()              # without the 7-byte header and the EndMark
        sizes = shard.all_gather_sizes(len(body))                        # the only numbers that cross ranks
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(body, gathered, dst=0)
        shard.barrier_all()
        if rank == 0:
            assert sizes == [len(b) for b in gathered]
            whole = oracle.header_bytes(oracle.mkprefs(bsid=7, indep=1)) + b"".join(gathered) + bytes(4)
            out, used = oracle.decompress_frame(whole, cap=len(data) + 64)
            assert used == len(whole) and out == data.tobytes()
            dev = torch.from_numpy(np.frombuffer(whole, dtype=np.uint8).copy()).cuda(dev_i)
            back = torch.zeros(len(data), dtype=torch.uint8, device="cuda:%d" % dev_i)
            eng.decompress_frame_async(dev, dev.numel(), back)
            r2 = eng.result()
            assert r2.size == len(data) and back.cpu().numpy().tobytes() == data.tobytes()
        eng.close()
        outq.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        outq.put((rank, "FAIL %r %s" % (e, traceback.format_exc()[-600:])))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_contiguous_runs_world2_hip_path():
    """The same sharding with the HIP codec, one process per rank: rank r on GPU r where two are visible, both ranks on the one GPU
    of a one-GPU box (two processes, two HIP contexts, gloo for the sizes)."""
    assert torch.cuda.device_count() >= 1
    import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_gpus_flag_is_honoured():
    """`bench.py --gpus N` must never be a silent 1-GPU run: without a launcher it starts N ranks itself (one child process per GPU)
    and refuses when fewer than N GPUs are visible; under a launcher WORLD_SIZE has to agree with --gpus."""
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = torch.cuda.device_count() + 1
    if n < 2: n = 2
    r = subprocess.run([sys.executable, bench, "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr, (r.returncode, r.stderr[-400:])
    assert "\"metric\"" not in r.stdout
    env["WORLD_SIZE"] = "3"
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr, (r.returncode, r.stderr[-400:])


@pytest.mark.gpu
def test_bench_n_rank_path_on_a_shared_gpu():
    """`bench.py --gpus 2` end to end on the one-GPU box: the parent starts two ranks (torch.distributed.run, 127.0.0.1), each builds its own
    stream (seed 1234 + rank), they meet at the barriers, rank 0 takes the maximum of the times and prints ONE JSON line with n_gpus = 2 and the
    aggregate over both ranks.  LZ4F_BENCH_SHARE_GPU=1 is what makes that possible here: both ranks on the visible GPU, gloo for the rendezvous
    (RCCL refuses two ranks on one device); on an N-GPU node the same code runs one rank per GPU over RCCL."""
    import json, subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LZ4F_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bytes", str(256 << 20), "--headline-only", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-1200:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["roundtrip_verified"] is True and d["scaling"] == "weak"
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["bytes_per_gpu"] == 256 << 20 and d["value"] > 0
    assert abs(d["value"] - 2 * (256 << 20) * 2 / (d["ms_per_step"] * 2 * 1e-3) / 2**30) < 0.02 * d["value"]      # the aggregate of both ranks over the slowest rank's time
    # the same launch with rank 0's one-frame leg behind the timed region: a single frame in page-locked host memory, its slabs dealt over
    # the N devices by lz4f_mi355x_use_devices (two LOGICAL devices here), while the other rank waits on the rendezvous store
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--bytes", str(256 << 20), "--legs", "multi", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-1200:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    m = d["multi_device_host"]
    assert "error" not in m, m
    assert d["n_gpus"] == 2 and m["devices"] == 2 and m["logical"] is True and m["roundtrip_verified"] is True and m["round_trip_GiBs"] > 0
