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
