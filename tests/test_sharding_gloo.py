"""CPU, 2 processes over gloo: the multi-GPU path of bench.py is "one rank per GPU, clips sharded, no data-path
collective".  What can be wrong without a GPU is the sharding itself (overlapping or missing clips), the
barrier / MAX-over-ranks timing reduction, and rank-0-only reporting -- tested here with world_size 2.
Each rank also decodes its own shard (with the CPU command-list interpreter, a test tool) and the union
is compared with a single-process decode of all clips."""
import hashlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mobiclipdecoder_amd import default_params, generate_clip, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _decode_hash(seed):
    from tests.interp_binding import InterpDecoder
    p = default_params("A", seed, n_frames=4, width=64, height=48)
    data, fo = generate_clip(p)
    d = InterpDecoder(64, 48, 1)
    h = hashlib.sha256()
    for f in range(4):
        d.Data, d.Offset = data[: fo[f + 1]], int(fo[f])
        y, uv = d.DecodeFrame()
        h.update(y.tobytes())
        h.update(uv.tobytes())
    return h.hexdigest()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # weak scaling (bench.py): each rank generates its own distinct streams from disjoint seeds
    seeds = [sharding.stream_seed("A", rank, i) for i in range(3)]
    hashes = {s: _decode_hash(s) for s in seeds}
    # strong-scaling helper: round-robin ownership of a global clip list
    mine = sharding.clips_of_rank(11, rank, world)
    dist.barrier()
    elapsed = sharding.max_over_ranks(dist, 1.0 + rank)  # rank 1 is "slower"
    gathered = [None] * world
    dist.all_gather_object(gathered, {"seeds": seeds, "hashes": hashes, "mine": mine, "elapsed": elapsed})
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_without_overlap_and_reduce_max():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_seeds = [s for g in got for s in g["seeds"]]
    assert len(set(all_seeds)) == len(all_seeds) == 6                      # disjoint streams per rank
    assert sorted(c for g in got for c in g["mine"]) == list(range(11))    # every clip owned exactly once
    assert all(g["elapsed"] == 2.0 for g in got)                           # MAX over ranks, same on every rank
    # union of the shards == single-process decode
    for g in got:
        for s, h in g["hashes"].items():
            assert _decode_hash(s) == h
    # job aggregate: weak scaling doubles the pixels for the same (max) time
    assert sharding.whole_job_mpix_per_s(2, 8, 4, 640, 480, 1.0) == 2 * sharding.whole_job_mpix_per_s(1, 8, 4, 640, 480, 1.0)


def test_single_rank_is_identity():
    assert sharding.max_over_ranks(None, 3.5) == 3.5
    assert sharding.clips_of_rank(5, 0, 1) == [0, 1, 2, 3, 4]
