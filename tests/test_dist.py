"""World-size-2 test of the N>1 path on CPU (gloo): db broadcast, contiguous read shards that keep mates
together, ragged gather.  Each rank classifies its shard with the checker (there is no GPU here); the
gathered result must equal the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, paired, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth
    from bonsai_amd import shard
    w = synth.make_world(O, seed=11, k=31, genome_len=3000)
    nb = w.n_buckets
    # rank 0 owns the db; the others receive it
    if rank == 0:
        flags, keys, vals = (torch.from_numpy(w.flags.view(np.int32).copy()), torch.from_numpy(w.keys.view(np.int64).copy()),
                             torch.from_numpy(w.vals.view(np.int32).copy()))
    else:
        flags, keys, vals = (torch.zeros(max(1, nb >> 4), dtype=torch.int32), torch.zeros(nb, dtype=torch.int64),
                             torch.zeros(nb, dtype=torch.int32))
    shard.broadcast_table(dist, flags, keys, vals, src=0)
    hdr = w.table.header()
    table = O.Table.wrap(hdr[0], hdr[1], hdr[2], hdr[3], flags.numpy().view(np.uint32), keys.numpy().view(np.uint64),
                         vals.numpy().view(np.uint32))
    reads = synth.simulate_reads(np.random.default_rng(99), w.genomes, 1001 if not paired else 1002, var_len=True)
    inc = 2 if paired else 1
    n_units = len(reads) // inc
    lo, hi = shard.shard_range(n_units, rank, world)
    bases, offsets = synth.concat(reads[lo * inc:hi * inc])
    res = O.classify_batch(table, w.tax, 31, bases, offsets, paired=paired)
    local = torch.from_numpy(res["taxon"].astype(np.int64))
    sizes = shard.shard_sizes(n_units, world)
    assert sizes[rank] == hi - lo and sum(sizes) == n_units
    got = shard.gather_results(dist, local, sizes, dst=0)
    if rank == 0:
        fb, fo = synth.concat(reads[:n_units * inc])
        full = O.classify_batch(w.table, w.tax, 31, fb, fo, paired=paired)
        q.put(bool(np.array_equal(got.numpy().astype(np.uint32), full["taxon"])))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("paired", [False, True])
def test_two_rank_shard_broadcast_gather(paired, oracle):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (1 if paired else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, paired, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_range_properties():
    from bonsai_amd import shard
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
