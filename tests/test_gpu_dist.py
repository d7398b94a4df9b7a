"""GPU tier of the N>1 path on a ONE-GPU box: two ranks (gloo rendezvous on 127.0.0.1, both on device 0 -- RCCL refuses two
ranks on one device, so the data-path collectives go through gloo here) each run the HIP classify on their read shard after
receiving the db by broadcast; the gathered taxids must equal the frozen reference-code expectations
(tests/golden/classify_ref.npz).  And `python bench.py --gpus 2` must launch its own two ranks and say n_gpus = 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, paired, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bonsai_amd
    from bonsai_amd import shard
    CL = np.load(os.path.join(HERE, "golden", "classify_ref.npz"))
    nb = int(CL["db_hdr"][0])
    if rank == 0:        # rank 0 owns the db (the reference-built khash arrays); the others receive it
        flags, keys, vals = (torch.from_numpy(CL["db_flags"].view(np.int32).copy()), torch.from_numpy(CL["db_keys_arr"].view(np.int64).copy()),
                             torch.from_numpy(CL["db_vals_arr"].view(np.int32).copy()))
    else:
        flags, keys, vals = (torch.zeros(max(1, nb >> 4), dtype=torch.int32), torch.zeros(nb, dtype=torch.int64),
                             torch.zeros(nb, dtype=torch.int32))
    shard.broadcast_table(dist, flags, keys, vals, src=0)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dflags, dkeys, dvals = flags.to(dev), keys.to(dev), vals.to(dev)
    ctx = bonsai_amd.Context(0)
    ctx.set_encoder(int(CL["k"]), None, canonicalize=True)
    torch.cuda.synchronize()        # (the default stream's handle is 0 = "the context's own stream": order torch's copies before it)
    ctx.load_table_device(nb, dflags.data_ptr(), dkeys.data_ptr(), dvals.data_ptr(), bonsai_amd.LAYOUT_MINBUCKET,
                          torch.cuda.current_stream().cuda_stream)
    p = np.full(int(max(CL["tax_child"].max(), CL["tax_parent"].max())) + 1, 0xFFFFFFFF, dtype=np.uint32)
    p[CL["tax_child"]] = CL["tax_parent"]
    p[1] = 0                       # build_parent_map forces the root (util.h:780-781)
    ctx.load_taxonomy(p)
    pre = "p_" if paired else "s_"
    bases, offs, exp = CL[pre + "bases"], CL[pre + "offs"], CL[pre + "res"]
    inc = 2 if paired else 1
    n_units = (offs.size - 1) // inc
    lo, hi = shard.shard_range(n_units, rank, world)
    o = offs[lo * inc:hi * inc + 1]
    d_b = torch.from_numpy(bases[int(o[0]):int(o[-1])].copy()).to(dev)
    d_pad = torch.cat([d_b, torch.zeros(8, dtype=torch.uint8, device=dev)])           # readable to the next 4-byte boundary
    d_o = torch.from_numpy((o - o[0]).astype(np.int64)).to(dev)
    d_t = torch.zeros(hi - lo, dtype=torch.int32, device=dev)
    lens = np.diff(o.astype(np.int64))
    torch.cuda.synchronize()
    ctx.classify_device(d_pad.data_ptr(), d_o.data_ptr(), (hi - lo) * inc, int(o[-1] - o[0]), int(lens.max()) if lens.size else 0,
                        paired, d_t.data_ptr(), None, None, None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sizes = shard.shard_sizes(n_units, world)
    got = shard.gather_results(dist, d_t.cpu().to(torch.int64), sizes, dst=0)       # gloo: gather through the host
    if rank == 0:
        q.put(bool(np.array_equal(got.numpy().astype(np.uint32), exp[:, 0])))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("paired", [False, True])
def test_two_ranks_one_device_hip_path(paired):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (3 if paired else 2)
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    procs = [mctx.Process(target=_worker, args=(r, world, port, paired, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_self_launch_two_ranks():
    """bench.py --gpus 2 without torchrun: refuses on a 1-GPU node, and with the one-device debug mapping launches 2 ranks itself."""
    small = ["--genomes", "32", "--genome-len", "65536", "--log2-buckets", "22", "--reads", "40000", "--steps", "2", "--warmup", "1",
             "--no-cpu", "--no-probe"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, env=env, capture_output=True, timeout=600)
        assert p.returncode != 0 and b"refusing" in p.stderr
    env2 = dict(env, BNS_BENCH_ONE_DEVICE="1", BNS_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, env=env2, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["reads_per_gpu"] == 40000 and out["value"] > 0
    # the N>1 line carries every rank's kernel time and a parity sample taken from the GATHERED result
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and "error" not in out
    for r in out["per_rank"]:
        assert r["kernel_ms"] > 0 and r["parity_sample"]["reads"] == 40000 and r["parity_sample"]["gathered_vs_local_gpu_mismatches"] == 0


def _bench(args, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]), p.stderr.decode()


def test_bench_rccl_collectives_at_world_one():
    """The N>1 code path of bench.py over the REAL backend on the one GPU there is: torch.distributed 'nccl' (= RCCL) process group
    of one rank, the broadcast of the three khash tensors, of the read pool and of the table geometry, the asynchronous per-step
    gather, all_gather_object / all_reduce of the timings -- and the per-rank parity sample (gathered slice vs re-derived reads,
    on the GPU and with the CPU oracle) must be clean."""
    small = ["--genomes", "32", "--genome-len", "65536", "--log2-buckets", "22", "--reads", "60000", "--steps", "3", "--warmup", "1",
             "--no-probe", "--cpu-sample", "20000", "--rank-sample", "30000", "--no-ref"]
    out, err = _bench(small, {"BNS_BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and "error" not in out
    pr = out["per_rank"]
    assert len(pr) == 1 and pr[0]["parity_sample"]["reads"] == 30000
    assert pr[0]["parity_sample"]["gathered_vs_local_gpu_mismatches"] == 0 and pr[0]["parity_sample"]["gathered_vs_oracle_mismatches"] == 0
    assert out["parity_sample"]["mismatches"] == 0 and pr[0]["parity_sample"]["classified_frac"] > 0.5


def test_bench_strong_scaling_two_ranks():
    """--scaling strong --total-reads T: the total is sharded (uneven shards allowed), value counts T per step"""
    small = ["--genomes", "32", "--genome-len", "65536", "--log2-buckets", "22", "--scaling", "strong", "--total-reads", "50001",
             "--steps", "2", "--warmup", "1", "--no-cpu", "--no-probe", "--rank-sample", "5000"]
    out, err = _bench(["--gpus", "2"] + small, {"BNS_BENCH_ONE_DEVICE": "1", "BNS_BENCH_BACKEND": "gloo"})
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["total_reads_per_step"] == 50001
    assert sorted(r["reads"] for r in out["per_rank"]) == [25000, 25001] and "error" not in out
    assert abs(out["value"] - 50001 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6


def test_bench_eight_ranks_one_device():
    """bench.py --dry-run-world 8: the 8-rank job walked on the one GPU there is -- eight processes (gloo rendezvous, all on device 0), the
    db at 1/8 of its size broadcast from rank 0, eight shards, the double-buffered gather, and the per-rank parity sample of the gathered
    result against reads re-derived on rank 0; every rank says where it runs (device, PCI id, NUMA node, CPUs, collective library)"""
    out, err = _bench(["--dry-run-world", "8", "--genome-len", "65536", "--log2-buckets", "24", "--reads", "6400000"], {}, timeout=1500)
    assert out["n_gpus"] == 8 and "error" not in out and "dry_run" in out
    assert [r["rank"] for r in out["per_rank"]] == list(range(8))
    assert out["collectives"]["shard_units"] == [r["shard"]["units"] for r in out["per_rank"]] and sum(out["collectives"]["shard_units"]) == out["config"]["total_reads_per_step"]
    firsts = [r["shard"]["first_unit"] for r in out["per_rank"]]
    assert firsts == [sum(out["collectives"]["shard_units"][:i]) for i in range(8)]
    assert out["collectives"]["broadcast_bytes"]["keys"] == 8 << 21 and out["collectives"]["gather_bytes_per_step"] == 4 * out["config"]["total_reads_per_step"]
    for r in out["per_rank"]:
        assert r["kernel_ms"] > 0 and r["parity_sample"]["gathered_vs_local_gpu_mismatches"] == 0 and r["parity_sample"]["gathered_vs_oracle_mismatches"] == 0
        assert r["placement"]["pci"] not in ("", "?") and r["placement"]["cpus"]
    for rank in range(8):
        assert "rank %d of 8: device 0 (PCI " % rank in err
