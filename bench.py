#!/usr/bin/env python
"""bench.py -- reads/s classified (150 bp) on N MI355X, the metric BASELINE.json names.

One step = one pass of the classify hot path (pack -> k-mer extract -> table probe -> vote/resolve)
over one batch of synthetic 150 bp reads that is already resident in HBM as ASCII, exactly what
bns_classify_batch_device (include/bonsai_amd.h) consumes.  Workload = BASELINE.json configs[1] as named:
k=31, db built with w=50 entropy minimizers (bonsai build -w50 -e) from ~1k bacteria-sized genomes, in HBM
(synthetic: 1024 genomes x 2.6 Mb = 2.7e9 bases -> ~2.2e8 keys in 2^29 khash buckets, SURVEY 8d C2's key
count), 10M reads per GPU.  `--db-window 0 --genome-len 262144` is the heavier every-k-mer db rounds 1-2
were tuned on (2.3e8 keys from 2.7e8 bases: every k-mer of a read is in the db); both are reported in
DESIGN.md.  N>1: one process per GPU, db RCCL-broadcast from rank 0, reads sharded, per-step gather of the taxids to
rank 0.  Weak scaling by default (--reads per GPU); `--scaling strong --total-reads T` shards a fixed total (configs[3]:
"1B reads sharded").  An N>1 line carries `per_rank`: every rank's kernel time and roofline fraction, and a parity sample
per rank -- rank 0 regenerates the first reads of rank r's last batch from the seed, classifies them itself and with the CPU
oracle, and compares both with the slice of the GATHERED result that came from rank r: a wrong broadcast, shard or gather
cannot print a fine-looking number.  BNS_BENCH_FORCE_DIST=1 runs the same collectives at world size 1 (how a one-GPU box
executes the RCCL init / broadcast / gather calls).

`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks (one per GPU, RCCL over xGMI)
through torch.distributed.run on 127.0.0.1 and fails loudly when the node has fewer than N devices; under the driver's
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.  `n_gpus` in the output is the world size
RCCL actually formed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# (the HIP runtime maps a process's streams onto 4 hardware queues by default, and two streams on one queue take turns: torch's
# streams + the context's kernel / upload / result streams are more than four, and the text leg's uploads then queue behind the
# kernels they feed -- 116 instead of 171 M reads/s.  Read at the runtime's start, so before torch; the caller's own setting wins.
# `bonsai classify` does the same in main().)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of >= 1 s, so that the driver's GPU-busy sampling sees the kernel)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--len-dist", choices=["fixed", "hiseq", "miseq"], default="fixed",
                    help="read lengths: fixed (--read-len) or drawn from the lengths of the reference's kraken_benchmarks reads "
                         "(tests/golden/{HiSeq,MiSeq}_accuracy_300.fa: 21..101 bp, mean 96 / 35..251 bp, mean 230) -- configs[4]'s shape")
    ap.add_argument("--k", type=int, default=31, help="k-mer length (configs name 31; other values for the secondary lines)")
    ap.add_argument("--genomes", type=int, default=1024)
    ap.add_argument("--genome-len", type=int, default=2_621_440, help="bases per synthetic genome (2.6 Mb: a small bacterial genome)")
    ap.add_argument("--log2-buckets", type=int, default=29)
    ap.add_argument("--layout", choices=["bucket", "khash", "minbucket"], default="minbucket")
    ap.add_argument("--bucket-slots-log2", type=int, default=0, help="clustered / bucket table: exact log2 of its size in 16-byte slots (0 = automatic)")
    ap.add_argument("--table-buckets", type=int, default=0, help="clustered table: exact number of 128-byte home buckets (0 = automatic: sized from the key count)")
    ap.add_argument("--identity", type=int, default=0, choices=[0, 32, 52], help="clustered table: minimizer identity bits (0 = chosen from the key count)")
    ap.add_argument("--packed", action="store_true", help="hand the reads over packed (2-bit words + invalid-base flags, bns_pack_reads / "
                                                          "bns_classify_batch_packed_device) instead of ASCII")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: --reads per GPU; strong: --total-reads sharded over the GPUs")
    ap.add_argument("--total-reads", type=int, default=0, help="--scaling strong: reads per step over ALL GPUs (configs[3]: 1e9)")
    ap.add_argument("--rank-sample", type=int, default=100_000, help="N>1: reads of every rank's last batch that rank 0 re-derives and compares with the gathered result")
    ap.add_argument("--genome-model", choices=["uniform", "repeats"], default="uniform",
                    help="synthetic genomes: iid-uniform bases, or with tandem repeats, homopolymer / low-complexity tracts and 1-5 kb "
                         "mobile elements duplicated across unrelated genomes (what skews minimizer buckets)")
    ap.add_argument("--min-span", type=int, default=0, help="clustered table's minimizer window k - m: 0 = chosen from the db (default), 8 / 11 / 15")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads timed on the host oracle and compared with the GPU result (rank 0, N=1): about 10 s of CPU work")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ref", action="store_true", help="CPU leg: the port only, even when the reference's compiled functions (oracle/_ref/libbns_ref.so) are there "
                                                          "(what the GPU TEST tier passes: it compares against committed vectors, never against that file)")
    ap.add_argument("--paired", action="store_true")
    ap.add_argument("--spacing", default="", help="spaced seed as bonsai -s, e.g. 1x15,0x15 (configs[2])")
    ap.add_argument("--ablate", type=lambda x: int(x, 0), default=0, help="profiling only: bns_debug_set bits (ablation bits make results wrong)")
    ap.add_argument("--db-window", type=int, default=50,
                    help="the db holds the window minimizers only (bonsai build -w W): 50 = configs[1] as named; "
                         "0 (or anything <= k) = every k-mer")
    ap.add_argument("--db-score", choices=["lex", "entropy"], default="entropy", help="minimizer score for --db-window")
    ap.add_argument("--share-block", type=int, default=4096,
                    help="length of the blocks genomes share with their relatives (default 4096: a 150 bp read mostly sees one taxon; "
                         "smaller blocks put several taxa -- leaf, genus, phylum LCAs -- into one read's vote)")
    ap.add_argument("--no-probe", action="store_true", help="skip the standalone probe-kernel roofline leg")
    ap.add_argument("--probe-keys", type=int, default=1 << 27)
    ap.add_argument("--no-inflate", action="store_true", help="skip the BGZF-inflate leg (bns_inflate_members_device on 4096 members: reported, never `value`)")
    ap.add_argument("--no-text", action="store_true", help="skip the text-path leg (bns_classify_text on FASTQ text in page-locked memory: reported, never `value`)")
    ap.add_argument("--text-reads", type=int, default=6_000_000)   # (1.9 GB of FASTQ: under the call's 2^31 and long enough that the ends of the pipeline are a few per cent)
    ap.add_argument("--emulate-rank", type=int, default=-1,
                    help="ONE GPU doing the work of rank R of a --world W job, without a process group: R's shard of --total-reads "
                         "(--scaling strong) or R's weak-scaling batch, generated from the seeds the real rank would use.  The line "
                         "then reports that rank's reads/s (what the 8-GPU run replicates), n_gpus = 1")
    ap.add_argument("--world", type=int, default=8, help="--emulate-rank: the world size being emulated")
    ap.add_argument("--stream-load", action="store_true",
                    help="the khash arrays leave the device before the clustered table is laid out and are loaded back STREAMED from "
                         "host memory (bns_load_table): how a db whose arrays and table do not fit the HBM together is loaded "
                         "(8e9 keys: 210 GB of arrays, 221 GB table)")
    ap.add_argument("--save-db", default="", help="write the db this run built as DIR/bns.db + DIR/nodes.dmp (the reference's on-disk layout: database.h:33-56) "
                                                   "for runs of the CLI against a db of the benchmark's size (tools/r06_db_load.sh)")
    ap.add_argument("--save-reads", type=int, default=0, help="with --save-db (fixed-length single reads): the first N reads of the first batch as DIR/reads.fq "
                                                              "(FASTQ, 314 bytes per 150-bp record) and what the timed kernel says about them as DIR/taxa.u32")
    ap.add_argument("--dry-run-world", type=int, default=0,
                    help="walk the N-rank job on THIS node's one GPU: W ranks (gloo rendezvous, all on device 0), the db at 1/W of its size, "
                         "--reads / W per rank, a few steps -- launch, shard bounds, broadcast sizes, gather buffers and the per-rank parity "
                         "sample of the real run, none of its numbers (the line says so)")
    a = ap.parse_args()
    if a.dry_run_world > 0:
        W = a.dry_run_world
        shrink = max(0, int(np.ceil(np.log2(W))))
        a.gpus = W
        a.log2_buckets = max(20, a.log2_buckets - shrink)
        a.genomes = max(16, a.genomes // W)
        a.reads = max(20_000, a.reads // (W * 8))
        a.steps = min(a.steps, 3); a.warmup = min(a.warmup, 1)
        a.no_probe = a.no_text = a.no_inflate = a.no_ref = True
        a.cpu_sample = min(a.cpu_sample, 20_000); a.rank_sample = min(a.rank_sample, 10_000)
        os.environ["BNS_BENCH_ONE_DEVICE"] = "1"
        os.environ.setdefault("BNS_BENCH_BACKEND", "gloo")
    return a


def self_launch(a):
    """`python bench.py --gpus N` outside torchrun: become the launcher of N ranks on this node."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus and os.environ.get("BNS_BENCH_ONE_DEVICE") != "1":
        sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to report a smaller run "
                         "under that label\n" % (a.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------
# synthetic world: 4-ary taxonomy over the genomes, genomes share blocks with their
# siblings / cousins so the db holds LCAs at every level.
# ------------------------------------------------------------------------------------------------
def make_taxonomy(n_genomes):
    """levels of a 4-ary tree; leaves are the genomes.  Returns (parent array, leaf ids)."""
    sizes = [n_genomes]
    while sizes[-1] > 1:
        sizes.append((sizes[-1] + 3) // 4)
    sizes = sizes[::-1]                       # root level first (size 1)
    base, nxt = [], 1
    for s in sizes:
        base.append(nxt)
        nxt += s
    parent = np.full(nxt, 0xFFFFFFFF, dtype=np.uint32)
    parent[1] = 0
    for lvl in range(1, len(sizes)):
        ids = base[lvl] + np.arange(sizes[lvl])
        parent[ids] = base[lvl - 1] + np.arange(sizes[lvl]) // 4
    leaves = (base[-1] + np.arange(n_genomes)).astype(np.uint32)
    return parent, leaves


def make_pool(n_genomes, genome_len, device, seed, B=4096):
    """uint8 codes 0..3, genome g = pool[g*G:(g+1)*G].  Block (g,b) is shared by the 4^s genomes of g's
    level-s group, s drawn per (64-genome group, b): 0 w.p. 13/16, 1: 1/8, 2: 1/32, 3: 1/32."""
    nb = genome_len // B
    rng = np.random.default_rng(seed)
    g = np.arange(n_genomes)[:, None]
    b = np.arange(nb)[None, :]
    u = rng.random((max(1, (n_genomes + 63) // 64), nb))[g // 64, b]
    s = np.where(u < 13 / 16, 0, np.where(u < 15 / 16, 1, np.where(u < 31 / 32, 2, 3)))
    key = (s.astype(np.int64) << 48) | ((g >> (2 * s)).astype(np.int64) << 24) | b
    uniq, inv = np.unique(key, return_inverse=True)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    blocks = torch.randint(0, 4, (uniq.size, B), dtype=torch.uint8, device=device, generator=gen)
    src = torch.from_numpy(inv.reshape(-1).astype(np.int64)).to(device)
    return blocks[src].reshape(-1)


def add_repeats(pool, n_genomes, G, seed):
    """--genome-model repeats: overwrite stretches of the iid-uniform genomes (in place) with what real genomes have and uniform
    ones lack -- per genome and per 8192-base slot at most one edit, so edits never overlap:
      * tandem repeats / low-complexity tracts: unit 1 (homopolymer), 2-6 (micro-satellite) or 10-50 bases, tract 60-1500 bases
        (one slot in 40);
      * mobile elements: 64 elements of 1-5 kb, each copied into ~6 % of the genomes whatever their taxonomy (one slot in 80):
        their k-mers get LCAs near the root, and the k-mers across every insertion boundary hang on the element's minimizers.
    Returns a dict of what was done (goes into the bench line)."""
    SL = 8192
    ns = G // SL
    if ns < 8:
        return {"model": "repeats", "note": "genomes too short for any edit"}
    rng = np.random.default_rng(seed)
    n_tr, n_me, E = max(1, ns // 40), max(1, ns // 80), 64
    dev = pool.device
    units = np.array([1, 1, 2, 2, 3, 4, 5, 6, 10, 20, 35, 50])
    el_len = rng.integers(1000, 5001, size=E)
    el_off = np.concatenate([[0], np.cumsum(el_len)])
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    elements = torch.randint(0, 4, (int(el_off[-1]),), dtype=torch.uint8, device=dev, generator=gen)
    g = np.repeat(np.arange(n_genomes), n_tr + n_me)
    slot = np.concatenate([rng.permutation(ns)[:n_tr + n_me] for _ in range(n_genomes)])
    is_me = np.tile(np.arange(n_tr + n_me) >= n_tr, n_genomes)
    unit = units[rng.integers(0, units.size, size=g.size)]
    which = rng.integers(0, E, size=g.size)
    lens = np.where(is_me, el_len[which], rng.integers(60, 1501, size=g.size))
    start = g.astype(np.int64) * G + slot.astype(np.int64) * SL + rng.integers(0, SL - 5000 - 1, size=g.size)
    first = np.cumsum(lens) - lens
    e = np.repeat(np.arange(g.size), lens)
    j = np.arange(int(lens.sum())) - first[e]
    dst = torch.from_numpy(start[e] + j).to(dev)
    me = torch.from_numpy(is_me[e]).to(dev)
    src_self = torch.from_numpy(start[e] + j % unit[e]).to(dev)
    src_el = torch.from_numpy(np.where(is_me[e], el_off[which[e]] + j, 0)).to(dev)
    vals = torch.where(me, elements[src_el], pool[src_self])          # (gathered from the untouched pool, then scattered)
    pool[dst] = vals
    return {"model": "repeats", "tandem_tracts": int((~is_me).sum()), "homopolymer_tracts": int(((unit == 1) & ~is_me).sum()),
            "mobile_elements": E, "mobile_element_copies": int(is_me.sum()), "edited_bases": int(lens.sum()),
            "edited_frac": float(lens.sum()) / float(n_genomes * G)}


def codes_to_ascii(c):
    # A=65 C=67 G=71 T=84 through a 4-entry table (uint8 in, uint8 out: no wide temporaries for multi-GB pools)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=c.device)
    if c.dim() == 1 and c.numel() > (1 << 30):
        out = torch.empty_like(c)
        for s0 in range(0, c.numel(), 1 << 30):
            out[s0:s0 + (1 << 30)] = lut[c[s0:s0 + (1 << 30)].long()]
        return out
    return lut[c.long()]


_SHA = None


def lib_source_sha256():
    """sha256 over the device library's sources: what ties a committed counter profile (profiles/traffic.json) to the binary"""
    global _SHA
    if _SHA is None:
        import hashlib
        h = hashlib.sha256()
        for f in ("bns_kernels.hip", "bns_device.hpp", "bns_kernels.hpp", "bns_api.hip"):
            h.update(open(os.path.join(ROOT, "bonsai_amd", "csrc", f), "rb").read())
        _SHA = h.hexdigest()
    return _SHA


def rank_placement(ctx, local, backend):
    """where this rank runs: its device's PCI id and NUMA node, the CPUs this process may use, the collective library and its version --
    one line per rank on stderr and `per_rank[].placement` in the N>1 line, so that a multi-GPU record says what it ran on"""
    import ctypes as C
    buf = C.create_string_buffer(64)
    pci = "?"
    try:
        if ctx.L.bns_device_pci_bus_id(local, buf, 64) == 0:
            pci = buf.value.decode()
    except Exception:
        pass
    numa = "?"
    try:
        numa = open("/sys/bus/pci/devices/%s/numa_node" % pci.lower()).read().strip()
    except Exception:
        pass
    cpus = sorted(os.sched_getaffinity(0))
    spans, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        spans.append("%d-%d" % (cpus[i], cpus[j]) if j > i else "%d" % cpus[i]); i = j + 1
    rccl = ""
    if backend == "nccl":
        try:
            rccl = "RCCL %s" % ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = "RCCL ?"
    return {"pci": pci, "numa_node": numa, "cpus": ",".join(spans), "backend": {"nccl": "nccl (RCCL)"}.get(backend, backend), "rccl": rccl}


def effective_cores():
    """Host cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one (the GPU
    boxes show 256 CPUs under a 16-CPU quota; 256 OpenMP threads on that are slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(-(-int(quota) // int(period)))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def inflate_leg(lib, device, n_members=4096, distinct=128):
    """The BGZF-inflate row of the host path (never `value`): n_members members of 65 280 bytes of FASTQ text each, zlib level
    6 like bgzip's, inflated by ONE bns_inflate_members_device call into device memory -- the batch size a reader hands over; the
    kernel's own HIP-event time.  Status and CRC-32 of every member checked against zlib's."""
    import ctypes as C
    import zlib
    rng = np.random.default_rng(5)
    base = []
    for k in range(distinct):
        m = 208
        rec = np.empty((m, 314), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        idx = np.arange(m) + k * m
        for j in range(7):
            rec[:, 8 - j] = ord("0") + (idx // 10 ** j) % 10
        rec[:, 9] = 10
        rec[:, 10:160] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, 150))]
        rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
        rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
        rec[:, 313] = 10
        t = rec.tobytes()[:65280]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        base.append((t, co.compress(t) + co.flush()))
    h = C.c_void_p()
    if lib.bns_inflater_create(device, C.byref(h)) != 0:
        return None
    try:
        in_len = np.array([len(base[i % distinct][1]) for i in range(n_members)], dtype=np.uint32)
        in_off = np.zeros(n_members, dtype=np.uint64); in_off[1:] = np.cumsum(in_len[:-1].astype(np.uint64))
        cb = int(in_off[-1]) + int(in_len[-1])
        pc = C.c_void_p()
        if lib.bns_inflater_host_alloc(h, cb + 64, C.byref(pc)) != 0:
            return None
        comp = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint8)), shape=(cb + 64,))
        for i in range(n_members):
            c = base[i % distinct][1]
            comp[int(in_off[i]):int(in_off[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
        out_len = np.full(n_members, 65280, dtype=np.uint32)
        out_off = np.arange(n_members, dtype=np.uint64) * 65280
        crc = np.zeros(n_members, dtype=np.uint32); status = np.zeros(n_members, dtype=np.uint32)
        d_text = torch.empty(n_members * 65280 + 64, dtype=torch.uint8, device="cuda:%d" % device)
        u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        kms, call = 1e9, 1e9
        lib.bns_inflater_last_kernel_ms.restype = C.c_float
        for rep in range(4):
            t0 = time.perf_counter()
            rc = lib.bns_inflate_members_device(h, pc, cb, in_off.ctypes.data_as(u64p), in_len.ctypes.data_as(u32p), out_off.ctypes.data_as(u64p), out_len.ctypes.data_as(u32p),
                                                n_members, C.c_void_p(d_text.data_ptr()), n_members * 65280, crc.ctypes.data_as(u32p), status.ctypes.data_as(u32p))
            dt = time.perf_counter() - t0
            if rc != 0:
                return {"error": "bns_inflate_members_device rc %d" % rc}
            if rep:
                kms = min(kms, float(lib.bns_inflater_last_kernel_ms(h))); call = min(call, dt)
        want = np.array([zlib.crc32(base[i % distinct][0]) & 0xFFFFFFFF for i in range(distinct)], dtype=np.uint32)
        bad = int((status != 0).sum() + (crc != want[np.arange(n_members) % distinct]).sum())
        first = bytes(d_text[:65280].cpu().numpy()) == base[0][0]
        tb = n_members * 65280
        lib.bns_inflater_host_free(h, pc)
        return {"entry": "bns_inflate_members_device", "members": n_members, "text_bytes": tb, "compressed_bytes": cb, "kernel": "inflate_wave_kernel", "kernel_ms": kms,
                "text_GB_per_s_kernel": tb / kms / 1e6, "call_ms": call * 1e3, "text_GB_per_s_call": tb / call / 1e9, "members_wrong": bad + (0 if first else 1),
                "note": "one batch of BGZF-sized members (FASTQ text, zlib level 6) from page-locked memory into device memory; kernel time by HIP events on the "
                        "inflater's stream; the call adds the upload of the compressed bytes and the status words back.  The host-ingest row for BGZF input: "
                        "reported beside `value`, never it."}
    finally:
        lib.bns_inflater_destroy(h)


def gz_stream_leg(lib, device, n_records=1_000_000):
    """The gzip-stream row of the host path (never `value`): ONE DEFLATE stream of FASTQ text (zlib level 6, written the way pigz
    does: 4 MiB pieces deflated side by side on threads, each primed with the 32 KiB in front of it), entered at block headers found on the
    device, inflated into device memory by bns_inflate_stream_device call by call; the kernels' HIP-event time.  The text's CRC-32 and its
    first MiB against what went in."""
    import ctypes as C
    import zlib
    from bonsai_amd._lib import GzResult
    rng = np.random.default_rng(7)
    m = n_records
    rec = np.empty((m, 314), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    for j in range(7):
        rec[:, 8 - j] = ord("0") + (np.arange(m) // 10 ** j) % 10
    rec[:, 9] = 10
    rec[:, 10:160] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, 150))]
    rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
    # (qualities in eight bins, runs of six: 20 000 distinct lines dealt at random -- the same line twice within DEFLATE's reach is rare)
    pool = np.frombuffer(b"#,5:AFIJ", dtype=np.uint8)[np.repeat(rng.choice(8, size=(20000, 25), p=[0.02, 0.03, 0.05, 0.05, 0.1, 0.2, 0.25, 0.3]), 6, axis=1)]
    rec[:, 163:313] = pool[rng.integers(0, 20000, size=m)]
    rec[:, 313] = 10
    text = rec.tobytes()
    del rec
    from concurrent.futures import ThreadPoolExecutor
    PIECE = 4 << 20
    n_pieces = (len(text) + PIECE - 1) // PIECE

    def deflate_piece(i):
        lo = i * PIECE
        co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, text[lo - 32768:lo]) if i else zlib.compressobj(6, zlib.DEFLATED, -15)
        return co.compress(text[lo:lo + PIECE]) + co.flush(zlib.Z_FINISH if i == n_pieces - 1 else zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(8) as ex:                        # (zlib lets go of the interpreter lock)
        body = b"".join(ex.map(deflate_piece, range(n_pieces)))
    h = C.c_void_p()
    if lib.bns_inflater_create(device, C.byref(h)) != 0:
        return None
    try:
        cb = len(body)
        pc = C.c_void_p()
        if lib.bns_inflater_host_alloc(h, cb + 64, C.byref(pc)) != 0:
            return None
        C.memmove(pc, body, cb)
        d_text = torch.empty(len(text) + 4096, dtype=torch.uint8, device="cuda:%d" % device)
        d_win = torch.empty(32768, dtype=torch.uint8, device="cuda:%d" % device)
        lib.bns_inflater_last_kernel_ms.restype = C.c_float
        if lib.bns_inflate_stream_reserve(h, cb) != 0:
            return {"error": "bns_inflate_stream_reserve"}
        best_k, best_c, calls, chunks, chained, crc, total = 1e9, 1e9, 0, 0, 0, 0, 0
        for rep in range(3):
            pos, fresh, kms, cms, calls, chunks, chained, crc, total = 0, True, 0.0, 0.0, 0, 0, 0, 0, 0
            while True:
                b0 = pos // 8
                res = GzResult()
                t0 = time.perf_counter()
                rc = lib.bns_inflate_stream_device(h, C.c_void_p(pc.value + b0), cb - b0, pos - 8 * b0, None if fresh else C.c_void_p(d_win.data_ptr()),
                                                   C.c_void_p(d_text.data_ptr() + total), len(text) + 4096 - total, C.c_void_p(d_win.data_ptr()), C.byref(res))
                cms += (time.perf_counter() - t0) * 1e3
                if rc != 0 or res.status != 0:
                    return {"error": "bns_inflate_stream_device rc %d status %d" % (rc, res.status)}
                kms += float(lib.bns_inflater_last_kernel_ms(h))
                crc = lib.bns_crc32_combine(crc, res.crc32, res.text_bytes); total += res.text_bytes
                calls += 1; chunks += res.n_chunks; chained += res.n_chained
                pos = 8 * b0 + res.end_bit; fresh = False
                if res.member_end or calls > 64:
                    break
            if rep:
                best_k = min(best_k, kms); best_c = min(best_c, cms)
        same = total == len(text) and crc == (zlib.crc32(text) & 0xFFFFFFFF) and bytes(d_text[:1 << 20].cpu().numpy()) == text[:1 << 20]
        lib.bns_inflater_host_free(h, pc)
        return {"entry": "bns_inflate_stream_device", "text_bytes": len(text), "compressed_bytes": cb, "calls": calls, "chunks_with_a_header": chunks, "chunks_taken": chained,
                "kernels": "gz_search / decode / valid / compose / groups / translate / crc_kernel", "kernels_ms": best_k, "text_GB_per_s_kernels": len(text) / best_k / 1e6,
                "calls_ms": best_c, "text_GB_per_s_calls": len(text) / best_c / 1e6, "text_wrong": 0 if same else 1,
                "wavefronts": "%d chunks: a third of the 3072 decoder wavefronts the CLI's 288 MiB calls fill (profiles/r06_gz.txt)" % chunks,
                "note": "one gzip member (FASTQ text, qualities in eight bins, zlib level 6) from page-locked memory into device memory: block headers found by a "
                        "wavefront per chunk, 16-bit symbols for the unknown 32 KiB in front, chunks that chain exactly; kernel time by HIP events on the inflater's "
                        "stream; the calls add the upload of the compressed bytes.  The host-ingest row for plain .gz input: reported beside `value`, never it."}
    finally:
        lib.bns_inflater_destroy(h)


def text_leg(ctx, a, bases_dev, offsets_dev, taxon_dev):
    """The host-ingest row at the C ABI (never `value`): the first --text-reads reads of the timed batch written out as FASTQ
    TEXT in page-locked host memory -> bns_classify_text (upload in pieces, record boundaries / names / 2-bit words by kernels, classify)
    -> taxon per read back; best of 3, compared with the timed launch's own result for those reads."""
    import ctypes as C
    from bonsai_amd import _lib
    T = min(a.text_reads, a.reads)
    L = a.read_len
    ho = offsets_dev[:T + 1].cpu().numpy().astype(np.int64)
    hb = bases_dev[:int(ho[-1])].cpu().numpy()
    if not np.all(np.diff(ho) == L):
        return None
    rec_len = 17 + 2 * L                                  # "@r<10 digits>\n" seq "\n+\n" qual "\n"
    rec = np.empty((T, rec_len), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(T)
    for d in range(10):
        rec[:, 2 + d] = ord("0") + (idx // 10 ** (9 - d)) % 10
    rec[:, 12] = 10
    rec[:, 13:13 + L] = hb.reshape(T, L)
    rec[:, 13 + L] = 10; rec[:, 14 + L] = ord("+"); rec[:, 15 + L] = 10
    rec[:, 16 + L:16 + 2 * L] = ord("I")
    rec[:, 16 + 2 * L] = 10
    Lb = ctx.L
    nbytes = rec.size
    pt = C.c_void_p(); po = C.c_void_p()
    if Lb.bns_host_alloc(ctx.h, nbytes + 64, C.byref(pt)) != 0 or Lb.bns_host_alloc(ctx.h, 4 * (T + 16), C.byref(po)) != 0:
        return None
    np.frombuffer((C.c_uint8 * nbytes).from_address(pt.value), dtype=np.uint8)[:] = rec.reshape(-1)
    out_t = np.frombuffer((C.c_uint8 * (4 * (T + 16))).from_address(po.value), dtype=np.uint32)
    o = _lib.TextOut(); o.taxon = po.value
    info = _lib.TextInfo()
    ptrs = (C.c_void_p * 1)(pt.value)
    sizes = np.array([nbytes], dtype=np.uint64)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        rc = Lb.bns_classify_text(ctx.h, ptrs, sizes.ctypes.data_as(C.POINTER(C.c_uint64)), 1, 0xFFFFFFFFFFFFFFFF, _lib.TEXT_FINAL, T + 16, C.byref(o), C.byref(info))
        e = time.perf_counter() - t0
        if rc != 0 or info.status != 0 or info.n_records != T:
            return {"error": "bns_classify_text rc %d status %d records %d" % (rc, info.status, info.n_records)}
        if best is None or e < best:
            best, parts = e, (float(info.ms_parse), float(info.ms_classify), int(info.n_slices), int(info.n_launches))
    mism = int((out_t[:T] != taxon_dev[:T].cpu().numpy().astype(np.uint32)).sum())
    Lb.bns_host_free(ctx.h, pt); Lb.bns_host_free(ctx.h, po)
    return {"entry": "bns_classify_text", "reads": T, "text_bytes": int(nbytes), "reads_per_s": T / best, "text_GB_per_s": nbytes / best / 1e9,
            "mismatches_vs_timed_launch": mism, "pcie_inclusive": True, "call_ms": best * 1e3, "ms_parse_kernels": parts[0], "ms_classify": parts[1], "slices": parts[2], "classify_launches": parts[3],
            "note": "FASTQ text in page-locked host memory -> upload in 64 MiB pieces -> records, names and 2-bit words by kernels (csrc/bns_ingest.hip) -> "
                    "classify -> taxon back; best of 3 after a warm-up call.  The host-ingest row at the C ABI: reported beside `value`, never it."}


def probe_leg(ctx, a, dev, stream, flags, keys, nb, khash_load):
    """probe_kernel<minbucket> over keys without locality; HIP-event time of the kernel itself (the library's own events on the
    launch stream).  Bytes: every lookup reads one whole 128-byte bucket (what the memory system moves) and needs 16 of them
    (SURVEY 8d's algorithmic figure)."""
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    n = a.probe_keys
    idx = torch.randint(0, nb, (n,), device=dev, generator=gen)
    q = keys[idx]
    present = ((flags[idx >> 4] >> ((idx & 15) << 1)) & 3) == 0
    rnd = torch.randint(0, 1 << 62, (n,), device=dev, generator=gen, dtype=torch.int64)
    use_rnd = ~present | (torch.rand(n, device=dev, generator=gen) > 0.5 / max(1e-9, khash_load))
    q = torch.where(use_rnd, rnd, q).contiguous()
    del idx, rnd, use_rnd, present
    out_v = torch.empty(n, dtype=torch.int32, device=dev)
    out_f = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.probe_device(q.data_ptr(), n, out_v.data_ptr(), out_f.data_ptr(), stream)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    iters = 5
    for _ in range(iters):
        ctx.probe_device(q.data_ptr(), n, out_v.data_ptr(), out_f.data_ptr(), stream)
    torch.cuda.synchronize()
    sm, c = ctx.timing_summary()
    ctx.set_timing(False)
    ms = sm / max(1, c)
    lps = n / (ms * 1e-3)
    # bytes the memory system moves per lookup: the counter figure of the committed probe profile when it was taken on THIS
    # source (profiles/probe_traffic.json, written by tools/summarize_prof.py), else the one bucket line a lookup cannot avoid
    fetched, fetched_src = 128.0, "one 128-byte bucket line per lookup (no counter profile for this source)"
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "probe_traffic.json")))
        if pj.get("source_sha256") == lib_source_sha256() and pj.get("keys") == n:
            fetched, fetched_src = float(pj["hbm_read_bytes_per_lookup"]), pj.get("source", "profiles/probe_traffic.json")
    except Exception:
        pass
    return {"kernel": "probe_kernel<minbucket>", "keys": n, "hit_frac": float(out_f.float().mean().item()), "kernel_ms": ms,
            "launches_timed": c, "lookups_per_s": lps, "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "fetched_bytes_per_lookup": fetched, "fetched_bytes_source": fetched_src,
            "achieved_fetched": lps * fetched / 1e9, "frac_fetched": lps * fetched / 1e9 / HBM_PEAK_GBS,
            "alg_bytes_per_lookup": 16, "achieved": lps * 16 / 1e9, "frac": lps * 16 / 1e9 / HBM_PEAK_GBS,
            "note": "keys without read locality: one 128-byte bucket fetched per lookup (frac_fetched = share of the 8 TB/s "
                    "spec the fetches occupy); frac = SURVEY 8d's algorithmic 16 B per lookup"}


def gen_reads(pool, n, L, n_genomes, G, device, seed, sub_rate=0.01, n_rate=0.001, paired=False):
    """n reads of L bases from random positions and strands of the genome pool, with substitutions and Ns.  paired: reads 2i and
    2i+1 are the two ends of one fragment (insert size uniform in [L, 3L], mate 2 on the opposite strand), as a sequencer makes them."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty(n * L, dtype=torch.uint8, device=device)
    j = torch.arange(L, device=device)
    chunk = 1 << 20
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        if paired:
            h = m // 2
            g = torch.randint(0, n_genomes, (h,), device=device, generator=gen)
            ins = torch.randint(L, min(3 * L, G) + 1, (h,), device=device, generator=gen)
            fs = (torch.rand(h, device=device, generator=gen) * (G - ins + 1).to(torch.float32)).to(torch.int64).clamp_(min=0)
            fs = torch.minimum(fs, G - ins)
            ff = torch.rand(h, device=device, generator=gen) < 0.5           # which end comes first
            left, right = g * G + fs, g * G + fs + ins - L
            start = torch.stack([torch.where(ff, right, left), torch.where(ff, left, right)], dim=1).reshape(-1)
            flip = torch.stack([ff, ~ff], dim=1).reshape(-1)
        else:
            g = torch.randint(0, n_genomes, (m,), device=device, generator=gen)
            off = torch.randint(0, G - L + 1, (m,), device=device, generator=gen)
            start = g * G + off
            flip = torch.rand(m, device=device, generator=gen) < 0.5
        idx = start[:, None] + torch.where(flip[:, None], L - 1 - j, j)
        codes = pool[idx]
        codes = torch.where(flip[:, None], 3 - codes, codes)
        sub = torch.rand((m, L), device=device, generator=gen) < sub_rate
        rnd = torch.randint(0, 4, (m, L), dtype=torch.uint8, device=device, generator=gen)
        codes = torch.where(sub, rnd, codes)
        asc = codes_to_ascii(codes)
        nm = torch.rand((m, L), device=device, generator=gen) < n_rate
        asc = torch.where(nm, torch.full_like(asc, 78), asc)
        out[s0 * L:(s0 + m) * L] = asc.reshape(-1)
        del idx, codes, sub, rnd, asc, nm
    return out


def empirical_lengths(name):
    """read lengths of the first 300 records of the reference's kraken_benchmarks/<name>_accuracy.fa (a committed fixture)"""
    lens, cur = [], None
    with open(os.path.join(ROOT, "tests", "golden", "%s_accuracy_300.fa" % name), "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur is not None:
                    lens.append(cur)
                cur = 0
            else:
                cur += len(line.strip())
    if cur is not None:
        lens.append(cur)
    return np.array(lens, dtype=np.int64)


def truncate_reads(bases, n, L, lens_pool, device, seed, n_total=None):
    """ragged batch out of a fixed-length one: read i keeps its first len[i] bases, len drawn from lens_pool (<= L).
    n_total: draw the lengths of a batch of that many reads and use the first n of them (re-deriving the head of a larger batch)"""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    pool_t = torch.from_numpy(lens_pool).to(device)
    lens = pool_t[torch.randint(0, pool_t.numel(), (n_total or n,), device=device, generator=gen)][:n]
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(lens, 0)
    out = torch.empty(int(offsets[-1].item()) + 8, dtype=torch.uint8, device=device)     # (+8: readable past the end)
    j = torch.arange(L, device=device)
    chunk = 1 << 21
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        keep = j[None, :] < lens[s0:s0 + m, None]
        out[int(offsets[s0].item()):int(offsets[s0 + m].item())] = bases[s0 * L:(s0 + m) * L].reshape(m, L)[keep]
    return out, offsets, lens


def make_batch(pool, n, L, NG, G, dev, rank, i, paired, lens_pool, head=None):
    """batch i of `rank`: (ASCII bases, offsets, total bases, lens or None).  head = S: only its first S reads, bit-identical to
    the head of the full batch (same generator seeds, same draw shapes: gen_reads works in chunks of 2^20 reads)."""
    n_gen = n if head is None else min(n, 1 << 20)
    b = gen_reads(pool, n_gen, L, NG, G, dev, seed=43 + 2 * rank + i, paired=paired)
    S = n if head is None else min(head, n_gen)
    if lens_pool is None:
        return b[:S * L], torch.arange(S + 1, device=dev, dtype=torch.int64) * L, S * L, None
    tb, to, lens = truncate_reads(b, S, L, lens_pool, dev, seed=143 + 2 * rank + i, n_total=n)
    return tb, to, int(to[-1].item()), lens


def oracle_check(O, table, tax, k, gaps, paired, d_bases, d_offsets, S, got_t, got_m, got_a, nthreads, repeat=1):
    """CPU oracle over the first S reads of a device batch vs the GPU's per-unit results; returns (mismatches, best seconds, taxon array)"""
    ho = d_offsets[:S + 1].cpu().numpy().astype(np.uint64)
    hb = d_bases[:int(ho[-1])].cpu().numpy()
    best = None
    for _ in range(repeat):
        t1 = time.perf_counter()
        res = O.classify_batch(table, tax, k, hb, ho, paired=paired, gaps=gaps, spaced_intended=True, nthreads=nthreads)
        e = time.perf_counter() - t1
        best = e if best is None or e < best else best
    su = S // 2 if paired else S
    mism = int((got_t[:su].cpu().numpy().view(np.uint32) != res["taxon"]).sum())
    if got_m is not None:
        mism += int((got_m[:su].cpu().numpy().view(np.uint32) != res["missing"]).sum() + (got_a[:su].cpu().numpy().view(np.uint32) != res["ambig"]).sum())
    return mism, best, res["taxon"]


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but the launcher formed %d rank(s); reporting n_gpus = %d\n" % (a.gpus, world, world))
        a.gpus = world
    dist = None
    # debugging aids for a 1-GPU box: BNS_BENCH_ONE_DEVICE=1 maps every rank to GPU 0, BNS_BENCH_BACKEND=gloo avoids
    # RCCL (which refuses two ranks on one device); BNS_BENCH_FORCE_DIST=1 forms the process group -- and runs every
    # collective of the N>1 path, over RCCL -- at world size 1.  The driver's multi-GPU runs use none of them.
    backend = os.environ.get("BNS_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("BNS_BENCH_FORCE_DIST") == "1"
    if os.environ.get("BNS_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ONE explicit stream for everything: torch's tensor ops, the library's kernels (it is handed to every bns_*_device call)
    # and the collectives' stream dependencies.  (The legacy default stream's handle is 0, which the library reads as "use the
    # context's own non-blocking stream" -- work there is invisible to torch's ordering, and a gather issued after a classify
    # would not wait for it.)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()               # what the backend really formed
    multi = dist is not None
    # shard identity: whose reads this process classifies.  --emulate-rank R --world W: rank R of a W-rank job on this one GPU.
    srank, sworld = rank, world
    if a.emulate_rank >= 0:
        if multi or not (0 <= a.emulate_rank < a.world):
            sys.stderr.write("bench.py: --emulate-rank needs a single process and 0 <= R < --world\n")
            return 2
        srank, sworld = a.emulate_rank, a.world

    def bcast(t, src=0):                            # gloo has no device collectives: stage through the host (debug path only)
        if backend == "nccl":
            dist.broadcast(t, src=src)
        else:
            h = t.cpu(); dist.broadcast(h, src=src); t.copy_(h)

    import bonsai_amd                     # after torch: shares torch's HIP runtime (same soname)
    ctx = bonsai_amd.Context(local)
    placement = rank_placement(ctx, local, backend if (world > 1 or force_dist) else "none")
    if world > 1 or force_dist or a.emulate_rank >= 0:
        sys.stderr.write("bench.py: rank %d of %d: device %d (PCI %s, NUMA node %s), this process on CPUs %s; collectives: %s%s\n" % (
            rank, world, local, placement["pci"], placement["numa_node"], placement["cpus"], placement["backend"],
            (" " + placement["rccl"]) if placement["rccl"] else ""))
    k, L = a.k, a.read_len
    gaps = None
    if a.spacing:
        from bonsai_amd import hostio
        gaps = hostio.parse_spacing(a.spacing, k)
    ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=True)
    if a.ablate:
        ctx.debug_set(a.ablate)
    parent, leaves = make_taxonomy(a.genomes)
    ctx.load_taxonomy(parent)
    G, NG = a.genome_len, a.genomes
    nb = 1 << a.log2_buckets
    stream = tstream.cuda_stream
    assert stream != 0

    # ---- db: built on rank 0's GPU (update_lca_map semantics), RCCL-broadcast, loaded everywhere
    t_setup = time.time()
    flags = torch.empty(max(1, nb >> 4), dtype=torch.int32, device=dev)
    keys = torch.empty(nb, dtype=torch.int64, device=dev)
    vals = torch.empty(nb, dtype=torch.int32, device=dev)
    pool = torch.empty(NG * G, dtype=torch.uint8, device=dev)
    hdr = np.zeros(4, dtype=np.uint64)
    genome_info = {"model": "uniform"}
    if rank == 0:
        pool = make_pool(NG, G, dev, seed=7, B=a.share_block)
        if a.genome_model == "repeats":
            genome_info = add_repeats(pool, NG, G, seed=11)
        pool_ascii = codes_to_ascii(pool)
        g_off = (torch.arange(NG + 1, device=dev, dtype=torch.int64) * G)
        taxid = torch.from_numpy(leaves.astype(np.int32)).to(dev)
        torch.cuda.synchronize()
        if a.db_window > k:                          # bonsai build -w W [-e]: the db holds the window minimizers only
            ctx.set_window(a.db_window, bonsai_amd.SCORE_ENTROPY_PATH if a.db_score == "entropy" else bonsai_amd.SCORE_LEX)
        hdr = ctx.build_table_device(pool_ascii.data_ptr(), g_off.data_ptr(), NG, NG * G, taxid.data_ptr(), nb,
                                     flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), stream)
        if a.db_window > k:
            ctx.set_window(0, bonsai_amd.SCORE_LEX)   # classify itself always runs unwindowed (bonsai.cpp:152-153)
        del pool_ascii
        if a.save_db:
            os.makedirs(a.save_db, exist_ok=True)
            torch.cuda.synchronize()
            with open(os.path.join(a.save_db, "bns.db"), "wb") as f:
                np.array([k, k], dtype=np.uint32).tofile(f)
                np.zeros(k - 1, dtype=np.uint8).tofile(f)
                np.asarray(hdr, dtype=np.uint64)[:4].tofile(f)
                for t in (flags, keys, vals):
                    t.cpu().numpy().tofile(f)
            with open(os.path.join(a.save_db, "nodes.dmp"), "w") as f:
                for c, p in enumerate(parent):
                    if p != 0xFFFFFFFF and c != 0:
                        f.write("%d\t|\t%d\t|\tno rank\t|\t\t|\n" % (c, p))
    if multi:
        for t in (flags, keys, vals):                # RCCL over xGMI, one message per array (bonsai_amd/shard.py: broadcast_table)
            bcast(t)
        bcast(pool)                                  # read generator input (bench only)
        torch.cuda.synchronize()
    # ---- reads: this rank's shard, two alternating batches (strong scaling: ONE batch, its size is the point).  Generated
    # BEFORE the classify table is laid out: the generator's temporaries are large for long reads
    if a.scaling == "strong":
        if a.total_reads <= 0:
            sys.stderr.write("bench.py: --scaling strong needs --total-reads\n")
            return 2
        from bonsai_amd import shard
        tu = a.total_reads // (2 if a.paired else 1)
        lo_u, hi_u = shard.shard_range(tu, srank, sworld)
        n = (hi_u - lo_u) * (2 if a.paired else 1)
    else:
        n = a.reads - (a.reads % 2)
    lens_pool = None
    if a.len_dist != "fixed":
        lens_pool = empirical_lengths("HiSeq" if a.len_dist == "hiseq" else "MiSeq")
        L = a.read_len = int(lens_pool.max())
    n_batches = 1 if a.scaling == "strong" else 2
    made = [make_batch(pool, n, L, NG, G, dev, srank, i, a.paired, lens_pool) for i in range(n_batches)]
    batches = [m[0] for m in made]
    offsets_l = [m[1] for m in made]
    totals = [m[2] for m in made]
    mean_alg = None
    comb = k + (int(gaps.sum()) if gaps is not None else 0)
    if lens_pool is not None:
        alg = [float((torch.clamp(m[3] - comb + 1, min=0) * 16 + (m[3] + 3) // 4 + 4).double().mean().item()) for m in made]
        mean_alg = sum(alg) / len(alg)
    del made
    packed_in = None
    if a.packed:            # the same batches as the packed entry point takes them: packed on the host (bns_pack_reads), resident before the clock starts
        packed_in = []
        for bi in range(n_batches):
            ho_ = offsets_l[bi].cpu().numpy().astype(np.uint64)
            words, bw, bm = bonsai_amd.pack_reads(batches[bi][:int(ho_[-1])].cpu().numpy(), ho_, threads=effective_cores())
            dense = np.zeros(words.size, dtype=np.uint32)
            dense[bw.astype(np.int64)] = bm
            packed_in.append((torch.from_numpy(words.view(np.int64)).to(dev), torch.from_numpy(dense.view(np.int32)).to(dev) if bw.size else None))
            del words, dense
    if not (multi and rank == 0):
        del pool                                                     # (rank 0 of an N>1 run re-derives the other ranks' reads later)
        pool = None
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                                         # hand the generator's scratch back to the device

    layout = {"bucket": bonsai_amd.LAYOUT_BUCKET, "khash": bonsai_amd.LAYOUT_KHASH, "minbucket": bonsai_amd.LAYOUT_MINBUCKET}[a.layout]
    if a.bucket_slots_log2:
        ctx.set_bucket_slots_log2(a.bucket_slots_log2)
    if a.table_buckets:
        ctx.set_table_buckets(a.table_buckets)
    if a.min_span:
        ctx.set_minimizer_span(a.min_span)
    if a.identity:
        ctx.set_minimizer_identity(a.identity)
    # One table geometry for the whole job: the library sizes the clustered table from the key count and the free HBM it
    # finds, which can differ between ranks -- rank 0 loads first, the others take its bucket count, window and identity.
    geo_t = torch.zeros(4, dtype=torch.int64, device=dev)
    host_arrays = None
    t_load = time.time()
    if a.stream_load:
        if multi or layout != bonsai_amd.LAYOUT_MINBUCKET:
            sys.stderr.write("bench.py: --stream-load is a single-process, clustered-layout mode\n")
            return 2
        host_arrays = (flags.cpu().numpy().view(np.uint32), keys.cpu().numpy().view(np.uint64), vals.cpu().numpy().view(np.uint32))
        del flags, keys, vals
        flags = keys = vals = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        ctx.debug_set(a.ablate | 0x1000)             # BNS_DBG_STREAM_LOAD: through the staging buffers whatever the free HBM says
        ctx.load_table(nb, host_arrays[0], host_arrays[1], host_arrays[2], layout=layout)
        ctx.debug_set(a.ablate)
    elif rank == 0:
        ctx.load_table_device(nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), layout, stream)
        if layout == bonsai_amd.LAYOUT_MINBUCKET:
            g0 = ctx.table_geometry()
            geo_t = torch.tensor([g0["buckets"], g0["span"], g0["identity_bits"], g0["group_fill"] + 1], dtype=torch.int64, device=dev)
    if multi:
        bcast(geo_t)
    if rank != 0:
        gb, gs, gi, gf = (int(x) for x in geo_t.tolist())
        if gb:
            ctx.set_table_buckets(gb)
            if gs:
                ctx.set_minimizer_span(gs)
            ctx.set_minimizer_identity(gi)
            ctx.set_table_fill(gf)                   # (rank 0's fill: arrival order or group by group -- every replica the same table)
        ctx.load_table_device(nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), layout, stream)
    torch.cuda.synchronize()
    t_load = time.time() - t_load
    info = ctx.table_info()
    tstats = ctx.table_stats()
    geo = ctx.table_geometry()
    if ctx.table_warning() and rank == 0:
        sys.stderr.write("bench.py: table: %s\n" % ctx.table_warning())

    n_units = n // 2 if a.paired else n
    # double-buffered results: the gather of step i (RCCL, its own stream) overlaps the classify of step i+1
    from bonsai_amd import shard
    sizes = shard.shard_sizes((a.total_reads // (2 if a.paired else 1)) if a.scaling == "strong" else n_units * sworld, sworld)
    pad = max(sizes) if multi else n_units                             # (strong scaling: shards may differ by one unit)
    taxons = [torch.zeros(pad, dtype=torch.int32, device=dev) for _ in range(2)]
    missing = torch.zeros(n_units, dtype=torch.int32, device=dev)
    ambig = torch.zeros(n_units, dtype=torch.int32, device=dev)
    gather_lists = [[torch.empty_like(taxons[0]) for _ in range(world)] if (multi and rank == 0) else None for _ in range(2)]
    gathered_host = [None, None]
    works = [None, None]
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    if a.save_db and a.save_reads and rank == 0 and not a.paired and a.len_dist == "fixed":
        S = min(a.save_reads, n)
        st_t = torch.zeros(S, dtype=torch.int32, device=dev)
        ctx.classify_device(batches[0].data_ptr(), offsets_l[0].data_ptr(), S, S * L, L, False, st_t.data_ptr(), None, None, None, None, stream)
        torch.cuda.synchronize()
        st_t.cpu().numpy().astype(np.uint32).tofile(os.path.join(a.save_db, "taxa.u32"))
        rl = 2 * L + 15                                        # "@r<8 digits>\n" + bases + "\n+\n" + quality + "\n"
        with open(os.path.join(a.save_db, "reads.fq"), "wb") as f:
            for r0 in range(0, S, 4_000_000):
                m = min(4_000_000, S - r0)
                rec = np.empty((m, rl), dtype=np.uint8)
                rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
                idx = (np.arange(m) + r0) % 100_000_000
                for d in range(8):
                    rec[:, 9 - d] = ord("0") + (idx // 10 ** d) % 10
                rec[:, 10] = 10
                rec[:, 11:11 + L] = batches[0][r0 * L:(r0 + m) * L].cpu().numpy().reshape(m, L)
                rec[:, 11 + L] = 10; rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
                rec[:, 14 + L:14 + 2 * L] = ord("I")
                rec[:, 14 + 2 * L] = 10
                rec.tofile(f)

    def step(i):
        j = i & 1
        bi = j % n_batches
        if works[j] is not None:                    # the buffer's previous gather must have drained
            works[j].wait()
            works[j] = None
        if packed_in is not None:
            pw, pm = packed_in[bi]
            ctx.classify_packed_device(pw.data_ptr(), pm.data_ptr() if pm is not None else None, offsets_l[bi].data_ptr(), n, totals[bi], L, a.paired,
                                       taxons[j].data_ptr(), missing.data_ptr(), ambig.data_ptr(), None, None, stream)
        else:
            ctx.classify_device(batches[bi].data_ptr(), offsets_l[bi].data_ptr(), n, totals[bi], L, a.paired, taxons[j].data_ptr(),
                                missing.data_ptr(), ambig.data_ptr(), None, None, stream)
        if multi:
            if backend == "nccl":
                works[j] = dist.gather(taxons[j], gather_lists[j], dst=0, async_op=True)
            else:                                   # gloo has no CUDA gather: stage through the host (debug path only)
                t = taxons[j].cpu()
                gl = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
                dist.gather(t, gl, dst=0)
                gathered_host[j] = gl

    def fence():
        for j in range(2):
            if works[j] is not None:
                works[j].wait()
                works[j] = None
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    ctx.set_timing(True)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    ksum_ms, kcount = ctx.timing_summary()
    ctx.set_timing(False)
    fetch_dbg = None
    if hasattr(ctx.L, "bns_debug_fetch_count"):     # measurement build only (-DBNS_COUNT_FETCHES, tools/measure.sh)
        import ctypes
        c2 = (ctypes.c_ulonglong * 8)()
        ctx.L.bns_debug_fetch_count.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        ctx.L.bns_debug_fetch_count(ctx.h, c2)
        fetch_dbg = {"buckets_fetched_per_launch": c2[0] / (a.steps + a.warmup), "probe_passes_per_launch": c2[1] / (a.steps + a.warmup),
                     "overflow_lookups_per_launch": c2[2] / (a.steps + a.warmup), "rounds_with_overflow_lookups_per_launch": c2[3] / (a.steps + a.warmup),
                     "quad_probe_iterations_per_launch": c2[4] / (a.steps + a.warmup)}
        if hasattr(ctx.L, "bns_debug_ovf_stats"):
            c3 = (ctypes.c_ulonglong * 3)()
            ctx.L.bns_debug_ovf_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            ctx.L.bns_debug_ovf_stats(ctx.h, c3)
            fetch_dbg["overflow_table"] = {"occupied_slots": c3[0], "full_buckets": c3[1], "slots": c3[2]}
            if os.environ.get("BNS_DUMP_OVF") and c3[2]:
                buf = np.zeros(int(c3[2]) * 2, dtype=np.uint64)
                ctx.L.bns_debug_ovf_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong]
                ctx.L.bns_debug_ovf_copy(ctx.h, buf.ctypes.data, buf.nbytes)
                np.save(os.environ["BNS_DUMP_OVF"], buf)
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    total_reads_step = n if a.emulate_rank >= 0 else (a.total_reads if a.scaling == "strong" else n * world)
    reads_per_s = (total_reads_step * a.steps) / dt
    # roofline of the dominant kernel (classify_kernel): SURVEY 8d algorithmic bytes
    kmers_per_read = max(0, L - comb + 1)
    alg_bytes_per_read = kmers_per_read * 16 + (L + 3) // 4 + 4          # 1962 B for L=150,k=31
    if mean_alg is not None:
        alg_bytes_per_read = mean_alg                                    # ragged batch: the mean over its reads
    kern_ms = ksum_ms / max(1, kcount)
    achieved_gbs = (alg_bytes_per_read * n) / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters: they cannot be collected from inside this process (rocprofv3 wraps the
    # command), so the figure is the one tools/measure.sh took for this very workload shape AND this very library source
    # (sha256 over the kernel / API sources, written into profiles/traffic.json by tools/summarize_prof.py) -- null, with a
    # warning, for any other shape or after any change to the kernels.
    traffic = None
    traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            same_shape = (tj.get("reads_per_launch") == n and tj.get("layout") == a.layout and tj.get("read_len", 150) == L
                          and not a.paired and not a.spacing and not a.packed and a.len_dist == "fixed" and tj.get("k", 31) == k
                          and tj.get("db_window", 0) == (a.db_window if a.db_window > k else k) and tj.get("genome_len", 1 << 18) == G
                          and tj.get("genomes", 1024) == NG and tj.get("genome_model", "uniform") == a.genome_model
                          and tj.get("table_buckets") == int(geo["buckets"]) and tj.get("identity_bits") == int(geo["identity_bits"]))
            if same_shape and tj.get("source_sha256") == lib_source_sha256():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = tj.get("source")
            elif same_shape and rank == 0:
                sys.stderr.write("bench.py: profiles/traffic.json was measured on other kernel sources (sha256 %s..., now %s...): "
                                 "roofline.traffic = null until tools/measure.sh is re-run\n" % (str(tj.get("source_sha256"))[:12], lib_source_sha256()[:12]))
        except Exception:
            traffic = None

    khash_bytes = nb * 12 + max(1, nb >> 4) * 4
    n_home = int(geo["buckets"]) if a.layout == "minbucket" else 0
    load_factor = (float(tstats["n_keys"]) / (n_home * 10)) if n_home else \
        (float(tstats["n_keys"]) / max(1, int(tstats["main_bytes"] // 16)) if a.layout == "bucket" else float(hdr[2]) / nb)
    which = "configs[2]" if a.spacing else ("configs[1]" if (k == 31 and a.db_window == 50 and a.db_score == "entropy" and NG == 1024
                                                              and a.len_dist == "fixed" and L == 150 and not a.paired and a.genome_model == "uniform")
                                            else "variant of configs[1]")
    if a.scaling == "strong":
        which = "configs[3] shape (strong scaling)"
    out = {
        "metric": "reads/s classified (150 bp)", "value": reads_per_s, "unit": "reads/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": a.scaling, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: k=%d canonical, db = %s of %d synthetic genomes (%s) x %d bases (%d keys, 2^%d khash "
                               "buckets, %s layout %.1f GB) in HBM, %d synthetic %d bp reads per GPU per step%s%s"
                               % (which, k, ("w=%d %s minimizers" % (a.db_window, a.db_score)) if a.db_window > k else "every k-mer",
                                  NG, a.genome_model, G, info["n_keys"] or int(hdr[2]), a.log2_buckets, a.layout,
                                  info["device_bytes"] / 1e9, n, L, ", paired" if a.paired else "",
                                  (", spaced seed " + a.spacing) if a.spacing else ""),
                   "reads_per_gpu": n, "total_reads_per_step": total_reads_step, "input": ("packed 2-bit words" if a.packed else "ASCII"), "read_len": L, "len_dist": a.len_dist,
                   "mean_read_len": (sum(totals) / float(len(totals)) / n), "k": k, "layout": a.layout, "paired": bool(a.paired),
                   "pair_model": ("two ends of one fragment, insert size uniform in [L, 3L]" if a.paired else None),
                   "table_overflow_keys": int(tstats["n_overflow_keys"]),
                   "db_window": a.db_window if a.db_window > k else k,
                   "db_score": (a.db_score if a.db_window > k else "none (every k-mer)"),
                   "genomes": NG, "genome_len": G, "genome_model": genome_info,
                   "db_keys": int(info["n_keys"] or int(hdr[2])),
                   "table_buckets": n_home or None, "table_bytes": int(info["device_bytes"]), "khash_bytes": int(khash_bytes),
                   "table_bytes_over_khash_bytes": float(info["device_bytes"]) / khash_bytes,
                   "load_factor": load_factor, "spacing": a.spacing or None,
                   "table_minimizer_m": (int(geo["m"]) if a.layout == "minbucket" else None),
                   "table_identity_bits": (int(geo["identity_bits"]) if a.layout == "minbucket" else None),
                   "table_spilled_keys": (int(geo["spilled_keys"]) if a.layout == "minbucket" else None),
                   "lib_source_sha256": lib_source_sha256()[:16],
                   "parallelism": "reads sharded x%d, db replicated (RCCL broadcast), taxids gathered" % world},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "classify_kernel",
                     "kernel_ms": kern_ms, "launches_timed": kcount, "alg_bytes_per_read": alg_bytes_per_read},
        "setup_s": t_setup, "table_load_s": t_load,
    }
    if a.emulate_rank >= 0:
        out["emulated"] = {"rank": srank, "world": sworld, "scaling": a.scaling,
                           "job_total_reads_per_step": a.total_reads if a.scaling == "strong" else n * sworld,
                           "note": "one GPU classifying rank %d's shard of a %d-rank job (same seeds, same shard bounds, no process "
                                   "group): value = THIS rank's reads/s; the N-GPU job is this times N minus the gather" % (srank, sworld)}
        out["config"]["parallelism"] = "emulated rank %d of %d (reads sharded, db replicated)" % (srank, sworld)
    if a.dry_run_world > 0:
        out["dry_run"] = ("%d ranks on ONE device over %s, db and reads cut down: a walk through the %d-GPU job's launch, shards, broadcast and gather -- "
                          "its reads/s says nothing about %d GPUs" % (a.dry_run_world, os.environ.get("BNS_BENCH_BACKEND", "nccl"), a.dry_run_world, a.dry_run_world))
    if a.stream_load:
        out["config"]["table_load"] = "khash arrays streamed from host memory (bns_load_table), %.1f s" % t_load
    if fetch_dbg:
        out["debug_fetch_count"] = fetch_dbg

    oracle = None
    if rank == 0 and not a.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        hf, hk, hv = host_arrays if host_arrays is not None else (flags.cpu().numpy().view(np.uint32), keys.cpu().numpy().view(np.uint64),
                                                                   vals.cpu().numpy().view(np.uint32))
        oracle = (O, O.Table.wrap(int(hdr[0]), int(hdr[2]), int(hdr[1]), int(hdr[3]), hf, hk, hv),
                  O.Taxonomy(pairs=[(int(c), int(p)) for c, p in enumerate(parent) if p != 0xFFFFFFFF and c != 0]))

    # ---- N>1 (or the forced one-rank group): every rank's kernel time, and per rank a parity sample taken from the GATHERED
    # result -- rank 0 re-derives the head of rank r's last batch from its seed, classifies it on its own GPU and (unless
    # --no-cpu) with the oracle, and compares both with what the gather delivered for rank r
    if multi:
        mine = {"rank": rank, "kernel_ms": kern_ms, "launches_timed": kcount, "reads": n,
                "achieved": achieved_gbs, "frac": achieved_gbs / HBM_PEAK_GBS, "placement": placement,
                "shard": {"first_unit": int(sum(sizes[:srank])), "units": int(sizes[srank])}}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            last_j = (a.steps - 1) & 1
            last_b = last_j % n_batches
            inc = 2 if a.paired else 1
            for r in range(world):
                S = min(a.rank_sample, sizes[r] * inc)
                S -= S % 2
                if S <= 0:
                    continue
                rb, ro, rtot, _ = make_batch(pool, sizes[r] * inc, L, NG, G, dev, r, last_b, a.paired, lens_pool, head=S)
                rb = torch.cat([rb, torch.zeros(8, dtype=torch.uint8, device=dev)])
                su = S // inc
                lt = torch.zeros(su, dtype=torch.int32, device=dev)
                ctx.classify_device(rb.data_ptr(), ro.data_ptr(), S, rtot, L, a.paired, lt.data_ptr(), None, None, None, None, stream)
                torch.cuda.synchronize()
                gsl = (gather_lists[last_j][r] if backend == "nccl" else gathered_host[last_j][r].to(dev))[:su]
                allr[r]["parity_sample"] = {"reads": S, "gathered_vs_local_gpu_mismatches": int((gsl != lt).sum().item()),
                                            "classified_frac": float((gsl != 0).float().mean().item())}
                if oracle is not None:
                    mism, _, _ = oracle_check(oracle[0], oracle[1], oracle[2], k, gaps, a.paired, rb, ro, S, gsl, None, None, effective_cores())
                    allr[r]["parity_sample"]["gathered_vs_oracle_mismatches"] = mism
            out["per_rank"] = allr
            # what travelled: the db broadcast (three arrays + the read pool + the table geometry) and the per-step gather
            out["collectives"] = {"backend": placement["backend"], "broadcast_bytes": {"flags": int(flags.numel() * 4), "keys": int(keys.numel() * 8), "vals": int(vals.numel() * 4),
                                                                                     "read_pool": int(pool.numel())},
                                  "gather_bytes_per_step": int(sum(sizes) * 4), "gather_buffers": 2,
                                  "shard_units": [int(x) for x in sizes]}
            bad = [r for r in allr if r.get("parity_sample") and (r["parity_sample"]["gathered_vs_local_gpu_mismatches"]
                                                                  or r["parity_sample"].get("gathered_vs_oracle_mismatches"))]
            if bad:
                out["error"] = "gathered results of rank(s) %s disagree with their re-derived reads" % [r["rank"] for r in bad]
            out["roofline"]["frac_min_over_ranks"] = min(r["frac"] for r in allr)
            out["roofline"]["kernel_ms_max_over_ranks"] = max(r["kernel_ms"] for r in allr)

    # ---- standalone probe kernel (the metric's "% HBM roofline on probe"), rank 0 at N=1: keys WITHOUT read locality
    # (half present, half random 62-bit misses), so every lookup fetches its own 128-byte bucket
    if rank == 0 and world == 1 and not a.no_probe and a.layout == "minbucket" and not a.spacing and not a.stream_load:
        out["probe_roofline"] = probe_leg(ctx, a, dev, stream, flags, keys, nb, float(hdr[2]) / nb)

    # ---- the text path at the C ABI (rank 0, N=1, the default single-end fixed-length shape)
    if rank == 0 and world == 1 and not a.no_text and not a.paired and not a.spacing and a.len_dist == "fixed" and not a.packed and a.emulate_rank < 0:
        try:
            torch.cuda.synchronize()
            lj_t = (a.steps - 1) & 1
            tl = text_leg(ctx, a, batches[((a.steps - 1) & 1) % n_batches], offsets_l[((a.steps - 1) & 1) % n_batches], taxons[lj_t])
            if tl:
                out["text_path"] = tl
                if tl.get("mismatches_vs_timed_launch"):
                    out["error"] = "text path and the timed launch disagree"
        except Exception as e:
            out["text_path"] = {"error": str(e)[:200]}

    # ---- BGZF members inflated on the device (rank 0, N=1)
    if rank == 0 and world == 1 and not a.no_inflate and a.emulate_rank < 0:
        try:
            torch.cuda.synchronize()
            il = inflate_leg(ctx.L, local)
            if il:
                out["inflate_path"] = il
                if il.get("members_wrong"):
                    out["error"] = "inflated members differ from zlib's"
        except Exception as e:
            out["inflate_path"] = {"error": str(e)[:200]}
        # ---- ONE gzip stream inflated on the device (round 6)
        try:
            torch.cuda.synchronize()
            gl = gz_stream_leg(ctx.L, local)
            if gl:
                out["gz_stream_path"] = gl
                if gl.get("text_wrong"):
                    out["error"] = "the inflated gzip stream differs from its text"
        except Exception as e:
            out["gz_stream_path"] = {"error": str(e)[:200]}

    # ---- parity sample + CPU baseline (rank 0, N=1 only): the oracle is the checker / the reported baseline
    if rank == 0 and world == 1 and oracle is not None:
        S = min(a.cpu_sample, n)
        S -= S % 2
        last = ((a.steps - 1) & 1) % n_batches
        lj = (a.steps - 1) & 1
        ncores = effective_cores()
        mism, best, taxa = oracle_check(oracle[0], oracle[1], oracle[2], k, gaps, a.paired, batches[last], offsets_l[last], S,
                                        taxons[lj], missing, ambig, ncores, repeat=3)
        out["cpu_baseline"] = {"value": S / best, "unit": "reads/s", "cores": ncores, "threads": ncores, "kind": "port",
                               "cpu_model": cpu_model(), "nproc": os.cpu_count(),
                               "sample": "first %d reads of the timed batch, same db (khash arrays as built), "
                                         "oracle/bns_oracle.c bo_classify_batch with OpenMP on %d threads (the cores the "
                                         "cgroup quota grants), best of 3" % (S, ncores)}
        # per-phase split of the port on this host and its calibration against the reference's own functions
        # (tools/cpu_calibrate.py, build container: profiles/r04_cpu_calibration.json)
        if not a.paired and not a.spacing:
            S2 = min(S, 1_000_000)
            ho2 = offsets_l[last][:S2 + 1].cpu().numpy().astype(np.uint64)
            ph = oracle[0].classify_batch_phase_seconds(oracle[1], oracle[2], k, batches[last][:int(ho2[-1])].cpu().numpy(), ho2, ncores)
            out["cpu_baseline"]["phase_split"] = {"reads": S2, "encode": ph[0] / ph[2], "probe": (ph[1] - ph[0]) / ph[2],
                                                  "vote_resolve": (ph[2] - ph[1]) / ph[2],
                                                  "note": "port's batch loop cut after encode / + kh_get / whole, best of 3 each"}
        try:
            cj = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_calibration.json")))
            out["cpu_baseline"]["calibration"] = {
                "ref_reads_per_s_per_thread": cj["ref_reads_per_s_per_thread"], "port_over_ref": cj["port_over_ref_nt"],
                "port_over_ref_1_thread": cj["port_over_ref_1t"], "threads": cj["threads_n"], "cpu_model": cj["cpu_model"],
                "source": "profiles/r04_cpu_calibration.json (tools/cpu_calibrate.py, build container: the reference's encoder loop, "
                          "kh_get, linear::counter and resolve_tree compiled with its own flags vs the port, same %d-byte table and reads)"
                          % cj["khash_bytes"]}
        except Exception:
            pass
        # ---- the REFERENCE's own code on this host, when its compiled form travelled with the repo (oracle/_ref/libbns_ref.so: the
        # reference's DNA4 / rhmask / canonical_representation / kh_get / linear::counter / resolve_tree compiled in the build container,
        # driven by the loop of encoder.h:246-271 under OpenMP -- oracle/ref_harness.cpp ref_classify_batch).  Same sample, same khash
        # arrays, same threads; its taxa are compared with the GPU's.  Then `value` is the reference's and kind says so.
        ref_so = os.path.join(ROOT, "oracle", "_ref", "libbns_ref.so")
        if os.path.exists(ref_so) and not a.paired and not a.spacing and not a.no_ref:
            try:
                import ctypes as C
                R = C.CDLL(ref_so)
                u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
                R.ref_khc_view.restype = C.c_void_p
                R.ref_khc_view.argtypes = [C.c_uint64] * 4 + [u32p, u64p, u32p]
                R.ref_khp_from_pairs.restype = C.c_void_p; R.ref_khp_from_pairs.argtypes = [u32p, u32p, C.c_uint32]
                R.ref_classify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_char_p, u64p, C.c_uint64, u32p, C.c_int, C.c_int, u64p]
                hf, hk, hv = host_arrays if host_arrays is not None else (flags.cpu().numpy().view(np.uint32), keys.cpu().numpy().view(np.uint64),
                                                                           vals.cpu().numpy().view(np.uint32))
                hf, hk, hv = np.ascontiguousarray(hf), np.ascontiguousarray(hk), np.ascontiguousarray(hv)
                dbv = R.ref_khc_view(int(hdr[0]), int(hdr[2]), int(hdr[1]), int(hdr[3]), hf.ctypes.data_as(u32p), hk.ctypes.data_as(u64p), hv.ctypes.data_as(u32p))
                ch = np.nonzero(np.asarray(parent) != 0xFFFFFFFF)[0].astype(np.uint32)      # every key of the parent map (1 -> 0 included)
                pa = np.ascontiguousarray(np.asarray(parent)[ch].astype(np.uint32))
                txv = R.ref_khp_from_pairs(ch.ctypes.data_as(u32p), pa.ctypes.data_as(u32p), int(ch.size))
                ho = offsets_l[last][:S + 1].cpu().numpy().astype(np.uint64)
                hb = np.ascontiguousarray(batches[last][:int(ho[-1])].cpu().numpy())
                out4 = np.zeros(4 * S, dtype=np.uint32)
                sink = C.c_uint64()
                tb = None
                for _ in range(3):
                    t0 = time.perf_counter()
                    R.ref_classify_batch(dbv, txv, k, 1, hb.ctypes.data_as(C.c_char_p), ho.ctypes.data_as(u64p), S, out4.ctypes.data_as(u32p), ncores, 2, C.byref(sink))
                    e = time.perf_counter() - t0
                    tb = e if tb is None or e < tb else tb
                rt = out4.reshape(S, 4)[:, 0]
                gpu_t = taxons[lj][:S].cpu().numpy().astype(np.uint32)
                ref_mism = int((rt != gpu_t).sum())
                port = dict(out["cpu_baseline"])
                out["cpu_baseline"].update({"value": S / tb, "kind": "reference", "port": {"value": port["value"], "sample": port["sample"]},
                                            "gpu_vs_reference_mismatches": ref_mism,
                                            "sample": "first %d reads of the timed batch, same khash arrays: the reference's own functions compiled in the build "
                                                      "container (oracle/_ref/libbns_ref.so: DNA4 / rhmask / canonical_representation / kh_get / linear::counter / "
                                                      "resolve_tree in the loop of encoder.h:246-271, OpenMP on %d threads), best of 3" % (S, ncores)})
                if ref_mism:
                    out["error"] = "GPU and the reference's own functions disagree on the sample"
            except Exception as e:  # the checker's reference leg is optional: the port's numbers stand
                out["cpu_baseline"]["reference_leg_error"] = str(e)[:200]
        out["parity_sample"] = {"reads": S, "mismatches": mism, "classified_frac": float((taxa != 0).mean())}
        if mism:
            out["error"] = "GPU and oracle disagree on the sample"
    if rank == 0:
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
