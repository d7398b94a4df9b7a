#!/bin/bash
# Parse-only throughput of the compressed-input readers on the box's host cores, by inflater thread count (no GPU work):
#   BGZF members side by side (libdeflate / zlib) and one gzip stream on many threads (pgzip).   gpurun -- bash tools/gz_scaling.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
g++ -O2 -std=c++17 -Ibonsai_amd/csrc/host -Iinclude tools/micro/host_parse_bench.cpp bonsai_amd/csrc/host/bns_host.o bonsai_amd/csrc/host/pgzip.o -o /tmp/host_parse_bench \
    -Lbonsai_amd/lib -lbonsai_amd -lz -ldl -lpthread -Wl,-rpath,$PWD/bonsai_amd/lib -Wl,-rpath,/opt/rocm/lib || exit 1
python - <<'PY'
import os, struct, zlib, time
import numpy as np
from multiprocessing import Pool
d = "/tmp/gzscale"; os.makedirs(d, exist_ok=True)
n = 16_000_000
rng = np.random.default_rng(1)
g = rng.integers(0, 4, 4_000_000).astype(np.uint8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
parts = []
for s0 in range(0, n, 2_000_000):
    m = 2_000_000
    st = rng.integers(0, g.size - 150, size=m)
    rec = np.empty((m, 314), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(s0, s0 + m)
    for j in range(8):
        rec[:, 9 - j] = ord("0") + (idx // 10 ** j) % 10
    rec[:, 1] = ord("r"); rec[:, 9] = 10
    rec[:, 10:160] = acgt[g[st[:, None] + np.arange(150)[None, :]]]
    rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
    rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
    rec[:, 313] = 10
    parts.append(rec.tobytes())
data = b"".join(parts); del parts
open(d + "/r.fq", "wb").write(data)
def member(chunk):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(chunk) + co.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
t = time.time()
with Pool(16) as p:
    ms = p.map(member, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=256)
open(d + "/r.bgzf.fq.gz", "wb").write(b"".join(ms) + member(b""))
co = zlib.compressobj(1, zlib.DEFLATED, 31)
open(d + "/r.l1.fq.gz", "wb").write(co.compress(data[:8_000_000 * 314]) + co.flush())
print("made in %.0f s" % (time.time() - t), flush=True)
PY
echo "plain FASTQ (16 M reads):  $(/tmp/host_parse_bench /tmp/gzscale/r.fq)"
for t in 1 2 4 8 12 16 24; do echo "BGZF, $t inflaters:  $(BNS_GZ_THREADS=$t /tmp/host_parse_bench /tmp/gzscale/r.bgzf.fq.gz)"; done
echo "BGZF, zlib, 12 inflaters:  $(BNS_NO_LIBDEFLATE=1 BNS_GZ_THREADS=12 /tmp/host_parse_bench /tmp/gzscale/r.bgzf.fq.gz)"
echo "gzip -1 stream (8 M reads), zlib reader:  $(BNS_NO_PGZ=1 /tmp/host_parse_bench /tmp/gzscale/r.l1.fq.gz)"
for t in 1 2 4 8 12 16 24; do echo "gzip -1 stream, $t threads:  $(BNS_GZ_THREADS=$t /tmp/host_parse_bench /tmp/gzscale/r.l1.fq.gz)"; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
