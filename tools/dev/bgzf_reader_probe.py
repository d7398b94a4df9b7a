"""one BGZF file through the host reader with the device inflating (debug aid for tests/test_inflate.py's reader test)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, synth
from bonsai_amd import hostio
rng = np.random.default_rng(5)
recs = []
for i in range(60000):
    L = int(rng.integers(30, 260))
    s_ = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L))
    recs.append(b"@q%d c%d\n" % (i, i % 7) + s_ + b"\n+\n" + bytes(rng.integers(33, 74, L).astype(np.uint8)) + b"\n")
doc = b"".join(recs)
synth.write_bgzf("/tmp/d_std.fq.gz", doc)
sys.stderr.write("file written\n")
hostio.lib().bnsh_set_bgzf_device(0)
r, _ = hostio.read_fastx("/tmp/d_std.fq.gz", chunk_size=int(os.environ.get("PROBE_CHUNK", 1 << 20)), block_bytes=int(os.environ.get("PROBE_BLOCK", 0)))
print(len(r), flush=True)
