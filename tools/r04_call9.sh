#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c9; mkdir -p "$O"
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_scale.py::test_refseq_scale_streamed > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.log"
LIBS="libbonsai_amd_v2.so libbonsai_amd.so"
shape() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib timeout 900 python bench.py --no-probe --steps 5 --warmup 1 --cpu-sample 200000 "$@" 2>/dev/null | tail -1 > "$O/$name.$lib.$rep.json"; echo -n "$name:$lib "; python tools/_line.py "$O/$name.$lib.$rep.json"
  done; done; }
shape allk34 --genome-len 262144 --db-window 0 --table-buckets 67000000
shape allk34p --genome-len 262144 --db-window 0 --table-buckets 67000000 --paired
shape big8e9 --genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load
BIG="--genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load --no-probe --no-cpu --steps 3 --warmup 1"
BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 900 python bench.py $BIG > "$O/count_8e9.json" 2> "$O/count_8e9.err"; grep -o '"debug_fetch_count.*' "$O/count_8e9.json" | cut -c1-400
# container path: one vs two contexts on the device (is the caller thread or the link the limit?)
python - <<'PY' 2>&1 | cut -c1-250
import os, subprocess, sys, time
import numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/tests")
import oracle_lib as O, synth
BIN=ROOT+"/bonsai_amd/bin/bonsai"
d="/tmp/ib3"; os.makedirs(d,exist_ok=True)
w=synth.make_world(O,seed=3,k=31,genome_len=50000); O.db_write(d+"/bns.db",31,31,None,w.table); synth.write_nodes_dmp(d+"/nodes.dmp")
g=np.concatenate(list(w.genomes.values())); rng=np.random.default_rng(1); n=32_000_000
with open(d+"/r.fq","wb") as f:
    for s0 in range(0,n,2_000_000):
        m=2_000_000; st=rng.integers(0,g.size-150,size=m); rec=np.empty((m,314),dtype=np.uint8)
        rec[:,0]=ord("@"); idx=np.arange(s0,s0+m)
        for j in range(8): rec[:,9-j]=ord("0")+(idx//10**j)%10
        rec[:,1]=ord("r"); rec[:,9]=10; rec[:,10:160]=g[st[:,None]+np.arange(150)[None,:]]
        rec[:,160]=10; rec[:,161]=ord("+"); rec[:,162]=10; rec[:,163:313]=73; rec[:,313]=10
        rec.tofile(f)
subprocess.run([BIN,"pack","-p","8","-o",d+"/r.bnsp",d+"/r.fq"],stderr=subprocess.DEVNULL)
body=open(d+"/r.bnsp","rb").read()
with open(d+"/r8.bnsp","wb") as f:
    f.write(body[:32])
    for _ in range(8): f.write(body[32:])
del body
for gsel in ("0","0,0","0,0,0"):
    for rep in range(2):
        t=time.time()
        p=subprocess.run([BIN,"classify","-K","-p","4","-g",gsel,d+"/bns.db",d+"/nodes.dmp",d+"/r8.bnsp"],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,BNS_CLI_TIMING="1"))
        dt=time.time()-t
        tl=[l[9:] for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "pack +" in l)]
        print("-g %s: %.2f s wall = %.0f M reads/s | %s"%(gsel,dt,8*n/dt/1e6," | ".join(tl)),flush=True)
PY
