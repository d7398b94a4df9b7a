#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c8; mkdir -p "$O"
python - > "$O/chunk.txt" 2>&1 <<'PY'
import os, subprocess, sys, time
import numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/tests")
import oracle_lib as O, synth
BIN=ROOT+"/bonsai_amd/bin/bonsai"
d="/tmp/ib2"; os.makedirs(d,exist_ok=True)
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
for lg in (27,28,29,30):
    subprocess.run([BIN,"pack","-p","8","-c",str(1<<lg),"-o",d+"/r.bnsp",d+"/r.fq"],stderr=subprocess.DEVNULL)
    body=open(d+"/r.bnsp","rb").read()
    with open(d+"/r8.bnsp","wb") as f:
        f.write(body[:32])
        for _ in range(8): f.write(body[32:])
    del body
    for rep in range(2):
        for extra in ([],["-b",d+"/t.bin"]):
            t=time.time()
            p=subprocess.run([BIN,"classify","-K","-p","4"]+extra+[d+"/bns.db",d+"/nodes.dmp",d+"/r8.bnsp"],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,BNS_CLI_TIMING="1"))
            dt=time.time()-t
            tl=[l[9:] for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "pack +" in l)]
            print("chunk 2^%d %s: %.2f s wall = %.0f M reads/s | %s"%(lg," ".join(extra[:1]),dt,8*n/dt/1e6," | ".join(tl)),flush=True)
PY
cat "$O/chunk.txt" | cut -c1-300
