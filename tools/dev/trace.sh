python tools/cli_bench.py 64000000 -p 4 > /dev/null 2>&1
for P in 1 2; do
BNS_CLI_TRACE=/tmp/trace$P.tsv ./bonsai_amd/bin/bonsai classify -a -p 4 -P $P -o /tmp/clibench/out.txt /tmp/clibench/bns.db /tmp/clibench/nodes.dmp /tmp/clibench/r.fq 2>/dev/null
python - /tmp/trace$P.tsv <<'PY'
import sys, collections
ev=collections.defaultdict(dict)
for l in open(sys.argv[1]):
    st,seq,t0,t1=l.split(); ev[st][int(seq)]=(float(t0),float(t1))
n=max(ev['W'])+1
print("chunks",n,"end",ev['W'][n-1][1])
for st in "RPGFW":
    busy=sum(b-a for a,b in ev[st].values()); print(st,"busy %.3f"%busy, "first %.4f last %.4f"%(min(a for a,b in ev[st].values()), max(b for a,b in ev[st].values())))
# per-chunk timeline for a few chunks mid-run
for q in list(range(0,6))+list(range(200,206)):
    print(q, " ".join("%s[%.4f-%.4f]"%(st,*ev[st][q]) for st in "RPGFW" if q in ev[st]))
# idle gaps of formatter: time between W end of q and F begin of q+1
gaps=[ev['F'][q+1][0]-ev['W'][q][1] for q in range(n-1)]
print("formatter idle total %.3f"%sum(gaps), "max %.4f"%max(gaps))
gapsG=[ev['G'][q+1][0]-ev['G'][q][1] for q in range(n-1)]
print("caller idle total %.3f"%sum(gapsG))
gapsP=[ev['P'][q+1][0]-ev['P'][q][1] for q in range(n-1)]
print("packer idle total %.3f"%sum(gapsP))
PY
done
