#!/usr/bin/env python
"""gpurun_out/<dir> of tools/secondary.sh -> profiles/<tag>_configs.jsonl ({"name", "args", "bench"} per line),
profiles/<tag>_host_path.jsonl, profiles/<tag>_cli.txt, profiles/<tag>_big8e9.txt.  usage: summarize_secondary.py gpurun_out/r03c r03"""
import glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
rows = []
for f in sorted(glob.glob(os.path.join(src, "*.json")), key=os.path.getmtime):
    name = os.path.basename(f)[:-5]
    if not os.path.exists(f[:-5] + ".args"):               # (left over from an earlier call's merge)
        continue
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        bench = json.loads(line)
    except Exception as e:                                    # a run that failed keeps its place, with the reason
        bench = {"error": str(e), "stderr_tail": open(f[:-5] + ".err").read()[-400:] if os.path.exists(f[:-5] + ".err") else ""}
    args = open(f[:-5] + ".args").read().strip() if os.path.exists(f[:-5] + ".args") else ""
    rows.append({"name": name, "args": args, "bench": bench})
with open(os.path.join(out, tag + "_configs.jsonl"), "w") as o:
    for r in rows:
        o.write(json.dumps(r) + "\n")
for a, b in (("host_path.jsonl", "_host_path.jsonl"), ("cli.txt", "_cli.txt"), ("big8e9.log", "_big8e9.txt")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        txt = "".join(l for l in open(p) if "amdgpu.ids" not in l)
        open(os.path.join(out, tag + b), "w").write(txt)
for r in rows:
    b = r["bench"]
    if "value" in b:
        print("%-14s %8.1f M %s/s  %.2f ms  frac %.3f" % (r["name"], b["value"] / 1e6, "reads", b["ms_per_step"], b["roofline"]["frac"]))
    else:
        print(r["name"], "FAILED", b.get("error"))
