"""one-line digest of a bench.py JSON line: python tools/_line.py file.json"""
import json, sys
for path in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception as e:
        print(path, "unreadable:", e); continue
    c, r = d["config"], d["roofline"]
    ps = d.get("parity_sample") or {}
    print("%s: %.3f Greads/s kernel %.3f ms frac %.3f | keys %.3g buckets %s load %.3f m %s id %s spilled %s ovf %s tbl/khash %.2f | parity %s/%s%s%s" % (
        path.split("/")[-1], d["value"] / 1e9, r["kernel_ms"], r["frac"], c["db_keys"], c.get("table_buckets"), c["load_factor"], c.get("table_minimizer_m"),
        c.get("table_identity_bits"), c.get("table_spilled_keys"), c.get("table_overflow_keys"), c.get("table_bytes_over_khash_bytes", 0),
        ps.get("mismatches"), ps.get("reads"), (" cpu %.2f M/s" % (d["cpu_baseline"]["value"] / 1e6)) if d.get("cpu_baseline") else "",
        (" ERROR " + d["error"]) if d.get("error") else ""))
