import sys, json
tag = sys.argv[1]
d = json.loads(sys.stdin.read())
print("%s step_ms %.3f kernel_ms %.3f ovf %s" % (tag, d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"].get("table_overflow_keys")))
