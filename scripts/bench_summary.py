import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d=json.loads(l); print("x-vec/s %.0f  ms/step %.3f" % (d["value"], d["ms_per_step"]))
r=d.get("roofline")
if r:
    print({k: round(v, 3) for k, v in r["per_class_ms_per_step"].items()})
    print("dominant", r["kernel"], "frac %.3f" % r["frac"], "launch_us %.1f" % r["launch_us"], r["bound"])
if "cpu_baseline" in d: print(d["cpu_baseline"])
