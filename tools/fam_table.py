import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", round(d["ms_per_step"], 4), "solo sum", r["step_ms_sum_of_kernels"], "in-situ sum", r.get("in_situ", {}).get("step_ms_sum_of_kernels"))
for leg, t in (("solo", r["families"]), ("in_situ", r.get("in_situ", {}).get("families", {}))):
    print(leg)
    for k, v in t.items():
        print(f"  {k:24s} n={v['launches_per_step']:5.1f} avg={v['avg_us']:8.2f} ms={v['ms_per_step']:.4f}")
