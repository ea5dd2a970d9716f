import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms/step", round(d["ms_per_step"], 4), "timed-regime kernel sum", r["step_ms_sum_of_kernels"], "solo kernel sum", r.get("solo", {}).get("step_ms_sum_of_kernels"))
for leg, t in (("timed regime", r["families"]), ("solo", r.get("solo", {}).get("families", {}))):
    print(leg)
    for k, v in t.items():
        print(f"  {k:24s} n={v['launches_per_step']:5.1f} avg={v['avg_us']:8.2f} ms={v['ms_per_step']:.4f}")
