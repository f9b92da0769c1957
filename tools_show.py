import json, sys
d = json.load(open(sys.argv[1]))
print("headline: frame %.4f ms  kernel %.4f ms  %.0f Mrays/s  frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["frac"]))
for k, v in d.get("extra", {}).items():
    print("  %-22s %9.3f ms %8.0f Mrays/s  frac %.3f  launches %d chunks %d" % (k, v["frame_ms"], v["Mrays_per_s"], v["roofline_frac"] or float("nan"), v["launches"], v["chunks"]))
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
