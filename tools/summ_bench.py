"""Print the headline numbers of bench.py JSON lines (files given on the command line)."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    k = d["steps"]
    print(f, "%.0f steps/s  %.2f ms/step" % (d["value"], d["ms_per_step"]),
          {n: round(v["seconds"] / k * 1e3, 2) for n, v in d["kernel_seconds"].items() if v["launches"]},
          "roof %.3f" % d["roofline"]["frac"])
