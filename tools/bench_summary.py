"""One-screen summary of a bench.py JSON line.   python tools/bench_summary.py [file=profiles/r03_bench.json]"""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_bench.json"
d = json.loads(open(path).read().strip().splitlines()[-1])
print("orvit  %.1f clips/s  %.2f ms  frac %.3f  traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
s = d.get("steve", {})
print("steve  graph %s ms  eager %s ms  product loop %s ms  %s" % (s.get("ms_per_step"), s.get("eager_ms_per_step"),
                                                                 s.get("product_loop_ms_per_step"), s.get("failed") or ""))
m = s.get("model_step", {})
print("steve model step  %s ms  forward %s ms  %s" % (m.get("ms_per_step"), m.get("forward_ms"), m.get("failed") or ""))
h = d.get("hr", {})
print("hr  fp8w %s clips/s (%s ms)  batch16 %s  bf16w %s" % (h.get("clips_per_s"), h.get("ms_per_step"),
                                                             h.get("large_batch", {}).get("clips_per_s"), h.get("bf16_weights", {}).get("clips_per_s")))
