import json
d=json.loads(open("gpurun_out/r3k_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
s=d["steve"]; print(s.get("ms_per_step"), s.get("eager_ms_per_step"), s.get("product_loop_ms_per_step"), s.get("failed"), s["roofline"]["frac"])
m=s.get("model_step"); print(m.get("ms_per_step"), m.get("forward_ms"), m.get("failed"))
h=d["hr"]; print(h.get("ms_per_step"), h.get("clips_per_s"), h["large_batch"]["clips_per_s"], h["bf16_weights"]["clips_per_s"])
