"""Summarise a rocprofv3 --kernel-trace --stats CSV pair: per-kernel time per step, and per-launch-shape
GEMM timings.  usage: python profiles/summarize.py <dir with *_kernel_stats.csv> <steps incl. warmup>"""
import collections
import csv
import glob
import re
import sys


def main(d, steps):
    stats = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
    trace = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time %.2f ms over %d steps -> %.2f ms/step" % (tot / 1e6, steps, tot / 1e6 / steps))
    for r in rows[:24]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"^void ", "", n).split("(")[0][:88]
        print("%8.2f ms/step %5.1f%% calls/step=%7.1f avg=%8.1fus  %s" % (
            float(r["TotalDurationNs"]) / 1e6 / steps, 100 * float(r["TotalDurationNs"]) / tot,
            int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n))
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        n = r["Kernel_Name"]
        if "gemm_nt_kernel" in n or "gemm_generic" in n or "traj_" in n:
            m = re.search(r"(gemm_nt_kernel|gemm_generic_kernel|traj_\w+)<([^>]*)>", n)
            key = (m.group(1) + "<" + m.group(2) + ">" if m else n[:40], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]),
                   r["Grid_Size_Y"], r["Grid_Size_Z"])
            g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("\nper launch shape (kernel, workgroups x, y, z):")
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:30]:
        print("  %-60s n/step=%6.1f avg=%8.1fus total/step=%7.2fms" % (str(k), len(v) / steps, sum(v) / len(v), sum(v) / 1e3 / steps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
