"""Per-kernel time per step from a rocprofv3 --kernel-trace CSV (the --stats CSV of ROCm 7.2 mis-attributes names).
usage: python profiles/trace_summary.py <dir or *_kernel_trace.csv> <steps incl. warmup> [top] [split_us]
split_us: launches of at least that many microseconds are listed as a separate row "<kernel> [>= split_us]" (a kernel that
runs both on the large per-frame tensors and on the small per-slot ones shows its two populations)."""
import collections, csv, glob, os, re, sys


def main(path, steps, top=40, split_us=None):
    f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n).split("(")[0][:84]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if split_us is not None and us >= split_us:
            n += " [>= %g us]" % split_us
        agg[n][0] += 1
        agg[n][1] += us
    tot = sum(v[1] for v in agg.values())
    print("total kernel time %.2f ms/step over %d steps" % (tot / 1e3 / steps, steps))
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%7.3f ms/step %5.1f%% %7.1f calls/step avg %8.1f us  %s" % (v[1] / 1e3 / steps, 100 * v[1] / tot, v[0] / steps,
                                                                      v[1] / v[0], n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 40,
         float(sys.argv[4]) if len(sys.argv) > 4 else None)
