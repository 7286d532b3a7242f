"""Kernel summary of the LAST step only of a rocprofv3 --kernel-trace CSV (the whole-trace average mixes in model
construction, optimizer-state creation and warm-up effects).  The step starts at the last launch of <marker> (default
im2col_kernel: the first kernel of the Motionformer forward).
usage: trace_last_step.py <dir> [marker] [top] [neighbours-of substring]"""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:84]


d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "im2col_kernel"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
nb = sys.argv[4] if len(sys.argv) > 4 else None
f = d if d.endswith(".csv") else glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(starts) >= 2, "marker not found twice"
lo, hi = starts[-2], starts[-1]          # the last COMPLETE step: between the last two markers
step = rows[lo:hi]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in step:
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
wall = (step[-1][1] - step[0][0]) / 1e3
print("last complete step: %d launches, kernel time %.3f ms, wall %.3f ms (first start to last end), idle %.3f ms"
      % (len(step), tot / 1e3, wall / 1e3, (wall - tot) / 1e3))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.3f ms %5.1f%% %6d calls avg %8.1f us  %s" % (v[1] / 1e3, 100 * v[1] / tot, v[0], v[1] / v[0], n))
if nb:
    ctx = collections.Counter()
    names = [r[2] for r in step]
    for i, n in enumerate(names):
        if nb in n:
            ctx[(names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-")] += 1
    print("\nneighbours of %r in the last step:" % nb)
    for (a, b), c in ctx.most_common(20):
        print("%5d  after %-56s before %s" % (c, a[:56], b[:56]))
