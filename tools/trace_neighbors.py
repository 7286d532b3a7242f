"""What runs next to a kernel?  From a rocprofv3 --kernel-trace CSV: for every launch whose name contains <substr>, the
names of the launches before and after it (by start time), counted.  usage: trace_neighbors.py <dir> <substr> [top]"""
import collections, csv, glob, os, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:70]


d, sub = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = d if d.endswith(".csv") else glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))))
names = [n for _, n in rows]
ctx = collections.Counter()
for i, n in enumerate(names):
    if sub in n:
        ctx[(names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-")] += 1
print("%d launches matching %r" % (sum(ctx.values()), sub))
for (a, b), c in ctx.most_common(top):
    print("%5d  after %-60s before %s" % (c, a, b))
