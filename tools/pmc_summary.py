"""Per-kernel sums of rocprofv3 --pmc counters.  usage: python tools/pmc_summary.py <dir with *_counter_collection.csv>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n).split("(")[0][:70]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(n, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("%-72s" % "kernel" + "".join("%22s" % c for c in names))
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
    print("%-72s" % n + "".join("%22.4g" % v.get(c, 0) for c in names))
