"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel.
usage: pmc_summary.py COUNTER_NAME PROFILE_DIR OUT_CSV"""
import collections
import csv
import glob
import sys

tag, d, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != tag:
            continue
        k = r["Kernel_Name"].replace("thip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
with open(out, "w") as f:
    f.write("kernel,dispatches,mean_%s_KB\n" % tag)
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        f.write("%s,%d,%.1f\n" % (k, n, v / n))
print(open(out).read())
