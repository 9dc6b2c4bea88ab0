import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    n = r["Name"]
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    nm = m.group(1) if m else n[:40]
    print("%-34s calls %5s avg %8.1f us per-iter %7.1f us %5.1f%%" % (nm[:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/32, 100*float(r["TotalDurationNs"])/tot))
print("total per iteration (kernel time)", tot/1e3/32)
