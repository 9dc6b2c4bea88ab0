import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
