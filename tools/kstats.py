import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) < 0.05: continue
    print(f"{r['Name'][:56]:56s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f} pct {float(r['Percentage']):5.1f}")
