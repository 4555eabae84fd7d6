import os, csv, sys, glob
f = max(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 35
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    n = r['Name'].replace('(anonymous namespace)::', '')[:70]
    print(f"{n:70s} calls/f={float(r['Calls'])/frames:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} us/frame={float(r['TotalDurationNs'])/1e3/frames:8.1f} {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print('GPU us/frame', tot / 1e3 / frames)
