import os, csv, collections, sys, glob
f = max(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:58]
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[n].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
filt = sys.argv[2] if len(sys.argv) > 2 else 'conv_mfma'
for n, d in agg.items():
    if filt not in n: continue
    m = {k: sum(v) / len(v) for k, v in d.items()}
    print(n, ' dur_us=%.1f' % (sum(dur[n]) / len(dur[n]) / 1e3), 'n=%d' % len(dur[n]))
    print('    ', {k: int(v) for k, v in sorted(m.items())})
