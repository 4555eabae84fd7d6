"""Cross-check of bench.py's event-timed roofline kernel against the rocprofv3 kernel trace of the same run.

usage: roofline_check.py <rocprof_dir> <bench_json_of_that_run>

The roofline layer (3x3 s1 64->64 at 400x400) runs its own copy of the Winograd kernel (template argument
ROOFLINE = 1: symbol `wino_mfma<..., 1>`), so its launches are exactly the rows of that symbol; only the timed
steps are compared (the last `launches` of them).
"""
import os, csv, glob, json, sys

d, bj = sys.argv[1], sys.argv[2]
b = json.loads(open(bj).read().strip().splitlines()[-1])
r = b['roofline']
f = max(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
import re
rows = [x for x in csv.DictReader(open(f)) if re.search(r'wino_mfma<[^>]*, 1>', x['Kernel_Name'])]
g = int(rows[0]['Grid_Size_X']) * int(rows[0]['Grid_Size_Y']) * int(rows[0]['Grid_Size_Z'])
rows.sort(key=lambda x: int(x['Start_Timestamp']))
rows = rows[-r['launches']:]
dur = [(int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e6 for x in rows]
avg = sum(dur) / len(dur)
print(f"kernel            : {rows[0]['Kernel_Name'].replace('(anonymous namespace)::', '')}  grid={g} threads")
print(f"bench.py          : {r['kernel']}")
print(f"HIP events (bench): avg {r['avg_launch_ms']:.5f} ms over {r['launches']} launches")
print(f"rocprofv3 trace   : avg {avg:.5f} ms over {len(dur)} launches (min {min(dur):.5f}, max {max(dur):.5f})")
print(f"ratio events/trace: {r['avg_launch_ms'] / avg:.4f}")
