"""HBM traffic per launch of the roofline kernel from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one
pass each, per forced tiling).

usage: hbm_traffic.py <out.json> <frames> <algorithmic_bytes> <tiling>=<dir_fetch>,<dir_write> [...]

The roofline layer (3x3 s1 64->64 at 400x400) runs its own copy of the Winograd kernel (symbol
`wino_mfma<..., 1>` / `wino4_mfma<..., 1>` / `wino6_mfma<4, 1>`), so its launches are the rows of that symbol.  Follows /opt/skills/guides/MI355X_MICROARCH.md (HBM): the counters are in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, so the corrected figure doubles it (both
are written; the kernel reads dwords, for which the guide calls the absolute uncalibrated, so the true read
side lies between the two).
"""
import os, csv, glob, json, re, sys


def rows(d, counter):
    f = max(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    rs = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and re.search(r'wino[46]?_mfma<[^>]*, 1>', r['Kernel_Name'])]
    if not rs:
        raise SystemExit(f'no {counter} rows for the roofline copy wino(4|6)_mfma<..., 1> in {f}')
    vals = [float(r['Counter_Value']) for r in rs]
    return sum(vals) / len(vals), len(vals), rs[0]['Kernel_Name'].replace('(anonymous namespace)::', '')


out, frames, alg = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # source_hash(): ties the figures to the HIP sources they were taken on (bench.py drops them when it differs)
res = {'source_hash': bench.source_hash(), 'frames_per_launch': frames, 'algorithmic_bytes_per_launch': alg, 'layer': 'conv 3x3 s1 64->64 @ 400x400',
       'unit': 'bytes', 'tilings': {}}
for spec in sys.argv[4:]:
    name, dirs = spec.split('=')
    fd, wd = dirs.split(',')
    fs, nf, kern = rows(fd, 'FETCH_SIZE')
    ws, nw, _ = rows(wd, 'WRITE_SIZE')
    res['tilings'][name] = {
        'kernel': kern, 'launches_fetch_pass': nf, 'launches_write_pass': nw,
        'fetch_size_kib_raw': fs, 'write_size_kib': ws,
        'read_bytes_corrected': 2 * fs * 1024, 'write_bytes': ws * 1024,
        'hbm_bytes_per_launch': 2 * fs * 1024 + ws * 1024,
        'hbm_bytes_per_launch_uncorrected': fs * 1024 + ws * 1024,
    }
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
