"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: hbm_traffic.py <dir_fetch> <dir_write> <kernel-substring> <out.json> [algorithmic_bytes]

Follows /opt/skills/guides/MI355X_MICROARCH.md (HBM): the counters are in KiB; on gfx950 FETCH_SIZE tallies
128-B requests at 64 B for wide coalesced reads, so the corrected figure doubles it (both are written).
"""
import csv, glob, json, sys


def avg(d, counter, sub):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    vals, names = [], set()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter and sub in r['Kernel_Name']:
            vals.append(float(r['Counter_Value']))
            names.add(r['Kernel_Name'])
    if not vals:
        raise SystemExit(f'no {counter} rows for {sub!r} in {f}')
    return sum(vals) / len(vals), len(vals), sorted(names)


fd, wd, sub, out = sys.argv[1:5]
alg = float(sys.argv[5]) if len(sys.argv) > 5 else None
fs, nf, names = avg(fd, 'FETCH_SIZE', sub)
ws, nw, _ = avg(wd, 'WRITE_SIZE', sub)
res = {
    'kernel': names[0].replace('(anonymous namespace)::', ''),
    'launches_fetch_pass': nf, 'launches_write_pass': nw,
    'fetch_size_kib_raw': fs, 'write_size_kib': ws,
    'read_bytes_corrected': 2 * fs * 1024, 'write_bytes': ws * 1024,
    'hbm_bytes_per_launch': 2 * fs * 1024 + ws * 1024,
    'hbm_bytes_per_launch_uncorrected': fs * 1024 + ws * 1024,
    'algorithmic_bytes_per_launch': alg,
    'note': 'FETCH_SIZE doubled per the gfx950 correction for wide coalesced reads; the kernel reads with dword buffer loads, '
            'for which the guide calls the absolute uncalibrated -- the true read side lies between the two figures',
}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
