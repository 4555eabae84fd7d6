"""Key figures of a bench.py JSON line.  usage: tools/print_bench.py <file.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']} host_start {d.get('value_host_start')} dtype {d['dtype']} frac {r['frac']} achieved {r['achieved']} launch_ms {r['avg_launch_ms']} traffic {r.get('traffic')}")
if "whole_frame" in r:
    print("whole_frame", r["whole_frame"])
    print({k: (v["ms_per_frame"], v["frac"]) for k, v in r["stages"].items()})
if "extras" in d:
    print({k: v for k, v in d["extras"].items() if k != "tilings"})
if "cpu_baseline" in d:
    print(d["cpu_baseline"])
