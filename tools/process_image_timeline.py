"""Merge the rocprofv3 kernel / HIP-API / memory-copy traces of tools/process_image_latency.py into the timeline of ONE warm
`process_image` call (developer tool).  usage: python tools/process_image_timeline.py <trace dir> [call index]"""
import csv
import glob
import sys

d = sys.argv[1]
def load(pat):
    f = glob.glob(f"{d}/**/*{pat}", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
kern, api, cp = load("kernel_trace.csv"), load("hip_api_trace.csv"), load("memory_copy_trace.csv")
kern.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(kern) if "resize_area" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30                      # which resize launch = which call (the tool warms up with 8 calls)
a, b = int(kern[starts[k]]["Start_Timestamp"]) - 120000, int(kern[starts[k + 1]]["Start_Timestamp"]) - 120000
ev = []
for r in kern:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a <= s < b: ev.append((s, e, "GPU ", r["Kernel_Name"].replace("void cv::", "").split("(")[0][:70]))
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a <= s < b and (e - s > 1500 or "Graph" in r["Function"] or "Synchronize" in r["Function"] or "Memcpy" in r["Function"]):
        ev.append((s, e, "HOST", r["Function"]))
for r in cp:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a <= s < b: ev.append((s, e, "COPY", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
t0 = ev[0][0]
last_gpu = None
for s, e, kind, name in ev:
    if kind == "GPU " and last_gpu is not None and s - last_gpu < 1500 and not any(k in name for k in ("resize", "extract", "stem", "head", "copy")):
        last_gpu = e
        continue
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {kind}  {name}")
    if kind == "GPU ": last_gpu = e
