"""Reduce a `rocprofv3 --kernel-trace` CSV of tools/process_image_latency.py to ONE warm `process_image` call: every kernel between two
resize launches in order, its duration and the idle time of the device in front of it (what the host and the copy engines cost).
usage: python tools/process_image_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "resize_area" in r["Kernel_Name"]]
if len(starts) < 12:
    print("no complete call found"); sys.exit(0)
it = rows[starts[-10]:starts[-9]]
t0, t1 = int(it[0]["Start_Timestamp"]), int(it[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it) / 1e3
print(f"one process_image call: {len(it)} kernels, {(t1 - t0) / 1e3:.1f} us from the resize kernel's start to the last kernel's end, kernels busy {busy:.1f} us")
prev = None
for r in it:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void cv::", "").split("(")[0][:80]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    if gap > 1.5 or any(k in name for k in ("resize", "extract", "head", "stem", "copy", "fill")):
        print(f"{(e - s) / 1e3:8.2f} us  idle before {gap:7.2f} us  {name}")
    prev = e
nxt = int(rows[starts[-9]]["Start_Timestamp"])
print(f"from the last kernel's end to the next call's resize kernel: {(nxt - t1) / 1e3:.1f} us (downloads, decode, Python, upload)")
