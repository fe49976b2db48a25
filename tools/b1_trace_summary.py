"""Reduce a `rocprofv3 --kernel-trace` CSV of tools/b1_trace.py to one iteration of the single-board path: every launch of the last
UNet B=1 + ResNet-18 B=64 pair in order, with its duration (end of the previous kernel to its own end = what a dependent chain pays)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
# one iteration = from a fused first-layer launch (FUSE0 instantiation: ", true>" as the last template flag of the halo kernel) to the next
import re
fuse0 = re.compile(r"true(, \d+)?>\(cv::ConvParams\)$")       # round 5: the template gained a trailing CHAIN parameter
starts = [i for i, r in enumerate(rows) if "conv3x3_halo_kernel" in r["Kernel_Name"] and fuse0.search(r["Kernel_Name"].rstrip())]
if len(starts) < 3:
    print("no complete iteration found"); sys.exit(0)
it = rows[starts[-2]:starts[-1]]
total = (int(it[-1]["End_Timestamp"]) - int(it[0]["Start_Timestamp"])) / 1e3
print(f"one iteration of the single-board path (UNet B=1 + ResNet-18 B=64, f16x3): {len(it)} launches, {total:.1f} us from first start to last end")
prev = None
agg = {}
for r in it:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void cv::", "").split("(")[0][:96]
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    wgs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(e - s) / 1e3:8.2f} us  idle before {gap:6.2f} us  workgroups {wgs:6d}  {name}")
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    prev = e
print("\nby kernel:")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  x{n:3d}  avg {t / n:6.2f} us  {name}")
