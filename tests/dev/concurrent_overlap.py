"""(test infrastructure) Reduce a rocprofv3 kernel trace of tests/dev/concurrent_probe.py to: wall time of the traced window, time with
at least one kernel running, sum of kernel durations (= overlap when larger than the busy time), and the same per kernel family.
usage: python tests/dev/concurrent_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]                                  # the warm part
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
t0, t1 = ev[0][0], max(e for _, e, _ in ev)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _ in ev)
calls = sum("resize_area" in n for _, _, n in ev)
print(f"window {(t1 - t0) / 1e6:.2f} ms, {calls} calls = {calls / ((t1 - t0) / 1e9):.0f} calls/s; device busy {busy / (t1 - t0):.3f} of the window; "
      f"sum of kernel durations {total / (t1 - t0):.3f} of the window (kernel time per call {total / max(calls, 1) / 1e3:.1f} us, busy time per call {busy / max(calls, 1) / 1e3:.1f} us)")
fam = {}
for s, e, n in ev:
    key = n.replace("void cv::", "").split("<")[0].split("(")[0][:40]
    f = fam.setdefault(key, [0, 0])
    f[0] += e - s; f[1] += 1
for k, (d, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"   {k:42s} {d / max(calls, 1) / 1e3:8.1f} us per call  {c / max(calls, 1):6.1f} launches per call  {d / c / 1e3:7.2f} us each")
