#!/bin/bash
# rocprofv3 kernel trace of the default bench command (no counters), reduced to the conv family's average launch duration
# over the TIMED region only (the engine's load-time calibration and the warm-up launch the same kernels on other sizes).
#   usage (GPU box, repo root): bash tools/rocprof_bench.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/rocprof}; shift || true
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
STEPS=3; WARM=1
rm -rf /tmp/rp_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -o r -- python3 "$REPO/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras "$@" > "$REPO/$OUT/bench_under_rocprof.json" 2> "$REPO/$OUT/rocprof.err")
find /tmp/rp_bench -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
TRACE=$(find /tmp/rp_bench -name '*kernel_trace.csv' | head -1)
python3 - "$TRACE" "$OUT/bench_under_rocprof.json" "$OUT/kernel_trace_reduced.json" $STEPS $WARM <<'PY'
import csv, json, sys
trace, bench_json, out, steps, warm = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
conv = [r for r in rows if any(t in r["Kernel_Name"] for t in ("conv_igemm_kernel", "conv3x3_halo_kernel", "inc0_mfma_kernel", "shortcut1x1s2"))]
line = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
per_step = line["roofline"]["launches_per_step"]
# order of conv launches in the process: calibration ... | warm-up | timed region | then, per model (UNet, ResNet-18), one un-timed
# pass followed by the event-timed profiling pass (bench.py: rooflines())
timed = conv[-(steps + 2) * per_step:-2 * per_step]
tail = conv[-2 * per_step:]
n_unet = line["roofline"]["launches_by_model"]["unet"]
prof = tail[n_unet:2 * n_unet] + tail[2 * n_unet + (per_step - n_unet):]
dur = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
by_kernel = {}
for r in timed:
    k = r["Kernel_Name"].split("(")[0].replace("void cv::", "")
    a = by_kernel.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
res = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-extras" % (steps, warm),
       "conv_launches_in_process": len(conv), "launches_per_step": per_step,
       "timed_region": {"conv_launches": len(timed), "conv_ms_total": dur(timed), "avg_launch_ms": dur(timed) / max(1, len(timed))},
       "event_timed_pass": {"conv_launches": len(prof), "avg_launch_ms": dur(prof) / max(1, len(prof))},
       "bench_line": {"avg_launch_ms": line["roofline"]["avg_launch_ms"], "ms_per_step": line["ms_per_step"], "value": line["value"],
                      "achieved_tflops": line["roofline"]["achieved"], "frac": line["roofline"]["frac"]},
       "timed_region_by_kernel": {k: {"launches": n, "avg_ms": t / n} for k, (n, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])},
       "note": "kernel_stats.csv averages every dispatch of the process, including the load-time range calibration (batch of 2 images / "
               "128 squares) and the warm-up; this reduction keeps the timed region's launches only"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("timed_region", "event_timed_pass", "bench_line")}))
PY
