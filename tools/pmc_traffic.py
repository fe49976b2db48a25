"""Reduce rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate passes) to per-launch HBM traffic of the
conv kernel family, with the gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per 128-B
request for 16-B/lane streaming reads (LDS-DMA included) -> double it; WRITE_SIZE is exact for 16-B/lane stores.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <label>
"""
import collections
import csv
import hashlib
import json
import sys
from pathlib import Path


def kernel_source_hash():
    """sha256 over the sources the kernels are built from (chessvision-3lc_amd/csrc: *.hip, *.h, *.cpp, Makefile), file names included,
    sorted: bench.py recomputes it and reports `traffic: null` when the committed counters were collected on other kernels."""
    root = Path(__file__).resolve().parent.parent / "chessvision-3lc_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted(p for p in root.iterdir() if p.suffix in (".hip", ".h", ".cpp") or p.name == "Makefile"):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def per_dispatch(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return d


def main():
    fetch, write, out, label = sys.argv[1:5]
    f, w = per_dispatch(fetch, "FETCH_SIZE"), per_dispatch(write, "WRITE_SIZE")
    conv = lambda n: "conv_igemm" in n or "conv3x3_halo" in n or "inc0_mfma" in n or "shortcut1x1s2" in n      # the kernels of the conv family
    fc = [v for k, (n, v) in f.items() if conv(n)]
    wc = [v for k, (n, v) in w.items() if conv(n)]
    assert len(fc) == len(wc) and fc, (len(fc), len(wc))
    n = len(fc)
    read_b = 2.0 * sum(fc) * 1024 / n          # FETCH_SIZE is in KiB; x2 = gfx950 wide-read correction
    write_b = sum(wc) * 1024 / n
    res = {"label": label, "conv_launches_profiled": n, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": read_b + write_b,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; read = 2*FETCH_SIZE*1024 (gfx950 "
                     "correction for 16-B/lane reads), write = WRITE_SIZE*1024"}
    # the memory-bound kernels of the same run, by kernel name (same corrections; their access widths are 16 or 32 B per lane)
    other = {}
    for k, (n_, v) in f.items():
        if conv(n_):
            continue
        short = n_.split("(")[0].replace("void cv::", "")
        if not any(t in short for t in ("stem", "head_kernel", "pack_", "upsample", "maxpool", "outc")):
            continue
        o = other.setdefault(short, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
        o["launches"] += 1
        o["read_bytes"] += 2.0 * v * 1024
        o["write_bytes"] += w.get(k, (None, 0.0))[1] * 1024
    res["memory_bound_kernels"] = {k: {"launches": o["launches"], "read_bytes_per_launch": o["read_bytes"] / o["launches"],
                                       "write_bytes_per_launch": o["write_bytes"] / o["launches"]} for k, o in other.items()}
    try:
        allres = json.load(open(out))
    except (OSError, ValueError):
        allres = {}
    res["kernel_source_sha256"] = kernel_source_hash()
    allres[label] = res
    json.dump(allres, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
