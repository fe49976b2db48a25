#!/usr/bin/env python3
"""Register / scratch / occupancy table of one HIP source's kernels (no GPU needed: hipcc cross-compiles gfx950).

    python tools/kernel_resources.py chessvision-3lc_amd/csrc/conv_igemm.hip [name filter ...]

Runs `hipcc --cuda-device-only -Rpass-analysis=kernel-resource-usage` and prints one line per kernel whose demangled name
contains every filter word.  Used before a GPU run to see that a new instantiation neither spills nor loses occupancy."""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path


def main() -> int:
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    src = Path(sys.argv[1]).resolve()
    filters = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        proc = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", str(src), "-o",
                               f"{tmp}/out.co", "-Rpass-analysis=kernel-resource-usage"], cwd=str(src.parent), capture_output=True, text=True)
    if proc.returncode != 0:
        print(proc.stderr[-4000:])
        return proc.returncode
    blocks = re.split(r"remark: Function Name: ", proc.stderr)[1:]
    names = [b.split(" [")[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for b, name in zip(blocks, dem):
        if not all(f in name for f in filters):
            continue
        def field(key: str) -> str:
            m = re.search(key + r": (\w+)", b)
            return m.group(1) if m else "?"
        scratch, occ = field(r"ScratchSize \[bytes/lane\]"), field(r"Occupancy \[waves/SIMD\]")
        print(f"V {field('VGPRs'):>3} A {field('AGPRs'):>3} S {field('TotalSGPRs'):>3} scratch {scratch:>4} occ {occ}  {name.replace('cv::', '')[:150]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
