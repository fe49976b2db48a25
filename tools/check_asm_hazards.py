#!/usr/bin/env python3
"""Static check of the SHIPPED device code for the one hazard the compiler cannot see: `split_pair` (cv_kernels.h) emits
`v_fma_mixlo_f16` / `v_fma_mixhi_f16` through inline asm, and the hazard recogniser does not look inside inline asm when it pads the wait
states a matrix instruction needs before a vector instruction may touch its registers.

What is checked, per kernel of the built library, on the disassembly (no GPU needed): for every v_fma_mix* instruction,
  * its destination register must not be the accumulator (vDst / SrcC) of any v_mfma issued in the WINDOW instructions before it
    (write-after-read on SrcC, write-after-write on vDst), and
  * its vector sources must not be the vDst of such a v_mfma (read-after-write)
unless an ordinary vector instruction in between touched that register first -- the compiler pads ITS instructions' hazards, and the
matrix pipe retires in order, so a padded access to the register settles it for everything later.
WINDOW = 24 instructions: more than the 19 wait states the longest matrix instruction of this code base (16 passes) ever asks for.
Branches are ignored (a linear scan is conservative for straight-line epilogues and loops alike).

    python tools/check_asm_hazards.py [library.so]        # exit 0 = clean; prints every violation otherwise
"""
from __future__ import annotations

import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WINDOW = 24


def regs(tok: str) -> set[int]:
    """v12 -> {12}; v[4:7] -> {4..7}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_disassembly(text: str) -> tuple[int, int, list[str]]:
    """(kernels seen, v_fma_mix instructions seen, violations)"""
    kernels = mixes = 0
    bad: list[str] = []
    name = "?"
    recent: list[tuple[str, list[str]]] = []                     # (mnemonic, operand tokens) of the last WINDOW instructions
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name, recent = m.group(1), []
            kernels += 1
            continue
        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//", line)
        if not m:
            continue
        mnem, ops = m.group(1), [t.strip() for t in m.group(2).split(",")]
        if mnem.startswith("v_fma_mix"):
            mixes += 1
            dst = regs(ops[0])
            srcs = set().union(*(regs(re.split(r"\s", t)[0]) for t in ops[1:]))
            touched: set[int] = set()                                # registers an ordinary vector instruction has read or written since
            for k, (pm, pops) in enumerate(reversed(recent)):          # (nearest first): the compiler padded THAT access, and the matrix
                if pm.startswith("v_mfma"):                            # pipe retires in order -- the hazard was settled there
                    acc = regs(pops[0]) | (regs(pops[3]) if len(pops) > 3 else set())
                    hit = (dst & acc) - touched
                    if hit:
                        bad.append(f"{name}: {mnem} writes v{sorted(hit)} = accumulator of a {pm} issued {k + 1} instructions earlier, untouched in between")
                    hit = (srcs & regs(pops[0])) - touched
                    if hit:
                        bad.append(f"{name}: {mnem} reads v{sorted(hit)} = result of a {pm} issued {k + 1} instructions earlier, untouched in between")
                elif pm.startswith("v_") and not pm.startswith("v_fma_mix"):
                    for t in pops:
                        touched |= regs(re.split(r"\s", t)[0])
        recent.append((mnem, ops))
        if len(recent) > WINDOW:
            recent.pop(0)
    return kernels, mixes, bad


def disassemble(lib: Path) -> str:
    with tempfile.TemporaryDirectory() as tmp:
        copy = Path(tmp) / lib.name
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp)
        out = []
        for co in sorted(Path(tmp).glob(lib.name + ".*gfx950*")):
            out.append(subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout)
        return "\n".join(out)


def main() -> int:
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "chessvision-3lc_amd" / "lib" / "libchessvision_hip.so"
    kernels, mixes, bad = check_disassembly(disassemble(lib))
    print(f"{lib.name}: {kernels} kernels, {mixes} v_fma_mix instructions, {len(bad)} violation(s) within {WINDOW} instructions of a v_mfma")
    for b in bad[:40]:
        print("  " + b)
    return 1 if bad or not mixes else 0


if __name__ == "__main__":
    sys.exit(main())
