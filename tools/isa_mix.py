#!/usr/bin/env python3
"""Instruction mix of one device kernel from the compiler's gfx950 assembly (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude \
          --cuda-device-only -S -o /tmp/zf_solver.s zfista_amd/csrc/zf_solver.hip
    python tools/isa_mix.py /tmp/zf_solver.s <mangled-kernel-name-substring> [--elems-per-iter 8 --trials 8]

Prints, for the basic block with the most fp64 VALU instructions (the tile loop of the trial
kernel: straight-line code for UB units x 2 elements x S trials), the opcode histogram, the VALU
count per element and trial, and the kernel's register / scratch footprint."""
from __future__ import annotations

import argparse
import collections
import json
import re
import sys


def kernel_text(lines, name):
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and name in ln and ln.rstrip().split(":")[0].startswith("_Z") and ":" in ln:
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel containing {name!r} not found")
    out = []
    for ln in lines[start:]:
        out.append(ln)
        if ln.strip().startswith(".end_amdhsa_kernel"):
            break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--elems-per-block", type=float, default=None,
                    help="elements one pass through the hot block handles per lane (UB units x 2)")
    ap.add_argument("--trials", type=int, default=8)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lines = open(a.asm).read().splitlines()
    body = kernel_text(lines, a.kernel)
    blocks, cur, label = [], [], "entry"
    meta = {}
    for ln in body:
        s = ln.strip()
        if not s or s.startswith(";"):
            m = re.match(r";\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|codeLenInByte|TotalNumVgprs):\s*(\d+)", s)
            if m:
                meta[m.group(1)] = int(m.group(2))
            continue
        if re.match(r"^\.?[A-Za-z_0-9$]+:", s):
            if cur:
                blocks.append((label, cur))
            label, cur = s.split(":")[0], []
            continue
        if s.startswith("."):
            m = re.match(r"\.amdhsa_(next_free_vgpr|private_segment_fixed_size|accum_offset)\s+(\d+)", s)
            if m:
                meta[m.group(1)] = int(m.group(2))
            continue
        cur.append(s.split()[0])
    if cur:
        blocks.append((label, cur))

    def f64_valu(ops):
        return sum(1 for o in ops if o.startswith("v_") and "f64" in o)

    hot_label, hot = max(blocks, key=lambda b: f64_valu(b[1]))
    hist = collections.Counter(hot)
    valu = sum(c for o, c in hist.items() if o.startswith("v_"))
    f64 = f64_valu(hot)
    vmem = sum(c for o, c in hist.items() if o.startswith(("global_", "buffer_", "flat_")))
    salu = sum(c for o, c in hist.items() if o.startswith("s_"))
    report = dict(kernel=a.kernel, hot_block=hot_label, instructions=len(hot), valu=valu, valu_f64=f64,
                  valu_other=valu - f64, vmem=vmem, salu=salu, histogram=dict(hist.most_common()), meta=meta)
    if a.elems_per_block:
        report["valu_per_element_trial"] = valu / a.elems_per_block / a.trials
        report["f64_per_element_trial"] = f64 / a.elems_per_block / a.trials
    whole = collections.Counter(o for _, ops in blocks for o in ops)
    report["whole_kernel"] = dict(instructions=sum(whole.values()),
                                  ds_bpermute=whole.get("ds_bpermute_b32", 0),
                                  scratch=sum(c for o, c in whole.items() if o.startswith("scratch_")))
    txt = json.dumps(report, indent=1)
    print(txt)
    if a.json:
        open(a.json, "w").write(txt + "\n")


if __name__ == "__main__":
    sys.exit(main())
