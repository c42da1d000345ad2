#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in gfx950 assembly files (hipcc -S --cuda-device-only): no GPU needed.
    tools/kernel_resources.py file.s [file.s ...]"""
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


for f in sys.argv[1:]:
    t = open(f).read()
    print("==", f)
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: int(re.search(rf"\.amdhsa_{k} (\d+)", body).group(1))   # noqa: E731
        vg, sc, lds = g("next_free_vgpr"), g("private_segment_fixed_size"), g("group_segment_fixed_size")
        waves = 512 // ((vg + 7) // 8 * 8) if vg else 8
        short = re.sub(r"\(zf_step_args.*", "", demangle(name)).replace("void ", "")
        print(f"  {short:78s} vgpr {vg:3d} (waves/SIMD {min(waves, 8)})  lds {lds:6d}  scratch {sc}")
