#!/usr/bin/env python3
"""Static check of the device code: every quad broadcast (four v_mov_b32_dpp with quad_perm [j,j,j,j], j = 0..3, see
quad_pair_slots in kernels.hpp) must execute as one group with no change of the exec mask in between.  A cross-lane read
that the compiler moved into a divergent branch reads 0 from the lanes that are switched off there.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o full.s caustic_hip.hip
       python tools/check_dpp_groups.py full.s"""
import re
import sys

lines = [l.strip() for l in open(sys.argv[1])]
good = bad = 0
fn = None
i = 0
while i < len(lines):
    t = lines[i]
    m = re.match(r"^(_Z\w+):", t)
    if m:
        fn = m.group(1)
    if "v_mov_b32_dpp" in t and "quad_perm" in t:
        perms, j = [], i
        while j < len(lines) and j < i + 12:
            u = lines[j]
            if "saveexec" in u or u.startswith("s_cbranch") or u.startswith("s_or_b64 exec") or u.endswith(":"):
                break
            mm = re.search(r"quad_perm:\[(\d),", u)
            if mm:
                perms.append(int(mm.group(1)))
            j += 1
        if sorted(set(perms)) == [0, 1, 2, 3]:
            good += 1
        else:
            bad += 1
            print("split group in", fn, "at line", i + 1)
        i = max(j, i + 1)
    else:
        i += 1
print(f"{good} intact groups, {bad} split")
sys.exit(1 if bad else 0)
