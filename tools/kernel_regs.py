#!/usr/bin/env python3
"""Prints VGPR / SGPR / spill / scratch figures per kernel of a built libcaustic_hip.so (gfx950 code object).
usage: kernel_regs.py lib.so [name-filter]"""
import re, subprocess, sys, tempfile, os
lib = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "fb"), os.path.join(d, "co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}", "--unbundle"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True))
for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.check_output(["c++filt", g("name")], text=True).strip()
    name = re.sub(r"\(.*", "", name)
    if flt in name:
        print(f"{name:70s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s}")
