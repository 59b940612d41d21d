"""Registers, scratch, LDS and occupancy of every instantiation of a kernel, from hipcc's resource-usage remarks:
    python tools/kernel_resources.py colpass.hip k_colpass [filter-substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, kern = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
                        "-ffp-contract=off", f"-I{ROOT}/include", "-c", f"{ROOT}/mbpls_amd/csrc/{src}", "-o", f"{d}/o.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
txt = r.stderr
blocks = re.split(r"remark: Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    m = re.search(kern + r"(?:<(.*)>)?\(", d)
    if not m or flt not in (m.group(1) or ""):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(f"{(m.group(1) or kern):58s} VGPR {g('    VGPRs'):>3s} AGPR {g('AGPRs'):>3s} scratch {g('ScratchSize .bytes/lane.'):>4s} "
          f"occupancy {g('Occupancy .waves/SIMD.')} LDS {g('LDS Size .bytes/block.')}")
