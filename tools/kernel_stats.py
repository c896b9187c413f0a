#!/usr/bin/env python3
"""Per-kernel register, scratch, LDS and code size of one translation unit of the library (device-only compile, gfx950):
    python tools/kernel_stats.py kernels_pipe.hip [extra hipcc flags]
scratch (private_segment_fixed_size) other than 0 means spills or a run-time-indexed register array."""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "knaster_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "kernels_pipe.hip"
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only",
                        "--no-gpu-bundle-output", "-c", src, "-o", co, *sys.argv[2:]], cwd=CSRC, check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True, check=True).stdout
    size = {}
    for ln in syms.splitlines():
        f = ln.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    rows = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("knh_dev::", "").replace("void ", "")
        rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), size.get(name, 0)))
    for r in sorted(rows):
        print(f"{r[1]:>4} vgpr {r[2]:>4} sgpr {r[3]:>6} scratch {r[4]:>7} lds {r[5]:>8} code B  {r[0][:150]}")


if __name__ == "__main__":
    main()
