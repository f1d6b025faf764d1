#!/usr/bin/env python3
"""tools/kregs.py [file.s] [filter]: VGPR / SGPR / spill / LDS table of every kernel in a device assembly listing
(hipcc -S --cuda-device-only; default: compiles csrc/gdyn_kernels.hip to /tmp/k.s first).  The register and LDS budgets
of DESIGN.md section 4 are checked with it after every kernel change."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else None
    flt = [a for a in sys.argv[1:] if not a.endswith(".s")]
    if path is None:
        path = "/tmp/k.s"
        src = os.path.join(ROOT, "2022a-genome-dynamics_amd", "csrc", "gdyn_kernels.hip")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                               "-o", path, src], stderr=subprocess.DEVNULL)
    text = open(path).read()
    # the amdhsa.kernels metadata: one YAML record per kernel
    recs = re.split(r"\n  - \.agpr_count:", text[text.find("amdhsa.kernels:"):])[1:]
    rows = []
    for r in recs:
        def f(key):
            m = re.search(r"\." + key + r":\s*(\S+)", r)
            return m.group(1) if m else "?"
        name = f("name")
        try:
            name = subprocess.check_output(["c++filt", name], text=True).strip().replace("(StepParams)", "").replace("(BuildParams)", "")
        except (OSError, subprocess.CalledProcessError):
            pass
        rows.append((name, f("vgpr_count"), f("sgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"), f("group_segment_fixed_size"),
                     f("private_segment_fixed_size")))
    print(f"{'kernel':70s} vgpr sgpr vspill sspill  lds  scratch")
    for row in rows:
        if flt and not all(x in row[0] for x in flt):
            continue
        print(f"{row[0][:70]:70s} {row[1]:>4s} {row[2]:>4s} {row[3]:>6s} {row[4]:>6s} {row[5]:>5s} {row[6]:>6s}")


if __name__ == "__main__":
    main()
