#!/usr/bin/env python3
"""tools/census.py [kernel-symbol] [out.txt] -- per-section instruction census of a kernel from its ISA.

The section-stamp build of the kernels (`-DGD_DEV -DGD_ABL=30` for k_step, `=34` for k_fill: csrc/gdyn_stamps.h) puts an
`s_memtime` at every section boundary; the product build carries the same code without them (its ISA is checked to have the
same VALU count outside the stamp bookkeeping).  This tool compiles the stamped build to assembly, cuts the kernel at the
`s_memtime` markers and counts, per section, the STATIC instructions by issue class (VALU / SALU / LDS / VMEM / branch+wait),
with every loop body (compiler-marked `Loop Header`) listed on its own: the dynamic count of a section is its straight-line
part + sum over its loops of body x trips; the trip counts of the benchmark state are given beside the loops they belong to
in profiles/r05_kstep_census.txt."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECTIONS_STEP = ["entry: block map, scalar context loads", "record + tile descriptor (wait for the record)", "bond table + tile DMA issue",
                 "per-bead loads issued (build position, first adjacency and list chunks)", "noise (Philox, Box-Muller; waves 1-7)",
                 "wave 0: pending callback, context constants", "barrier (tile arrival)", "own position, skin check, pair loop", "bonds",
                 "bending + point sources", "wall", "integrate, displacement bound, reductions, store", "epilogue (stamp write-out)"]


def classify(op):
    if op.startswith(("v_", "ds_bpermute", "ds_swizzle")) and not op.startswith("v_readfirstlane_never"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_endpgm", "s_setprio")):
        return "ctl"
    if op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_buffer"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    sym = sys.argv[1] if len(sys.argv) > 1 else "_Z6k_stepILi0ELb0ELb1ELi1ELb1ELb0EEv10StepParams"
    out = sys.argv[2] if len(sys.argv) > 2 else None
    abl = "34" if "k_fill" in sym else "30"
    src = os.path.join(ROOT, "2022a-genome-dynamics_amd", "csrc", "gdyn_kernels.hip")
    asm = f"/tmp/census_{abl}.s"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                           "-DGD_DEV", f"-DGD_ABL={abl}", "-o", asm, src], stderr=subprocess.DEVNULL)
    lines = open(asm).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    # sections: cut at s_memtime; loops: from a "Loop Header" label to the last backward branch to it
    sections, cur = [], []
    for ln in body:
        cur.append(ln)
        if re.match(r"\s+s_memtime", ln):
            sections.append(cur); cur = []
    sections.append(cur)
    rows = []
    for k, sec in enumerate(sections):
        # loop ranges inside the section
        labels = {m.group(1): i for i, ln in enumerate(sec) for m in [re.match(r"(\.LBB\d+_\d+):", ln)] if m}
        headers = [i for i, ln in enumerate(sec) if "Loop Header" in ln]
        loops = []
        for h in headers:
            lab = re.match(r"(\.LBB\d+_\d+):", sec[h]).group(1)
            depth = int(re.search(r"Depth=(\d+)", sec[h]).group(1))
            last = h
            for i in range(h, len(sec)):
                m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", sec[i])
                if m and m.group(1) == lab:
                    last = i
                # blocks of the loop that sit in front of / behind the header are marked "in Loop: Header=BBn_m"
            hdr_tag = "Header=" + lab[2:]
            members = [i for i, ln in enumerate(sec) if hdr_tag in ln and re.match(r"\.LBB", ln)]
            lo = min([h] + members)
            # extent: up to the last line that belongs to a block tagged with this header (the block runs to the next label)
            hi = last
            for mi in members:
                j = mi + 1
                while j < len(sec) and not re.match(r"\.LBB\d+_\d+:", sec[j]):
                    j += 1
                hi = max(hi, j - 1)
            loops.append((lo, hi, depth, lab))
        in_loop = [None] * len(sec)
        for idx, (lo, hi, depth, lab) in enumerate(sorted(loops, key=lambda t: t[2])):      # deeper loops overwrite
            for i in range(lo, hi + 1):
                in_loop[i] = (lab, depth)
        cnt = {}
        for i, ln in enumerate(sec):
            m = re.match(r"\s+([a-z_0-9]+)", ln)
            if not m or ln.strip().startswith((";", ".")):
                continue
            key = in_loop[i][0] if in_loop[i] else "straight"
            c = cnt.setdefault(key, {})
            cl = classify(m.group(1))
            c[cl] = c.get(cl, 0) + 1
        rows.append((k, cnt, {lab: depth for (_, _, depth, lab) in loops}))
    names = SECTIONS_STEP if "k_step" in sym else [f"section {i}" for i in range(len(sections))]
    text = [f"# static instruction census of {sym} (stamped build -DGD_ABL={abl}); classes: VALU / SALU / LDS / VMEM / SMEM / ctl (branches, waits)", ""]
    tot = {}
    for k, cnt, depths in rows:
        text.append(f"[{k:2d}] {names[k] if k < len(names) else ''}")
        for key in sorted(cnt, key=lambda x: (x != "straight", x)):
            c = cnt[key]
            tag = "straight-line" if key == "straight" else f"loop {key} (depth {depths.get(key, '?')})"
            text.append("      %-28s VALU %4d  SALU %4d  LDS %3d  VMEM %3d  SMEM %3d  ctl %3d" %
                        (tag, c.get("VALU", 0), c.get("SALU", 0), c.get("LDS", 0), c.get("VMEM", 0), c.get("SMEM", 0), c.get("ctl", 0)))
            for cl, v in c.items():
                tot[cl] = tot.get(cl, 0) + v
    text.append("")
    text.append("static total: " + "  ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    s = "\n".join(text) + "\n"
    if out:
        open(out, "w").write(s)
    sys.stdout.write(s)


if __name__ == "__main__":
    main()
