"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace output directory (csv): mean gap per (kernel, next kernel) pair.
usage: gaps.py <dir> [last_n_rows]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 4000):]
gaps = collections.defaultdict(list)
for a, b in zip(rows[:-1], rows[1:]):
    ka, kb = a["Kernel_Name"].split("(")[0][-28:], b["Kernel_Name"].split("(")[0][-28:]
    gaps[(ka, kb)].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:16]:
    print("%-30s -> %-30s n %5d  mean gap %.2f us" % (k[0], k[1], len(v), sum(v) / len(v)))
