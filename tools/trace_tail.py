"""Last kernels of a rocprofv3 --kernel-trace csv: name, grid, LDS, duration.  usage: trace_tail.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 40):]:
    print("%-58s grid %8s wg %5s lds %7s  %8.1f us" % (r["Kernel_Name"][:58], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
                                                       r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
