"""tools/kt_tail.py DIR [N] -- dev-only: print the last N kernels of a rocprofv3 --kernel-trace csv (our kernels)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = [r for r in csv.DictReader(open(f)) if "mms" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    wg = int(r["Workgroup_Size_X"])
    print("%7.1f us (gap %5.1f)  wgs %5d x %3s x %3s  lds %6s  %s" % ((e - s) / 1e3, gap, int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], r["Grid_Size_Z"], r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:84]))
