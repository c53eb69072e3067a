"""tools/trace_gaps.py DIR -- dev: per-kernel durations AND inter-kernel gaps from a rocprofv3 --kernel-trace CSV of a
graph-replayed bench.py run (the --stats averages are of launches the tracer may have serialised; the begin/end
timestamps say what actually happened back to back).  Prints, for the two kernels of the Layer-API step, the median
duration, the median gap to the next kernel, and the implied time per step."""
import csv
import glob
import statistics as st
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "euclid" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fwd = [r for r in rows if "pair32" in r["Kernel_Name"]]
print("%d euclid launches traced (%d forward)" % (len(rows), len(fwd)))
# keep the long steady run: consecutive (forward, backward) pairs whose gap to the previous kernel is < 20 us
dur = {"fwd": [], "bwd": []}
gap = {"fwd->bwd": [], "bwd->fwd": []}
step = []
for i in range(1, len(rows) - 1):
    a, b = rows[i], rows[i + 1]
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 20000:
        continue
    kind = "fwd" if "pair32" in a["Kernel_Name"] else "bwd"
    dur[kind].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap["fwd->bwd" if kind == "fwd" else "bwd->fwd"].append(g)
    if kind == "fwd" and i + 2 < len(rows) and "pair32" in rows[i + 2]["Kernel_Name"]:
        s = int(rows[i + 2]["Start_Timestamp"]) - int(a["Start_Timestamp"])
        if s < 40000:
            step.append(s)
for k, v in dur.items():
    if v:
        print("  %s kernel: median %.2f us  (p10 %.2f, p90 %.2f)  n=%d" % (k, st.median(v) / 1e3, sorted(v)[len(v) // 10] / 1e3, sorted(v)[9 * len(v) // 10] / 1e3, len(v)))
for k, v in gap.items():
    if v:
        print("  gap %s: median %.2f us  (p10 %.2f, p90 %.2f)" % (k, st.median(v) / 1e3, sorted(v)[len(v) // 10] / 1e3, sorted(v)[9 * len(v) // 10] / 1e3))
if step:
    print("  forward start -> next forward start: median %.2f us per step over %d steps" % (st.median(step) / 1e3, len(step)))
    print("  (sum of medians: %.2f us)" % ((st.median(dur["fwd"]) + st.median(dur["bwd"]) + st.median(gap["fwd->bwd"]) + st.median(gap["bwd->fwd"])) / 1e3))
