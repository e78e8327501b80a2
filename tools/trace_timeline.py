"""One step's kernel timeline out of a rocprofv3 --kernel-trace CSV: start / end (us, relative to the step's first kernel),
duration and queue of every dispatch of the LAST complete step found between two launches of `anchor` (default: the
sampler).    python tools/trace_timeline.py <kernel_trace.csv> [anchor substring] [steps back]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "importance_z_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-back - 1], idx[-back]
# the step = everything from a few dispatches before the anchor (the clear / marking start before the sampler) up to the next one
lo = a
while lo > 0 and int(rows[a]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 40000 and lo > a - 12:
    lo -= 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
