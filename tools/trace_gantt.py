"""tools/trace_gantt.py KERNEL_TRACE.csv -- the launches of the LAST batch of a rocprofv3 kernel trace of tools/dualiso_batch_bench.py
(a batch starts with its k_di_analyse launch): start and end relative to the batch's first launch, queue, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = [int(r["Start_Timestamp"]) for r in rows if "k_di_analyse" in r["Kernel_Name"]][-1]
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a >= t0:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlv::", "")
        print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f}  {(b - a) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {k}")
