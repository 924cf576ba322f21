"""tools/trace_tail.py KERNEL_TRACE.csv [batches] -- per-kernel time per batch over the LAST batches of a rocprofv3 kernel trace of
tools/dualiso_batch_bench.py (a batch starts with its k_di_analyse launch): lets two runs of the same code be compared kernel by kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if "k_di_analyse" in r["Kernel_Name"]]
t0 = starts[-nb]
acc = collections.Counter(); cnt = collections.Counter()
for r in rows:
    if int(r["Start_Timestamp"]) >= t0:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlv::", "")
        acc[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
end = max(int(r["End_Timestamp"]) for r in rows)
print(f"last {nb} batches: {(end - t0) / nb / 1e3:9.1f} us per batch from first launch to last end")
for k, v in acc.most_common():
    print(f"  {k:28s} {v / nb / 1e3:9.1f} us per batch  ({cnt[k] / nb:.1f} launches)")
