"""print a rocprofv3 kernel_stats.csv: name, calls, average, share"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:40]:40s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
