"""Aggregate rocprofv3 --pmc CSV output (one directory per counter group) into kernel,counter,mean rows."""
import csv, glob, sys, collections

root = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean_per_dispatch")
for (k, c), v in sorted(acc.items()):
    print(f'"{k}",{c},{len(v)},{sum(v) / len(v)}')
