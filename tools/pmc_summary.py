"""Summarise rocprofv3 --pmc counter CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
                acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
        print("==", f)
        for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"{k:70s} {c:12s} dispatches={n:5d} mean={s / n:14.1f} total={s:16.1f}")
