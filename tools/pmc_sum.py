"""Average of each PMC counter per dispatch of kernels whose name contains a substring.
    python tools/pmc_sum.py DIR sf_render_kernel"""
import collections, csv, glob, os, sys
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
