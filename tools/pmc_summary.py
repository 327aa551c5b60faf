"""Per-kernel means of rocprofv3 --pmc counters (developer tool): python tools/pmc_summary.py DIR [DIR...] > out.json"""
import csv, glob, json, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            short = next((s for s in ("k_knn_pair", "k_knn_fast", "k_knn_exact", "k_fit_svd", "k_fit", "k_hist", "k_scatter", "k_pack") if s in name), None)
            if short is None:
                continue
            per_dispatch[(short, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (short, _, cname), v in per_dispatch.items():
            acc[short][cname].append(v)
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, indent=1))
