"""Average rocprofv3 --pmc counters per kernel name over the last dispatches of each kernel."""
import csv, glob, sys, collections
import numpy as np
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-24:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            if not any(x in k for x in ("integrate", "raycast", "icp_iter")):
                continue
            print(k, {c: "%.4g" % np.mean(v[-4:]) for c, v in agg[k].items()})
