"""Per kernel name: the dispatch with the largest SQ_INSTS_VALU (an active launch), all its counters."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        disp = collections.defaultdict(dict)
        name = {}
        for r in csv.DictReader(open(f)):
            disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kfh::", "")
        best = {}
        for k, c in disp.items():
            key = list(c.keys())[0] if "SQ_INSTS_VALU" not in c else "SQ_INSTS_VALU"
            if name[k] not in best or c[key] > disp[best[name[k]]][key]:
                best[name[k]] = k
        for n, k in sorted(best.items()):
            print(n[-36:], {a: "%.4g" % b for a, b in sorted(disp[k].items())})
