"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel symbol and counter: launches, mean, min, max."""
import csv, collections, glob, re, statistics, sys, json
def summarize(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(path)):
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("ksk::", "")
                name = name.split("(")[0]
                t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                agg[name][r["Counter_Name"]].append((float(r["Counter_Value"]), t))
    return agg
if __name__ == "__main__":
    agg = summarize(sys.argv[1:])
    for k, cs in sorted(agg.items()):
        if not re.search(r"k_(dot|gs_update|spmv|panel|gs_finish)", k): continue
        for c, v in sorted(cs.items()):
            vals = [x for x, _ in v]; ts = [t for _, t in v]
            print("%-34s %-30s n=%4d mean=%14.1f min=%14.1f max=%14.1f  mean_ns=%10.0f" % (k[:34], c, len(vals), statistics.mean(vals), min(vals), max(vals), statistics.mean(ts)))
