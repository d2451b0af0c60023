"""Build profiles/traffic_rNN.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command:
HBM bytes per EXECUTED launch of every sweep / SpMV / panel kernel symbol. FETCH_SIZE counts KiB and, on gfx950, only
half of the bytes of a 128-B request (x2 correction calibrated in profiles/r01b_pmc_calibration_and_mfma_summary.txt);
launches that exit at their device-side gate (fetch < 1 MB) are counted but excluded from the per-launch mean.
usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [source description]"""
import csv, glob, json, re, sys, collections

def load(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("ksk::", "").split("(")[0]
            out[name].append(float(r["Counter_Value"]))
    return out

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for name, f in sorted(fetch.items()):
    if not re.search(r"k_(dot_sweep|gs_update|spmv|panel)", name):
        continue
    w = write.get(name, [])
    ex = [v for v in f if v > 1024.0]
    if not ex:
        continue
    exw = sorted(w)[len(w) - len(ex):] if len(w) >= len(ex) else w      # the executed launches are the ones that write
    fb = 2.0 * 1024.0 * sum(ex) / len(ex)
    wb = 1024.0 * (sum(exw) / len(exw) if exw else 0.0)
    res[name] = {"launches_total": len(f), "launches_executed": len(ex), "fetch_bytes_per_executed_launch": fb,
                 "write_bytes_per_executed_launch": wb, "hbm_bytes_per_executed_launch": fb + wb,
                 "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, x2 gfx950 correction for FETCH_SIZE"}
if len(sys.argv) > 4:
    res["_source"] = sys.argv[4]
json.dump(res, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(res), "kernel symbols")
