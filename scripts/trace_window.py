"""The launches around one restart in a rocprofv3 --kernel-trace CSV: name, duration and the idle time before each (a window around the last
launch whose name contains the pattern given as second argument)."""
import csv, sys, glob, re
rows = []
for path in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("ksk::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_panel_mult"
idx = [i for i, r in enumerate(rows) if pat in r[2]]
c = idx[-2] if len(idx) > 1 else idx[-1]
lo, hi = max(1, c - 8), min(len(rows), c + 14)
for i in range(lo, hi):
    s, e, n = rows[i]
    print("  gap %7.2f us   run %7.2f us   %s" % ((s - rows[i - 1][1]) / 1e3, (e - s) / 1e3, n[:60]))
