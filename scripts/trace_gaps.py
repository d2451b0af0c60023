"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: per (previous kernel -> next kernel) pair the mean idle time."""
import csv, sys, glob, re, collections
rows = []
for path in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").replace("ksk::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2        # steady state: second half
rows = rows[skip:]
gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(n0, n1)].append(s1 - e0)
for s, e, n in rows:
    dur[n].append(e - s)
tot = rows[-1][1] - rows[0][0]
print("span %.3f ms, %d kernels, busy %.1f %%" % (tot / 1e6, len(rows), 100.0 * sum(e - s for s, e, _ in rows) / tot))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("  %-44s n=%5d mean %8.2f us" % (k[:44], len(v), sum(v) / len(v) / 1e3))
print("gaps:")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("  %-36s -> %-36s n=%5d mean %7.2f us" % (k[0][:36], k[1][:36], len(v), sum(v) / len(v) / 1e3))
