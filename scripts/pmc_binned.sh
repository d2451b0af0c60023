#!/bin/bash
# Kernel times and FETCH_SIZE / WRITE_SIZE of the binned and the XCD-sliced product of the config-5-shaped matrix (n = 5e6, 1.65e8 nonzeros).
# usage (through gpurun, from the repository root): scripts/pmc_binned.sh > gpurun_out/r02_pmc_binned_spmv.txt
set -o pipefail
root=$(pwd); out=$root/gpurun_out/pmc_binned; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for layout in ${LAYOUTS:-binned sliced}; do
  export KSGPU_SPMV=$layout
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${layout}_stats -- python3 $root/scripts/spmv_random.py > $out/${layout}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${layout}_fetch -- python3 $root/scripts/spmv_random.py > $out/${layout}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${layout}_write -- python3 $root/scripts/spmv_random.py > $out/${layout}_write.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections, re
def kname(t):
    m = re.search(r'(k_[a-z_]+)', t)
    return m.group(1) if m else t[:24]
import os
for layout in os.environ.get("LAYOUTS", "binned sliced").split():
    print("== KSGPU_SPMV=%s" % layout)
    for f in glob.glob("gpurun_out/pmc_binned/%s_stats/*/*kernel_stats.csv" % layout):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in ("k_binned", "k_spmv_sliced", "k_sum_parts")):
                print("  %-24s calls %3s  average %9.1f us" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
    for ctr, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):          # KiB units; gfx950: FETCH_SIZE counts 64-byte halves (x2)
        tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
        for f in glob.glob("gpurun_out/pmc_binned/%s_%s/*/*counter_collection.csv" % (layout, "fetch" if ctr == "FETCH_SIZE" else "write")):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr and any(k in r["Kernel_Name"] for k in ("k_binned", "k_spmv_sliced", "k_sum_parts")):
                    nm = kname(r["Kernel_Name"]); tot[nm] += float(r["Counter_Value"]); cnt[nm] += 1
        for nm in tot:
            print("  %-24s %-10s %9.1f MB per launch" % (nm, ctr, mul * tot[nm] / cnt[nm] * 1024 / 1e6))
PY
rm -rf $out
