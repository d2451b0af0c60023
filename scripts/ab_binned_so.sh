#!/bin/bash
# Same-box A/B of the binned product between two builds (new = the tree's libksgpu.so, base = slepc_amd/libksgpu_base.bin): kernel times of
# scripts/spmv_random.py under rocprofv3 --stats, alternating.   usage (through gpurun): scripts/ab_binned_so.sh > gpurun_out/<file>
set -o pipefail
root=$(pwd); out=$root/gpurun_out/ab_bso; rm -rf $out; mkdir -p $out
cp slepc_amd/libksgpu.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
export KSGPU_SPMV=binned
for v in new base new base; do
  if [ $v = base ]; then cp $root/slepc_amd/libksgpu_base.bin $root/slepc_amd/libksgpu.so; else cp /tmp/new.so $root/slepc_amd/libksgpu.so; fi
  rm -rf $out/st
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 $root/scripts/spmv_random.py > $out/log 2>&1
  echo "== $v"
  python3 - $out/st <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_binned" in r["Name"]:
            print("  %-44s calls %3s  average %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/new.so $root/slepc_amd/libksgpu.so
rm -rf $out
