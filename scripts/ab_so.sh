#!/bin/bash
# A/B of two builds of libksgpu.so on one box: new, base, new, base (base = slepc_amd/libksgpu_base.bin, a copy of the other build's .so)
cp slepc_amd/libksgpu.so /tmp/new.so
for v in new base new base; do
  if [ $v = base ]; then cp slepc_amd/libksgpu_base.bin slepc_amd/libksgpu.so; else cp /tmp/new.so slepc_amd/libksgpu.so; fi
  echo "=== $v"
  python scripts/ab_run.py "$@" 2>/dev/null
done
cp /tmp/new.so slepc_amd/libksgpu.so
