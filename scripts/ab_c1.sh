#!/bin/bash
cp slepc_amd/libksgpu.so /tmp/new.so
for v in new base new base; do
  if [ $v = base ]; then cp slepc_amd/libksgpu_base.bin slepc_amd/libksgpu.so; else cp /tmp/new.so slepc_amd/libksgpu.so; fi
  echo "=== $v: $(python scripts/c1_trace.py 2>/dev/null | tail -1)"
done
cp /tmp/new.so slepc_amd/libksgpu.so
