#!/bin/bash
# A/B of two builds on one box for the small and mid-size configurations: new, base, new, base (slepc_amd/libksgpu_base.bin = the other build)
cp slepc_amd/libksgpu.so /tmp/new.so
for v in new base new base; do
  if [ $v = base ]; then cp slepc_amd/libksgpu_base.bin slepc_amd/libksgpu.so; else cp /tmp/new.so slepc_amd/libksgpu.so; fi
  echo "=== $v: C1 $(python scripts/c1_trace.py 2>/dev/null | tail -1 | sed 's/.*} //')"
  python bench.py --no-cpu-baseline --no-c5 --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('    C3 %.1f steps/s | C2 %.1f' % (d['value'], d['configs']['C2']['value']))
"
done
cp /tmp/new.so slepc_amd/libksgpu.so
