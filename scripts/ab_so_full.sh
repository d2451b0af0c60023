#!/bin/bash
# A/B of two builds of libksgpu.so on one box incl. config 5: new, base, new, base (slepc_amd/libksgpu_base.bin = the other build)
cp slepc_amd/libksgpu.so /tmp/new.so
for v in new base new base; do
  if [ $v = base ]; then cp slepc_amd/libksgpu_base.bin slepc_amd/libksgpu.so; else cp /tmp/new.so slepc_amd/libksgpu.so; fi
  echo "=== $v"
  python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('C3 %.1f steps/s | C2 %.1f | C5 %.2f' % (d['value'], d['configs']['C2']['value'], d['configs']['C5']['value']))
print('   ', [(c['class'], c['ms_per_step']) for c in d['kernel_classes_untimed_instrumented_pass'][:5]])
print('    C5', [(c['class'], c['ms_per_step']) for c in d['configs']['C5']['kernel_classes'][:4]])
"
done
cp /tmp/new.so slepc_amd/libksgpu.so
