#!/bin/bash
# usage: scripts/ab.sh "<ENV=1 ...>" ...   -> runs bench.py (150 steps) once per env-setting, prints value + per-class summary
for envs in "$@"; do
  echo "=== env: [$envs]"
  env $envs python bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steps/s %.1f  ms/step %.4f  kernel_ms/step %.4f' % (d['value'], d['ms_per_step'], d['step_traffic']['kernel_ms_per_step']))
for k in d['kernel_classes_untimed_instrumented_pass']:
    print('   %-22s n=%5d ms=%8.2f  ms/step=%.4f  %7.1f GB/s' % (k['class'],k['launches'],k['ms_total'],k['ms_per_step'],k['hbm_GBps']))
"
done
