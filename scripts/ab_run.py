"""A/B helper: run bench.py's headline against whatever libksgpu.so is in place, tolerating an OLDER build that lacks entry points the current
bindings declare (they are dropped from the binding table; the legs that need them are switched off). Prints one summary line."""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slepc_amd import _lib
so = ctypes.CDLL(_lib.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
missing = [n for n in list(_lib._SIG) if not hasattr(so, n)]
for n in missing:
    del _lib._SIG[n]
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-configs", "--no-pmc"] + sys.argv[1:]
r, w = os.pipe()
saved = os.dup(1)
os.dup2(w, 1)
try:
    bench.main()
finally:
    sys.stdout.flush(); os.dup2(saved, 1); os.close(w)
out = os.read(r, 1 << 22).decode().strip().splitlines()[-1]
d = json.loads(out)
rl = d["roofline"]
print("C3 %.1f steps/s  %.4f ms/step | %s %.1f us frac %.4f | copy %.0f GB/s | missing %d" % (d["value"], d["ms_per_step"], rl["kernel"], rl["avg_launch_us_executed"], rl["frac"], rl.get("measured_copy_GBps", 0), len(missing)))
print("   ", [(c["class"], c["ms_per_step"]) for c in d.get("kernel_classes_untimed_instrumented_pass", [])[:5]])
