"""bench.py's CPU baseline leg with several OpenMP team sizes (the GPU box shows 256 CPUs but grants a 16-core quota)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for t in (sys.argv[1:] or ["16", "32", "128"]):
    os.environ["BENCH_CPU_THREADS"] = t
    r = bench.cpu_baseline(216)
    print(t, "threads:", json.dumps({k: r[k] for k in ("value", "cores", "seconds")}), "6 steps:", r["same_6_steps_all_cores"]["value"], flush=True)
