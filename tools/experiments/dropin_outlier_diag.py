import sys, os, gc, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import dropin_driver as D
mode = sys.argv[1]
if mode == "nogc":
    gc.disable()
if mode == "gcstats":
    gc.callbacks.append(lambda phase, info: print("gc", phase, info, time.perf_counter()) if info.get("generation", 0) == 2 else None)
r = D.run("C2", steps=160, warmup=10, timers=False)
print(mode, r["ms_per_step"], r["median_ms_per_step"], r["max_ms_per_step"], r["slowest_steps"])
