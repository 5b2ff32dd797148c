"""Development aid: run scripts/gpu_cache_sweep.py's workloads under several settings of the RT_WF_* knobs.
usage: python scripts/gpu_knobs.py "RT_WF_RAYS_PER_WAVE=16" "RT_WF_COOP=1 RT_WF_CACHE_NODES=0" ..."""
import os, subprocess, sys
for setting in sys.argv[1:] or [""]:
    print("===", setting or "(defaults)", flush=True)
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=")
        env[k] = v
    subprocess.run([sys.executable, "-u", "scripts/gpu_cache_sweep.py", "child"], env=env, check=True, timeout=300)
