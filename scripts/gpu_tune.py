"""Development aid: A/B of the wavefront tuning knobs on one box (fresh context per setting)."""
import os, subprocess, sys
code = "import sys; sys.path.insert(0,'.'); from scripts.gpu_perf import run; run('room',1920,1080); run('scene',1920,1080)"
for refill in (64, 32, 20, 8):
    for blocks in (4, 6, 8):
        env = dict(os.environ, RT_WF_REFILL=str(refill), RT_WF_BLOCKS=str(blocks))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout
        print(f"refill {refill} blocks {blocks}")
        for l in out.splitlines():
            if 'per-kernel' in l: print("   ", l.strip())
