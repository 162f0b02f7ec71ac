#!/usr/bin/env python3
"""Same-box sweep of the developer switches: every variant twice, alternating with the default, median step time of a
30-step run each (bench.py's step_ms_min_median_max; the garbage collector is frozen, so runs agree to ~0.05 ms).

    python tools/sweep_switches.py [variant ...]        # variant = NAME=VALUE[,NAME=VALUE]; no arguments: the built-in list
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["UCLSTM_P3_MIN_BLOCKS=64", "UCLSTM_P3_MIN_BLOCKS=96", "UCLSTM_P3_MIN_BLOCKS=160", "UCLSTM_P3_MIN_BLOCKS=192", "UCLSTM_P3_MIN_BLOCKS=256",
            "UCLSTM_P3_LEAD=6", "UCLSTM_PACK_SEGMENTS=", "UCLSTM_PACK_SEGMENTS=0.5", "UCLSTM_BN_BWD_BLOCKS=2048", "UCLSTM_BN_BWD_BLOCKS=4096",
            "UCLSTM_HOIST_X=1", "UCLSTM_PARAM_GRADS_ON_SIDE=0", "UCLSTM_BN_RUNNING_ON_SIDE=0", "UCLSTM_WGRAD_NO192=1", "UCLSTM_WGRAD_RING=0",
            "UCLSTM_POOL_SKIP=0"]


def run(env_add):
    env = dict(os.environ)
    for kv in env_add.split(","):
        if kv:
            k, v = kv.split("=", 1)
            env[k] = v
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "8", "--no-roofline"], env=env,
                         capture_output=True, text=True, timeout=300)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return d["step_ms_min_median_max"][1], d["ms_per_step"]


variants = sys.argv[1:] or VARIANTS
base = []
for v in variants:
    b = run("")
    base.append(b[0])
    a1 = run(v)
    a2 = run(v)
    print(f"{v:34s} median step {a1[0]:.2f} / {a2[0]:.2f} ms   (default just before: {b[0]:.2f}; means {a1[1]:.2f} / {a2[1]:.2f} vs {b[1]:.2f})", flush=True)
print(f"default: {min(base):.2f} .. {max(base):.2f} ms over {len(base)} runs")
