"""dev: the two-rank rehearsal of bench.py at a larger size (32 utterances, 120 frames): codes of repeated runs compared"""
import os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(g, tag):
    out = f"/tmp/big_{tag}.npy"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(g), "--steps", "1", "--warmup", "0", "--frames", "120",
           "--global-batch", "32", "--no-cpu-baseline", "--dump-codes", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)
ref = run(1, "one")
for i in range(5):
    c = run(2, f"two_{i}")
    d = np.argwhere(c != ref)
    print("two-rank run", i, "codes that differ from the one-rank run:", len(d), "of", c.size, [tuple(int(v) for v in x) for x in d[:5]], flush=True)
