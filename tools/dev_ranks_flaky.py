"""dev: do bench.py's two-rank tiny runs (tests/test_bench_ranks.py: two ranks sharing one GPU) return the codes of the
one-rank run, run after run?  usage: python tools/dev_ranks_flaky.py N_RUNS [ENV=VALUE ...]
(How the packed-fp32 fault behind -fno-slp-vectorize was found: profiles/README.md; the probes used are kept as
tools/dev_cotenancy_probe.patch.)"""
import os
import subprocess
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_runs = int(sys.argv[1])
env = dict(os.environ, **dict(a.split("=", 1) for a in sys.argv[2:]))


def run(g, tag):
    out = f"/tmp/codes_{tag}.npy"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(g), "--steps", "1", "--warmup", "0", "--frames", "10",
           "--global-batch", "6", "--no-cpu-baseline", "--dump-codes", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


ref = run(1, "one")
bad = 0
for i in range(n_runs):
    c = run(2, f"two_{i}")
    d = np.argwhere(c != ref)
    if len(d):
        bad += 1
        print("two-rank run", i, "mismatches:", len(d), [(tuple(int(v) for v in x), int(ref[tuple(x)]), int(c[tuple(x)])) for x in d[:8]], flush=True)
print("two-rank runs that differ from the one-rank run:", bad, "of", n_runs, flush=True)
