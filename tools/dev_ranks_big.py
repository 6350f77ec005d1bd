"""dev: the two-rank rehearsal of bench.py at a larger size (32 utterances, 120 frames): the codes of repeated two-rank runs are
compared with each other, bit for bit (two processes share the one card, which is where the packed-fp32 fault showed; see
profiles/README.md).  The one-rank run of the same 32 utterances is printed beside them as information only: it decodes all 32 in
one batch, and above 16 utterances the decode step runs other kernel forms (csrc/dec_linear.hip valle_dec_wave_linear: K = d_model
summed on one wave instead of two; csrc/attn_decode.hip: 6-chunk passes instead of 12), whose fp32 sums come in another order.  On
random weights under greedy sampling a last-bit difference flips a near tie within some tens of frames and every later code
follows, so that comparison is not expected to be equal; tests/test_bench_ranks.py holds the equality at sizes where both sides
run the same forms."""
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
one = run(1, "one")
first = run(2, "two_0")
d = np.argwhere(first != one)
print("one rank against two ranks (other kernel forms above 16 utterances):", len(d), "of", one.size, "codes differ, first at",
      [tuple(int(v) for v in x) for x in d[:1]], flush=True)
bad = 0
for i in range(1, 5):
    c = run(2, f"two_{i}")
    d = np.argwhere(c != first)
    bad += len(d) > 0
    print("two-rank run", i, "against two-rank run 0:", len(d), "of", c.size, "codes differ",
          [tuple(int(v) for v in x) for x in d[:5]], flush=True)
print("done:", bad, "of 4 repeated two-rank runs differ from the first", flush=True)
sys.exit(1 if bad else 0)
