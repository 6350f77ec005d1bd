"""Dev: idle-gap analysis of a rocprofv3 --kernel-trace CSV (which kernels leave the GPU idle, and for how long).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [min_gap_us]"""
import csv, glob, sys, collections

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
f = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed step = the second half of the trace by kernel count of the decode attention
att = [i for i, r in enumerate(rows) if "attn_decode" in r[2]]
half = att[len(att) // 2]
# start at the dense prefill before it: walk back to the previous embed_sum
start = half
while start > 0 and "embed_sum" not in rows[start][2]:
    start -= 1
rows = rows[start:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0
gaps = collections.Counter()
gapn = collections.Counter()
big = []
end = rows[0][1]
busy += rows[0][1] - rows[0][0]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    g = s - end
    if g > 0:
        key = (rows[i - 1][2][:40], n[:40])
        gaps[key] += g
        gapn[key] += 1
        if g > min_gap * 1e3:
            big.append((g / 1e3, rows[i - 1][2][:50], n[:50], (s - t0) / 1e6))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print(f"span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, kernels {len(rows)}")
print("largest idle totals by (previous kernel -> next kernel):")
for k, v in gaps.most_common(14):
    print(f"  {v / 1e6:7.3f} ms in {gapn[k]:6d} gaps (avg {v / gapn[k] / 1e3:6.2f} us)  {k[0]} -> {k[1]}")
print(f"gaps above {min_gap} us: {len(big)}, total {sum(b[0] for b in big) / 1e3:.2f} ms")
for b in sorted(big, reverse=True)[:12]:
    print(f"  {b[0]:8.1f} us at +{b[3]:7.2f} ms  {b[1]} -> {b[2]}")
