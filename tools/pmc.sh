#!/bin/bash
# usage: tools/pmc.sh TAG -- HBM traffic counters of the decode layer chain (tools/micro_dec, eager), one counter per pass;
# writes gpurun_out/TAG_pmc_decode_layer.json (FETCH_SIZE x2: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md;
# calibrated in the same run on a pure stream of known size)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$1_$C -- $R/tools/micro_dec 16 536 0 1 > $R/gpurun_out/pmc_$1_$C.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections, json, re
acc = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/pmc_$1_%s/*/*_counter_collection.csv" % C)[0]
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != C: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", ""); a[k][0] += 1; a[k][1] += float(r["Counter_Value"])
    acc[C] = a
def per(C, k): n, v = acc[C][k]; return v / n * 1024.0      # the counters are in KB
cal = next(k for k in acc["FETCH_SIZE"] if k.startswith("stream2_kernel<98304, 256>"))
raw = per("FETCH_SIZE", cal)
ctx, B = 537, 16
alg = {"qkv": 3 * 1024 * 1024 * 2, "attn": B * 16 * ctx * 256, "o": 1024 * 1024 * 2, "gu": 2 * 4096 * 1024 * 2, "down": 4096 * 1024 * 2}
def role(k):
    if k.startswith("attn_decode"): return "attn"
    m = re.match(r"dec_wave_kernel<(\d+), (\d+), (\d+)", k)
    if not m: return None
    e, st, nw = map(int, m.groups())
    return {10: "qkv", 2: "gu"}.get(e) or ("down" if e == 1 and st * nw * 64 * 4 == 4096 else "o" if e == 1 else None)
rows, tot_m, tot_a = [], 0.0, 0.0
for k in acc["FETCH_SIZE"]:
    r = role(k)
    if r is None: continue
    fb, wb = per("FETCH_SIZE", k) * 2.0, per("WRITE_SIZE", k)
    rows.append({"kernel": k, "role": r, "launches": acc["FETCH_SIZE"][k][0], "fetch_bytes": int(fb), "write_bytes": int(wb),
                 "algorithmic_bytes": alg[r], "ratio": round((fb + wb) / alg[r], 3)})
    tot_m += fb + wb; tot_a += alg[r]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/micro_dec 16 536 0 1 (eager), MI355X; tools/pmc.sh $1",
       "calibration": {"kernel": cal + " reads 25,165,824 B", "raw_fetch_bytes": int(raw), "corrected": int(raw * 2),
                       "note": "raw = 1/2 of the bytes: x2 correction applied to FETCH_SIZE"},
       "ctx": ctx, "B": B, "per_layer": sorted(rows, key=lambda r: ["qkv", "attn", "o", "gu", "down"].index(r["role"])),
       "layer_measured_bytes": int(tot_m), "layer_algorithmic_bytes": int(tot_a), "layer_traffic_over_algorithmic": round(tot_m / tot_a, 4)}
json.dump(out, open("$R/gpurun_out/$1_pmc_decode_layer.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
