cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/vall-e_amd:$LD_LIBRARY_PATH
for ROLE in gu qkv down; do
export MICRO_GEMM_ONLY=$ROLE
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pg_$ROLE -- $R/tools/micro_gemm > /tmp/pg_$ROLE.log 2>&1 || { tail -5 /tmp/pg_$ROLE.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(glob.glob("/tmp/pg_$ROLE/*/*_counter_collection.csv")[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "TCC_HIT_sum": n[k] += 1
for k, c in acc.items():
    if "gemm" in k: print("$ROLE", k, {x: round(v / n[k]) for x, v in c.items()})
PY
done
