# LDS bank conflicts per kernel over the bench step: one rocprofv3 --pmc pass (kernel trace only), summarised per kernel name
O=gpurun_out/pmc_lds; mkdir -p $O
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-alt-leg --no-host-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/run -o p -- python3 bench.py $F > $O/run.log 2>&1 || exit 11
python3 - "$(ls $O/run/*/p_counter_collection.csv $O/run/p_counter_collection.csv 2>/dev/null | head -1)" > $O/summary.log <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]; c = r["Counter_Name"]; v = float(r["Counter_Value"])
    key = re.sub(r"\(anonymous namespace\)::", "", n)[:90]
    if c == "SQ_LDS_BANK_CONFLICT": agg[key][0] += v
    elif c == "SQ_LDS_IDX_ACTIVE": agg[key][1] += v
    d = (r.get("Dispatch_Id"), c)
    if c == "SQ_LDS_IDX_ACTIVE": agg[key][2] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel (summed over the dispatches of 3 forward+backward passes), sorted by LDS active cycles")
for k, (bc, act, n) in rows[:24]:
    print("%-92s dispatches %5d  LDS active %.3e  conflict %.3e  = %5.1f %%" % (k, n, act, bc, 100.0 * bc / act if act else 0.0))
PY
rm -rf $O/run
cat $O/summary.log
