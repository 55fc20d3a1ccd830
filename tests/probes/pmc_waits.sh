# wave-cycle breakdown per kernel over the bench step: one rocprofv3 --pmc pass (kernel trace only), summarised per kernel name
O=gpurun_out/pmc_waits; mkdir -p $O
F="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-leg --no-alt-leg --no-host-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/run -o p -- python3 bench.py $F > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 11; }
python3 - "$(ls $O/run/*/p_counter_collection.csv $O/run/p_counter_collection.csv 2>/dev/null | head -1)" > $O/summary.log <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    key = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:78]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
print("# fractions of SQ_WAVE_CYCLES per kernel, summed over the dispatches of 3 forward+backward passes (two branch streams on), sorted by wave cycles")
print("%-80s %10s %8s %8s %8s %8s %8s" % ("kernel", "wave cyc", "wait", "w.inst", "w.lds", "active", "mfma/busy"))
for k, c in rows[:22]:
    w = c["SQ_WAVE_CYCLES"] or 1.0
    print("%-80s %10.3e %8.3f %8.3f %8.3f %8.3f %8.3f" % (k, w, c["SQ_WAIT_ANY"] / w, c["SQ_WAIT_INST_ANY"] / w, c["SQ_WAIT_INST_LDS"] / w, c["SQ_ACTIVE_INST_ANY"] / w,
                                                   c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] or 1.0)))
PY
rm -rf $O/run
cat $O/summary.log
