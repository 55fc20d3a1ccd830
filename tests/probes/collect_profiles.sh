#!/bin/bash
# Round evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) FETCH_SIZE / WRITE_SIZE counter
# passes of the same command (separate runs, as MI355X_MICROARCH.md prescribes), (3) the full bench line.
# Usage on the GPU box: bash tests/probes/collect_profiles.sh <out-dir-under-gpurun_out>
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p $OUT
FLAGS="--no-cpu-baseline --no-roofline --no-fp32-leg --no-alt-leg --no-host-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 6 --warmup 2 $FLAGS > $OUT/kt.log 2>&1 || exit 11
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 profiles/make_step_gaps.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) $OUT/step_gaps.json > $OUT/step_gaps.log 2>&1 || exit 16
cp $(find $OUT/kt -name "*domain_stats.csv" | head -1) $OUT/domain_stats.csv 2>/dev/null
rm -rf $OUT/kt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 2 --warmup 1 $FLAGS > $OUT/fetch.log 2>&1 || exit 12
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 2 --warmup 1 $FLAGS > $OUT/write.log 2>&1 || exit 13
python3 profiles/make_pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1 || exit 14
cp $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $OUT/fetch_counters.csv; cp $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/write_counters.csv
rm -rf $OUT/fetch $OUT/write
timeout -k 10 400 python3 bench.py > $OUT/bench.log 2>&1 || exit 15
tail -1 $OUT/bench.log > $OUT/bench.json
ls -la $OUT
