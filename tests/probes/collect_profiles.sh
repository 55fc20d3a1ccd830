#!/bin/bash
# Round evidence for profiles/: (1) rocprofv3 --kernel-trace --stats of the bench command with the two branch streams and with one
# stream (every kernel alone on the GPU), (2) per-queue gaps of the traced step, eager and hipGraph replay, (3) FETCH_SIZE /
# WRITE_SIZE counter passes of the same command (separate runs, as MI355X_MICROARCH.md prescribes), (4) the full bench line.
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
echo "[collect] two-stream trace done"
D2R_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o kt -- python3 bench.py --steps 6 --warmup 2 $FLAGS > $OUT/kt1.log 2>&1 || exit 17
cp $(find $OUT/kt1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_stream.csv
rm -rf $OUT/kt1
echo "[collect] one-stream trace done"
# hipGraph replay of the same step (bf16: the captured step carries no loss scaler), and the eager bf16 step beside it
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ktg -o kt -- python3 bench.py --steps 6 --warmup 2 --dtype bf16 --graph 1 $FLAGS > $OUT/ktg.log 2>&1 || exit 18
python3 profiles/make_step_gaps.py $(find $OUT/ktg -name "*kernel_trace.csv" | head -1) $OUT/step_gaps_graph_bf16.json > $OUT/step_gaps_graph.log 2>&1 || echo "[collect] graph gaps failed"
rm -rf $OUT/ktg
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kte -o kt -- python3 bench.py --steps 6 --warmup 2 --dtype bf16 $FLAGS > $OUT/kte.log 2>&1 || exit 19
python3 profiles/make_step_gaps.py $(find $OUT/kte -name "*kernel_trace.csv" | head -1) $OUT/step_gaps_eager_bf16.json > $OUT/step_gaps_eager.log 2>&1 || echo "[collect] eager gaps failed"
rm -rf $OUT/kte
echo "[collect] graph / eager traces done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 2 --warmup 1 $FLAGS > $OUT/fetch.log 2>&1 || exit 12
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 2 --warmup 1 $FLAGS > $OUT/write.log 2>&1 || exit 13
python3 profiles/make_pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1 || exit 14
rm -rf $OUT/fetch $OUT/write
echo "[collect] counter passes done"
timeout -k 10 500 python3 bench.py > $OUT/bench.log 2> $OUT/bench.err || exit 15
tail -1 $OUT/bench.log > $OUT/bench.json
ls -la $OUT
