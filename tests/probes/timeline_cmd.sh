set -e
O=gpurun_out/tl; mkdir -p $O
F="--no-cpu-baseline --no-roofline --no-fp32-leg --no-alt-leg --no-host-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 6 --warmup 2 $F > $O/kt.log 2>&1
python3 profiles/make_step_timeline.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) $O/step_timeline.json $O/last_step.txt > $O/timeline.log
rm -rf $O/kt
