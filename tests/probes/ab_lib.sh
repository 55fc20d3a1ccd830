# A/B of two builds of the library on ONE box: $1 = the other build (a copy of libd2r_hip.so); alternates new / old twice
F="--steps 40 --warmup 6 --no-cpu-baseline --no-fp32-leg --no-alt-leg --no-host-leg --no-roofline"
cp d2r_amd/libd2r_hip.so /tmp/lib_new.so
run() { timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['fwd_bwd_only']['ms_per_step_per_rank'])"; }
for r in 1 2; do cp /tmp/lib_new.so d2r_amd/libd2r_hip.so; run new; cp $1 d2r_amd/libd2r_hip.so; run old; done
cp /tmp/lib_new.so d2r_amd/libd2r_hip.so
