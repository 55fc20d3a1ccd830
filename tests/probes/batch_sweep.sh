# the timed step at per-GPU batch 16 / 32 / 64 / 128 (C2 geometry otherwise): samples/s, ms/step, fwd+bwd-only ms
F="--steps 20 --warmup 4 --no-cpu-baseline --no-fp32-leg --no-alt-leg --no-host-leg --no-roofline"
for b in 16 32 64 128; do timeout -k 10 300 python bench.py $F --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batch $b:', d['value'], 'samples/s', d['ms_per_step'], 'ms/step', d['fwd_bwd_only']['ms_per_step_per_rank'], 'ms fwd+bwd')"; done
