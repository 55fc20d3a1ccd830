# usage: bash tests/probes/ab_tuning.sh CODE [CODE ...]: the timed step under d2r_gemm_tuning codes, two rounds each
F="--steps 40 --warmup 6 --no-cpu-baseline --no-fp32-leg --no-alt-leg --no-host-leg --no-roofline"
for r in 1 2; do for t in "$@"; do timeout -k 10 200 python bench.py $F --tuning $t 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t', d['value'], d['ms_per_step'], d['fwd_bwd_only']['ms_per_step_per_rank'], d['final_loss'])"; done; done
