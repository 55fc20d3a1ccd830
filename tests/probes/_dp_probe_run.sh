# Repeats the two-rank data-parallel rehearsal (gloo, both ranks on cuda:0) and prints each run's comparison of the plain and the
# overlapped gradient all-reduce.  Usage on the GPU box: bash tests/probes/_dp_probe_run.sh <out-dir-under-gpurun_out> [runs]
OUT=gpurun_out/${1:-dp}
mkdir -p $OUT
export D2R_DIST_BACKEND=gloo
for i in $(seq 1 ${2:-3}); do
  rm -rf /tmp/dpp$i; mkdir -p /tmp/dpp$i
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2950$i tests/probes/dp_two_ranks_one_gpu.py /tmp/dpp$i > $OUT/probe$i.log 2>&1 || { echo "run $i failed"; tail -5 $OUT/probe$i.log; break; }
  python - <<PY >> $OUT/probe$i.log 2>&1
import torch
for r in (0,1):
    d = torch.load("/tmp/dpp$i/rank%d.pt" % r)
    print(r, {k: (v if k not in ("bounds",) else len(v)) for k, v in d.items()})
PY
  tail -2 $OUT/probe$i.log | cut -c1-300; grep -o "first_grad_divergence.: .\{0,3000\}" $OUT/probe$i.log | head -1
done
