set -x
mkdir -p gpurun_out/r2d
for i in 1 2; do
  rm -rf /tmp/dpp$i; mkdir -p /tmp/dpp$i
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 2950$i tests/probes/dp_two_ranks_one_gpu.py /tmp/dpp$i > gpurun_out/r2d/probe$i.log 2>&1
  python - <<PY >> gpurun_out/r2d/probe$i.log 2>&1
import torch
for r in (0,1):
    d = torch.load("/tmp/dpp$i/rank%d.pt" % r)
    print(r, {k: (v if k not in ("bounds",) else len(v)) for k, v in d.items()})
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -x -k "aggregate or interaction_module or full_model or four_cells" > gpurun_out/r2d/k8.log 2>&1
tail -3 gpurun_out/r2d/k8.log; tail -4 gpurun_out/r2d/probe1.log; tail -4 gpurun_out/r2d/probe2.log
