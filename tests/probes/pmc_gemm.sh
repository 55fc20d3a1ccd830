cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcg
i=0
for shape in "NT 6304 768 768 1" "NT 6304 3072 768 1" "NT 6304 3072 768 3" "NN 6304 768 3072 5" "TN 3072 768 6304 3"; do
  i=$((i+1))
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmcg/a$i -o a -- python3 tests/probes/gemm_one.py $shape > gpurun_out/pmcg/a$i.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --kernel-trace --output-format csv -d gpurun_out/pmcg/b$i -o b -- python3 tests/probes/gemm_one.py $shape > gpurun_out/pmcg/b$i.log 2>&1
  echo "$shape" > gpurun_out/pmcg/shape$i.txt
done
rm -f gpurun_out/pmcg/*/*_kernel_trace.csv
ls gpurun_out/pmcg
