cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --no-pipeline --no-extras --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2_a -- python3 $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r2_b -- python3 $R/bench.py $ARGS > /dev/null 2>&1
for d in a b; do python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r2_$d -name "*counter_collection.csv") | grep -A1 "mlp_heads\|mlp_chain" ; done
python3 -m pytest $R/tests/test_ops_gpu.py $R/tests/test_postprocess_gpu.py -x -q -k "three_nn or collision or detector or importance" 2>&1 | tail -4
python3 $R/tools/bench_ops.py --ops nn
