# PMC comparison of the ball-query query kernels (GPU box): S4G_BQ_MODE / S4G_BQ_G variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c/pmc
rm -rf $O; mkdir -p $O
for v in "grid 2" "lean 1" "lean 2"; do set -- $v
  export S4G_BQ_MODE=$1 S4G_BQ_G=$2
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/$1$2 -- python3 $R/tools/bench_ops.py --ops ball1 > /dev/null 2>&1
  echo "== $1 G=$2"
  python3 $R/tools/pmc_summary.py $(find $O/$1$2 -name "*counter_collection.csv") | grep -A1 "query_kernel"
done
