mkdir -p gpurun_out/r3c
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ball or query or fmad or pipeline or random" 2>&1 | tail -8 > gpurun_out/r3c/pytest_bq.txt
cat gpurun_out/r3c/pytest_bq.txt
for mode in grid lean; do for g in 1 2 4; do
  [ $mode = grid ] && [ $g != 2 ] && continue
  echo "== mode=$mode G=$g"; S4G_BQ_MODE=$mode S4G_BQ_G=$g python tools/bench_ops.py --ops ball,qgroup,group
done; done 2>&1 | tee gpurun_out/r3c/bq_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_ops -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py --ops ball,qgroup,group > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r3c/prof_ops -name "*.db" | head -1); [ -n "$db" ] && python3 tools/rocpd_summary.py $db 14 > gpurun_out/r3c/ops_kernel_stats.md; find gpurun_out/r3c -name "*.db" -delete
cat gpurun_out/r3c/ops_kernel_stats.md | head -30
