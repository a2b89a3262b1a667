python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ball or query or fmad or pipeline or random" 2>&1 | tail -3
for c in 1 2; do python tools/bench_ops.py --ops ball1,qgroup,group 2>/dev/null; done
