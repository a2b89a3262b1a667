python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps" 2>&1 | tail -3
for v in 2 4 2 4; do echo "== S4G_FPS_SPEC=$v"; S4G_FPS_SPEC=$v python tools/bench_ops.py --ops fps,fps51k 2>/dev/null; S4G_FPS_SPEC=$v python tools/bench_ops.py --ops fps --batch 1 2>/dev/null | head -1; done
