# A/B of two builds of the library in ONE box: tools/ab_bench.sh <alt-lib> [bench args...]
# interleaved rounds (default build, alternative build), prints value / step / fps / contraction ms
alt=$1; shift
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for v in base alt; do
    # the override is scoped to the one command (never exported: a leftover S4G_HIP_LIB would redirect
    # every later load of the library in this shell)
    if [ $v = alt ]; then lib=$alt; else lib=; fi
    S4G_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
f=[v['ms'] for n,v in k.items() if n.startswith('fps[N=25600')]
print('$v round $round: %.1f scenes/s  %.3f ms/step  median %.3f  contraction %.3f ms  frac %.4f  fps0 %.2f ms' % (d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['ms_per_step'], d['roofline']['frac'], f[0] if f else -1))
"
  done
done 2>&1 | tee gpurun_out/ab/last.txt
