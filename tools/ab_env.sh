# A/B of one environment switch on ONE box, interleaved: tools/ab_env.sh VAR=a VAR=b [bench args...]
a=$1; b=$2; shift 2
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for v in "$a" "$b"; do
    env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
big=' '.join('%s=%.3f' % (n[5:].split(' ')[0][:22], v['ms']) for n,v in k.items() if n.startswith('gemm[') and v['ms'] > 0.3)
print('$v round $round: %.1f scenes/s  %.3f ms/step  median %.3f  contraction %.3f ms  frac %.4f | %s' % (d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['ms_per_step'], d['roofline']['frac'], big))
"
  done
done 2>&1 | tee gpurun_out/ab/last_env.txt
