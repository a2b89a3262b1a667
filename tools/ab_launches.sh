# per-launch times of the contraction launches for several environments in ONE box:
# tools/ab_launches.sh "<bench args>" "VAR=v ..." "VAR=v ..."   ("-" = no overrides)
args=$1; shift
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('== $envs : %.1f scenes/s  %.3f ms/step  contraction %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_step']))
for n,v in sorted(d['kernels'].items()):
    if n.startswith('gemm['): print('   %-70s %.4f ms  %7.1f TF' % (n, v['ms'], v['TFLOPs']))
"
done
