# A/B of several builds of the library in ONE box: tools/ab_libs.sh "<bench args>" lib1[@VAR=v,...] lib2 ...
# ("default" = the shipped library); interleaved rounds, prints value / contraction ms / the big launches
args=$1; shift
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for lib in "$@"; do
    # an entry is <lib>[@VAR=value[,VAR=value...]]: the environment assignments apply to that run only
    l=${lib%%@*}; envs=; [ "$lib" != "$l" ] && envs=$(echo ${lib#*@} | tr ',' ' ')
    if [ $l = default ]; then l=; fi
    env S4G_HIP_LIB=$l $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
g=lambda s:[v['ms'] for n,v in k.items() if s in n]
print('%-34s r$round: %7.1f sc/s %6.3f ms/step contr %6.3f frac %.4f heads %.3f sa0 %.3f sa1 %.3f sa2 %.3f fp1 %.3f tiled %.3f fps %.3f' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline']['frac'], g('heads')[0], g('sa0.1')[0], g('sa1.1')[0], g('sa2.1')[0], g('fp1.1')[0], sum(v['ms'] for n,v in k.items() if n.startswith('gemm[') and '+' not in n), max(v['ms'] for n,v in k.items() if n.startswith('fps['))))
"
  done
done 2>&1 | tee gpurun_out/ab/last.txt
