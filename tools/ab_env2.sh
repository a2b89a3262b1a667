# A/B of environment knobs on the default bench: tools/ab_env2.sh "A=1" "A=0" ...   (two rounds)
for rep in 1 2; do for kv in "$@"; do
  echo "== $kv"
  env $kv python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dense', d['roofline']['ms_per_step'], 'one batch', d['latency']['latency_ms_one_batch'], 'b1', d['latency']['latency_ms_b1'])"
done; done
