for rep in 1 2; do for a in "--in-flight 2" "--in-flight 1" "--in-flight 3"; do
 for g in 2 1; do echo "== $a geo=$g"; S4G_GEO_STREAMS=$g python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs4 --no-extras $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'dense', d['roofline']['ms_per_step'])"; done; done; done
