#!/bin/bash
# Round 6, VERDICT item 3a: the five single-layer launches of a step (sa1.0f|fp1.0d, sa2.0f|fp0.0d, fp0.0s, fp0.1, fp1.0s) under
# the launch forms the library has -- tiled kernel everywhere / chain-form kernel everywhere / the shipped per-shape choice --
# and the whole step with two contraction streams.  Writes gpurun_out/ab_short_*.json (bench lines).
set -u
mkdir -p gpurun_out
export S4G_TEST_KNOBS=1
for v in default 0 1; do
  if [ "$v" = default ]; then unset S4G_GEMM_SINGLE_CHAIN; else export S4G_GEMM_SINGLE_CHAIN=$v; fi
  python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/ab_short_chain_$v.json 2> gpurun_out/ab_short_chain_$v.err
done
unset S4G_GEMM_SINGLE_CHAIN
S4G_DENSE_STREAMS=2 python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/ab_short_dense2.json 2> gpurun_out/ab_short_dense2.err
python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/ab_short_default_again.json 2> /dev/null
