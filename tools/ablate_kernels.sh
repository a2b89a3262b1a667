#!/bin/bash
# Measurement experiment (GPU box): the chain / heads kernels with parts removed, per-launch ms from
# bench.py's table.  tools/ablate_kernels.sh chain|heads bits [bits ...]
#   chain bits: 1 no W refill, 2 no LDS operand reads, 4 no panel epilogue, 8 no final epilogue, 16 no loader
#   heads bits: 1 no W refill, 2 no LDS operand reads, 4 no panel epilogue, 8 nor its barriers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
which=$1; shift
if [ $which = chain ]; then obj=mlp_gemm.o; macro=S4G_CHAIN_ABLATE; else obj=mlp_heads.o; macro=S4G_HEADS_ABLATE; fi
# ablated variants are built into their OWN objects and library (the results of such builds are
# garbage): the shipped libs4g_hip.so is never touched, bench.py loads the variant via S4G_HIP_LIB
ABL=$(pwd)/gpurun_out/ablate_build
mkdir -p $ABL
for a in "$@" 0; do
  rm -f $ABL/$obj
  if ! make -C s4g_release_amd/csrc -j8 OBJDIR=$ABL LIB=$ABL/libs4g_hip_ablate.so HIPFLAGS_EXTRA=-D$macro=$a > $ABL/make.log 2>&1; then
    echo "ablate build $macro=$a failed, see $ABL/make.log" >&2; exit 1
  fi
  for cfg in "" "--points 51200 --batch 32 --precision bf16"; do
    S4G_HIP_LIB=$ABL/libs4g_hip_ablate.so python bench.py $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
want=('heads','sa0.1','sa1.1','sa2.1','fp2.1')
k=['%s %.3f' % (n.split('[')[1].split(' ')[0], v['ms']) for n,v in d['kernels'].items() if any(w in n for w in want)]
print('$which ablate=$a', '${cfg:+cfg4}' or 'default', '  '.join(k))
" | tee -a gpurun_out/ablate_$which.txt
  done
done
