# kernel-trace stats of tools/bench_ops.py for the given ops: tools/trace_ops.sh nn
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_ops; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -o run -- python3 $R/tools/bench_ops.py --ops "$1" > $O/out.txt 2>/dev/null
db=$(find $O/t -name "*.db" | head -1); python3 $R/tools/rocpd_summary.py $db 16 > $O/stats.md; find $O -name "*.db" -delete
cat $O/out.txt; cat $O/stats.md
