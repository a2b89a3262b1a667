# kernel-trace timeline of the default bench without per-launch event pairs: gaps per stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --steps 12 --warmup 3 --timer-every 100000 --no-cpu-baseline --no-extras > $O/bench_traced.json 2> $O/err.txt
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $db 0.6 > $O/timeline.txt
python3 $R/tools/rocpd_summary.py $db 30 > $O/kernel_stats.md
find $O -name "*.db" -delete
cat $O/timeline.txt | head -60
