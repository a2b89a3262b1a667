# timeline of one un-pipelined forward pass: tools/trace_one_batch.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/one_batch; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-pipeline --timer-every 100000 --no-cpu-baseline --no-extras "$@" > $O/bench.json 2> $O/err.txt
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocpd_one_batch.py $db > $O/timeline.txt
find $O -name "*.db" -delete
cat $O/timeline.txt
