#!/bin/bash
# Round-6 profile capture on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats of the default bench and of configs[4], then separate PMC passes
#   (FETCH_SIZE / WRITE_SIZE cannot share a pass) for the contraction launches of one step and
#   for the ball_query + group_points operator pair.  Outputs land in gpurun_out/r6prof/;
#   S4G_PROFILE_ROUND=6 tools/publish_profiles.py (which runs tools/make_traffic_json.py) turns them into profiles/r06_*.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG4="--points 51200 --batch 32 --precision bf16"
NOPIPE="--steps 2 --warmup 1 --no-pipeline --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/stats_default -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_default_profiled.json 2> $O/err_default.txt
rocprofv3 --kernel-trace --stats -d $O/stats_cfg4 -o run -- python3 $R/bench.py $CFG4 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_cfg4_profiled.json 2> $O/err_cfg4.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_default_$c -- python3 $R/bench.py $NOPIPE > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_cfg4_$c -- python3 $R/bench.py $CFG4 $NOPIPE > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_ops_$c -- python3 $R/tools/bench_ops.py --ops ball,group,qgroup > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_default_SQ -- python3 $R/bench.py $NOPIPE > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_cfg4_SQ -- python3 $R/bench.py $CFG4 $NOPIPE > /dev/null 2>&1
# effective shader clock per kernel: GRBM_GUI_ACTIVE / wall time (MI355X_MICROARCH.md, "DVFS give-back")
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_default_CLK -- python3 $R/bench.py $NOPIPE > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_cfg4_CLK -- python3 $R/bench.py $CFG4 $NOPIPE > /dev/null 2>&1
for d in default cfg4; do
  db=$(find $O/stats_$d -name "*.db" | head -1)
  [ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db 45 > $O/${d}_kernel_stats.md
done
# the reference-shaped modules on the HIP operators (INTEGRATION.md levels 1-2): the LAST pass of a kernel trace
rocprofv3 --kernel-trace --stats -d $O/stats_modules -o run -- python3 $R/bench.py --impl modules --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_modules_profiled.json 2> $O/err_modules.txt
db=$(find $O/stats_modules -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_last_pass.py $db fps_cell_sort 40 > $O/modules_last_pass.md
# the attainable MFMA rate of the board this capture ran on, and power / clock of every contraction launch alone
$R/tools/micro/mfma_ceiling 1.5 > $O/mfma_ceiling.md 2>&1
python3 $R/tools/power_probe.py --seconds 1.5 > $O/power_clock_default.md 2>/dev/null
python3 $R/tools/power_probe.py --seconds 1.5 --weights randomized > $O/power_clock_randomized.md 2>/dev/null
python3 $R/tools/power_probe.py --seconds 1.5 --points 51200 --batch 32 --precision bf16 > $O/power_clock_cfg4.md 2>/dev/null
# rounds 1-5's network (randomize_bn_) on the same build: kernel stats for the headline-vs-old comparison
rocprofv3 --kernel-trace --stats -d $O/stats_randomized -o run -- python3 $R/bench.py --weights randomized --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_randomized_profiled.json 2> $O/err_randomized.txt
db=$(find $O/stats_randomized -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db 45 > $O/randomized_kernel_stats.md
# the production entry: 25 pipelined GraspDetector steps (tools/detect_loop.py)
rocprofv3 --kernel-trace --stats -d $O/stats_detect -o run -- python3 $R/tools/detect_loop.py 25 > $O/detect_loop.txt 2> $O/err_detect.txt
db=$(find $O/stats_detect -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db 40 > $O/detect_kernel_stats.md
# un-profiled lines of the same build, for the record
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_default.json 2>/dev/null
python3 $R/bench.py $CFG4 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2>/dev/null
# keep only what the summaries need (the raw trace databases are large)
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls -la $O
