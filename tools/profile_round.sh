#!/bin/bash
# The evidence set of a round, in one go (run on the GPU box from the repo root):  tools/profile_round.sh r03a
#   gpurun_out/<tag>_bench.json            the default `python bench.py` line
#   gpurun_out/<tag>_stats/                rocprofv3 --kernel-trace --stats of the headline loop (kernel averages)
#   gpurun_out/<tag>_kbench_B32/_B256_*    rocprofv3 --kernel-trace --stats of the multislice launch alone (tools/kbench.py)
#   gpurun_out/<tag>_pmc_B32, _B256/       five --pmc passes each over tools/kbench.py (+ summary.json); traffic_latest.json refreshed
#   gpurun_out/<tag>_timeline_*.txt        kernel timelines of the B=32 step, the per-angle step and the 16-virtual-rank step
#   gpurun_out/<tag>_rows.json             tools/bench_rows.py: config-2 / config-1 / config-5 shapes
#   gpurun_out/<tag>_bench_host2.json      `python bench.py --gpus 2 --comm host`: the self-launched 2-rank line on ONE GPU (three legs + immediate_p2p)
#   gpurun_out/<tag>_bench_p2p{2,4}.json   `python bench.py --gpus N --comm p2p`: the direct exchange, N ranks sharing ONE GPU
#   gpurun_out/<tag>_p2p2_stats/           rocprofv3 --kernel-trace --stats of RANK 0 of a 2-rank p2p run (rank 1 beside it, unprofiled)
#   gpurun_out/<tag>_rows_{c1,c5}_stats/   rocprofv3 --stats of the config-1 / config-5 rows of tools/bench_rows.py
# Counters are collected in their own runs (no --pmc together with trace domains other than --kernel-trace).
set -u
TAG=${1:-r03}
OUT=gpurun_out
ROOT=$(pwd)
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-driver --no-per-angle > $ROOT/$OUT/${TAG}_stats.log 2>&1 )
python tools/kstats.py $OUT/${TAG}_stats > $OUT/${TAG}_kernel_stats.txt 2>&1
for B in 32 256; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_kbench_B${B}_stats -- python3 $ROOT/tools/kbench.py $B 12 > $ROOT/$OUT/${TAG}_kbench_B$B.log 2>&1 )
  python tools/kstats.py $OUT/${TAG}_kbench_B${B}_stats > $OUT/${TAG}_kbench_B${B}_kernel_stats.txt 2>&1
  bash tools/pmc_sq.sh $OUT/${TAG}_pmc_B$B $B > $OUT/${TAG}_pmc_B$B.log 2>&1
done
python tools/traffic_update.py $OUT/${TAG}_pmc_B32 32 "profiles/${TAG:0:3}/${TAG}_pmc_B32.json" > $OUT/${TAG}_traffic.log 2>&1
for LEG in none per_angle vr16; do
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/${TAG}_tr_$LEG -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-driver --legs $LEG > /dev/null 2>&1 )
  f=$(find $OUT/${TAG}_tr_$LEG -name "*kernel_trace.csv" | head -1)
  python tools/trace_tail.py $f 40 > $OUT/${TAG}_timeline_$LEG.txt 2>&1
done
python tools/bench_rows.py > $OUT/${TAG}_rows.json 2>/dev/null
for R in c1 c5; do
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_rows_${R}_stats -- python3 $ROOT/tools/bench_rows.py $R > /dev/null 2>&1 )
done
python bench.py --gpus 2 --comm host --steps 3 --warmup 1 > $OUT/${TAG}_bench_host2.json 2> $OUT/${TAG}_bench_host2.err; echo "bench --gpus 2 --comm host rc=$?"
for N in 2 4; do
  python bench.py --gpus $N --comm p2p --steps 10 --warmup 3 > $OUT/${TAG}_bench_p2p$N.json 2> $OUT/${TAG}_bench_p2p$N.err; echo "bench --gpus $N --comm p2p rc=$?"
done
# rank 0 of a 2-rank p2p run under the profiler, rank 1 beside it as a plain process (both are ranks of an external "launcher": this
# shell; the profiled program follows "--" directly, no hop)
PORT=$((20000 + RANDOM % 20000))
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT ADM_RDV_PORT=$PORT ADM_RDV_JOB=prof$PORT HSA_ENABLE_IPC_MODE_LEGACY=0
( RANK=1 LOCAL_RANK=1 python3 bench.py --gpus 2 --comm p2p --steps 10 --warmup 3 --no-per-angle > /dev/null 2>&1 ) &
( cd /tmp && RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_p2p2_stats -- python3 $ROOT/bench.py --gpus 2 --comm p2p --steps 10 --warmup 3 --no-per-angle > $ROOT/$OUT/${TAG}_p2p2_rank0.json 2> $ROOT/$OUT/${TAG}_p2p2_rank0.err )
wait
unset WORLD_SIZE MASTER_ADDR MASTER_PORT ADM_RDV_PORT ADM_RDV_JOB
python tools/kstats.py $OUT/${TAG}_p2p2_stats > $OUT/${TAG}_p2p2_kernel_stats.txt 2>&1
python tools/bench_brief.py $OUT/${TAG}_bench.json
cat $OUT/${TAG}_kernel_stats.txt | head -20
cat $OUT/${TAG}_traffic.log | head -12
