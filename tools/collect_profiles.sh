#!/bin/bash
# Copy the judged files of an evidence set (tools/profile_round.sh TAG, merged back under gpurun_out/) into profiles/<round>/:
#   tools/collect_profiles.sh r05p        ->  profiles/r05/r05p_*
set -u
TAG=${1:?tag}; SRC=gpurun_out; DST=profiles/${TAG:0:3}
mkdir -p $DST
cp $SRC/${TAG}_bench.json $DST/${TAG}_bench.json
cp $SRC/${TAG}_rows.json $DST/${TAG}_bench_rows.json
cp $SRC/${TAG}_bench_host2.json $DST/${TAG}_bench_host2_selflaunched.json
cp $(find $SRC/${TAG}_stats -name "*kernel_stats.csv" | head -1) $DST/${TAG}_bench_immediate_B32_kernel_stats.csv
for B in 32 256; do
  cp $(find $SRC/${TAG}_kbench_B${B}_stats -name "*kernel_stats.csv" | head -1) $DST/${TAG}_kbench_B${B}_kernel_stats.csv
  cp $SRC/${TAG}_pmc_B$B/summary.json $DST/${TAG}_pmc_B$B.json
done
for N in 2 4; do cp $SRC/${TAG}_bench_p2p$N.json $DST/${TAG}_bench_p2p${N}_one_gpu.json; done
cp $(find $SRC/${TAG}_p2p2_stats -name "*kernel_stats.csv" | head -1) $DST/${TAG}_bench_p2p2_rank0_kernel_stats.csv
for R in c1 c5; do cp $(find $SRC/${TAG}_rows_${R}_stats -name "*kernel_stats.csv" | head -1) $DST/${TAG}_bench_rows_${R}_kernel_stats.csv; done
cp $SRC/${TAG}_timeline_none.txt $DST/${TAG}_timeline_B32.txt
cp $SRC/${TAG}_timeline_per_angle.txt $DST/${TAG}_timeline_per_angle.txt
cp $SRC/${TAG}_timeline_vr16.txt $DST/${TAG}_timeline_vr16.txt
ls -la $DST | grep ${TAG}
