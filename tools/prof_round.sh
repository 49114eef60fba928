# The three profiled commands + the plain bench of a profiles/rNN_x_* row (kernel stats, PMC HBM traffic, bench line).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-side-configs > $O/stats.log 2>&1
echo 'kernel stats done'; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp32-reference --no-side-configs > $O/fetch.log 2>&1
echo 'FETCH_SIZE pass done'; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp32-reference --no-side-configs > $O/write.log 2>&1
echo 'WRITE_SIZE pass done'; python3 tools/pmc_traffic.py $O/fetch $O/write $O/pmc_hbm_traffic_per_kernel.csv
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/fetch $O/write $O/stats
python3 bench.py --steps 8 --warmup 2 > $O/bench.log 2>&1
tail -c 300 $O/bench.log
ls -la $O
