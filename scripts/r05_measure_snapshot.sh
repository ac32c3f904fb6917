O=gpurun_out/r05m2; mkdir -p $O
export TMPDIR=/tmp
python3 bench.py > $O/bench_c2_line.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python3 bench.py --mode snapshot --steps 30 --warmup 3 --no-cpu-baseline > $O/snapshot_bench_line.json 2> $O/snapshot.err; echo "snapshot rc=$?"
python3 bench.py --mode snapshot --separate-deposit --steps 30 --warmup 3 --no-cpu-baseline --no-extras > $O/snapshot_separate_bench_line.json 2>> $O/snapshot.err; echo "snapshot sep rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_snapshot -- python3 bench.py --mode snapshot --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/trace_snapshot.json 2> $O/trace_snapshot.err
find $O/trace_snapshot -name "*kernel_stats.csv" -exec cp {} $O/snapshot_kernel_stats.csv \;
rm -rf $O/trace_snapshot
scripts/traffic_pmc.sh snapshot --mode snapshot > /dev/null 2>&1; cp gpurun_out/traffic_snapshot.json $O/
python3 bench.py --mode grid3d --steps 30 --warmup 3 --no-cpu-baseline > $O/grid3d_bench_line.json 2> $O/grid3d.err; echo "grid3d rc=$?"
