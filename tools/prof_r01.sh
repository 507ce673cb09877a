set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/prof/bench_stats.json 2> gpurun_out/prof/stats.err
echo stats done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-profile > gpurun_out/prof/bench_fetch.json 2> gpurun_out/prof/fetch.err
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-profile > gpurun_out/prof/bench_write.json 2> gpurun_out/prof/write.err
echo write done
find gpurun_out/prof -type f | head -50
