# rocprofv3 passes behind profiles/: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs.
# usage (on the GPU box): bash tools/prof_r01.sh [outdir-suffix] [extra bench.py flags...]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SUF=${1:-}
[ $# -gt 0 ] && shift
OUT=gpurun_out/prof$SUF
rm -rf $OUT
mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants "$@" > $OUT/bench_stats.json 2> $OUT/stats.err
echo stats done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-profile "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo fetch done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-profile "$@" > $OUT/bench_write.json 2> $OUT/write.err
echo write done
