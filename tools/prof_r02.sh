# rocprofv3 passes behind profiles/r02_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs
# (MI355X_MICROARCH.md, HBM section: one counter family per run, never together with a trace domain).
# usage (on the GPU box): [PASSES="stats fetch write"] bash tools/prof_r02.sh <tag> <script.py> [script flags...]
#   bash tools/prof_r02.sh default bench.py --steps 30 --warmup 5 --no-cpu-baseline --configs none
#   bash tools/prof_r02.sh retinaface tools/bench_configs.py retinaface --no-cpu
# output: gpurun_out/prof_<tag>/{stats,fetch,write}/ + the script's stdout per pass; tools/summarize_prof.py turns
# it into profiles/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
shift
OUT=gpurun_out/prof_$TAG
PASSES=${PASSES:-stats fetch write}
rm -rf $OUT
mkdir -p $OUT
echo "python3 $*" > $OUT/command.txt
for pass in $PASSES; do
    case $pass in
    stats) timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/bench_stats.json 2> $OUT/stats.err ;;
    fetch) timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err ;;
    write) timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/bench_write.json 2> $OUT/write.err ;;
    esac
    echo $pass done
done
