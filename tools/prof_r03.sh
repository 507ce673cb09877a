# Round-3 profiles: for every tag below, the rocprofv3 passes of tools/prof_r02.sh (kernel-trace stats, then FETCH_SIZE
# and WRITE_SIZE in separate --pmc runs; one counter family per run, never together with a trace domain).
# usage (on the GPU box):  bash tools/prof_r03.sh <tag> [<tag> ...]      tags: see the case list
# afterwards, here:       python tools/summarize_r03.py
set -e
cd $GRAFT_REPO_ROOT
for TAG in "$@"; do
    case $TAG in
    default)      CMD="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --configs none" ;;
    dense)        CMD="bench.py --dense --steps 30 --warmup 5 --no-cpu-baseline --no-variants --configs none" ;;
    unfused)      CMD="bench.py --unfused --steps 30 --warmup 5 --no-cpu-baseline --no-variants --configs none" ;;
    retinaface)   CMD="tools/bench_configs.py retinaface --no-cpu --one-stream" ;;
    fpn_ssd)      CMD="tools/bench_configs.py ssd --no-cpu --one-stream" ;;
    stress)       CMD="tools/bench_configs.py stress --no-cpu --one-stream" ;;
    stress_all)   CMD="tools/bench_configs.py stress_all --no-cpu --one-stream" ;;
    resize720p)   CMD="tools/bench_configs.py resize_only --no-cpu --one-stream" ;;
    dense_cfg)    CMD="tools/bench_configs.py dense --no-cpu --one-stream" ;;
    b1)           CMD="tools/bench_configs.py b1 --no-cpu" ;;
    *) echo "unknown tag $TAG"; exit 2 ;;
    esac
    bash tools/prof_r02.sh $TAG $CMD
    echo "== $TAG done"
done
