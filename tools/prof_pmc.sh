# one rocprofv3 --pmc pass with the given counters (<= 8 SQ_, 4 TCC_, ... per pass) over a python script
# usage (GPU box): bash tools/prof_pmc.sh <tag> "<counters>" <script.py> [flags...]; output gpurun_out/pmc_<tag>/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; CTRS=$2
shift; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 "$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].strip('"')
    n = n[5:] if n.startswith("void ") else n
    agg[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if k.startswith("vx_"):
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
