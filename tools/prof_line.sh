# kernel stats of any bench line -> gpurun_out/prof_line/kernel_stats.csv      usage: PROF_ARGS="--packer hadamard --blocks 16" bash tools/prof_line.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_line
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-verify ${PROF_ARGS:-} > $O/stats.log 2>&1 || exit 1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:16]:
    if "at::native" in r["Name"]: continue
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
