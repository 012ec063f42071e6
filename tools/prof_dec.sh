# kernel stats of the decompress line -> gpurun_out/prof_dec/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_dec
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-verify --op decompress > $O/stats.log 2>&1 || exit 1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:14]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
