# timeline of one compress step: kernel start / end and the idle gaps between them (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gaps
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-verify ${GAPS_ARGS:-} > $O/log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rspt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full step: find the last k_tile_stream main launch and print from there
idx = [i for i, r in enumerate(rows) if "k_tile_stream" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else 0
t0 = int(rows[start]["Start_Timestamp"])
prev_end = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rspt::", "")[:28]
    gap = "" if prev_end is None else "gap %6.1f us" % ((s - prev_end) / 1e3)
    print("%-28s start %8.1f  end %8.1f  dur %7.1f  %s" % (name, s / 1e3, e / 1e3, (e - s) / 1e3, gap))
    prev_end = max(prev_end or 0, e)
PY
rm -rf $O/t
