"""print the kernel timeline of steady-state steps 5..7 of a traced bench run (tools/timeline.sh)"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rspt::" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rspt::", "")[:30]
rows.sort(key=lambda r: r["s"])
# main front-end launches (the blind fix-up launch of the same kernel lasts a few us)
fe = [i for i, r in enumerate(rows) if ("k_tile_stream" in r["n"] or "k_fwht64k" in r["n"] or "k_dctr_cols" in r["n"]) and r["e"] - r["s"] > 30000]
lo, hi = (fe[5], fe[8]) if len(fe) > 8 else (0, len(rows))
t0 = rows[lo]["s"]
busy_end = 0
idle = 0.0
for i in range(lo, hi):
    r = rows[i]
    beside = [q["n"] for q in rows[max(0, i - 12):i + 12] if q is not r and q["s"] < r["e"] and q["e"] > r["s"]]
    gap = r["s"] - busy_end if busy_end and r["s"] > busy_end else 0
    idle += gap
    busy_end = max(busy_end, r["e"])
    print("%-30s q%-2s start %8.1f  end %8.1f  dur %7.1f  %s%s" % (r["n"], r.get("Queue_Id", "?")[-2:], (r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3,
                                                                   ("idle before %5.1f  " % (gap / 1e3)) if gap else "", ("beside: " + ", ".join(sorted(set(beside)))) if beside else ""))
span = (rows[hi]["s"] - t0) / 1e3 if hi < len(rows) else (busy_end - t0) / 1e3
print("three steps: %.1f us = %.1f us per step; device idle (no kernel of ours running) %.1f us per step" % (span, span / 3, idle / 3e3))
