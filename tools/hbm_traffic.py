#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same
bench command) into profiles/hbm_traffic.json: HBM bytes per compress_batch launch for
each stage name bench.py reports.

Corrections as MI355X_MICROARCH.md "HBM [CDNA4]" prescribes: both counters are in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE as is.
A stage that launches its kernel more than once per call (the front end's fix-up pass)
is summed per call.

usage: hbm_traffic.py <fetch_dir> <write_dir> [out.json]"""
import collections, csv, glob, json, os, sys

STAGES = {
    "preprocess": ["k_tile_stream", "k_tile_planes", "k_planar_planes", "k_tile_planar", "k_fwht", "k_dct"],
    "nb_scan": ["k_nb_scan"],
    "hzr_hist": ["k_hist"],  # (k_histlist and k_hist)
    "hzr_tree": ["k_tree"],
    "layout": ["k_layout"],
    "hzr_encode": ["k_encode("],
    "hzr_encode_small": ["k_encode_small"],
}


def collect(d, counter):
    tot = collections.Counter()
    calls = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for st, subs in STAGES.items():
                if any(s in name for s in subs):
                    tot[st] += float(r["Counter_Value"])
                    calls[st] += 1
    return tot, calls


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    ncall_f = fc["hzr_tree"] or 1   # one k_tree launch per compress_batch call
    ncall_w = wc["hzr_tree"] or 1
    res = {}
    detail = {}
    for st in STAGES:
        if st not in ft and st not in wt:
            continue
        rd = ft[st] * 1024 * 2 / ncall_f
        wr = wt[st] * 1024 / ncall_w
        res[st] = int(rd + wr)
        detail[st] = {"read_bytes": int(rd), "write_bytes": int(wr), "launches_per_call": fc[st] / ncall_f}
    res["_detail"] = detail
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    res["_kernels_sha"] = bench.kernels_sha()  # bench.py reports these figures only for the sources they were measured on
    res["_note"] = ("HBM bytes per rspt_hip_compress_batch_dev call; FETCH_SIZE(KiB)*1024*2 (gfx950 correction) + "
                    "WRITE_SIZE(KiB)*1024; separate --pmc passes; calls counted = %d / %d" % (ncall_f, ncall_w))
    s = json.dumps(res, indent=1)
    print(s)
    if out:
        open(out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
