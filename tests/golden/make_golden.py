#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference.

Runs only in the dev container: it needs oracle/_ref/librspt_ref.so, which
oracle/Makefile compiles from the reference's own sources under /root/reference
(nothing of the reference is copied into this repo).  The fixtures are data:
input descriptions (tests/cases.py builds the bytes deterministically) and the
reference's outputs -- sizes, hashes, final nb, PRDN, and the full stream for
small cases.

    python tests/golden/make_golden.py
"""
import json
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle.oracle import Oracle, Ref  # noqa: E402

FULL_LIMIT = 600  # bytes; longer streams are pinned by size + two hashes


def chunk_count(stream, hdr):
    """number of [u32 len][hzr stream] chunks = the packer's final nb."""
    pos, k = 1 + hdr, 0
    while pos < len(stream):
        (ln,) = struct.unpack_from("<I", stream, pos)
        pos += 4 + ln
        k += 1
    assert pos == len(stream)
    return k


def main():
    ref, orc = Ref(), Oracle()
    out = {"generator": "tests/golden/make_golden.py (oracle/_ref = reference compiled from /root/reference)", "hzr": {}, "packers": {}}
    for name, data in cases.hzr_kat_inputs().items():
        s = ref.hzr_encode(data)
        assert ref.hzr_decode(s, data.size) == data.tobytes()
        ok, n = ref.hzr_verify(s)
        assert ok and n == data.size
        e = {"in_size": int(data.size), "in_crc32": zlib.crc32(data.tobytes()), "size": len(s), "fnv1a": orc.fnv1a(s), "crc32": zlib.crc32(s)}
        if len(s) <= FULL_LIMIT:
            e["stream"] = s.hex()
        out["hzr"][name] = e
    for c in cases.packer_cases():
        pk = ref.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
        s = pk.compress(c["data"])
        dec, used, rc = pk.decompress(s)
        assert used == len(s) and rc == 0, (c["name"], used, len(s))
        hdr = 3 * c["nch"] if c["kind"] in ("dct", "hadamard") else 0
        e = {
            "kind": c["kind"], "bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "nb": c["nb"],
            "in_crc32": zlib.crc32(c["data"].tobytes()),
            "size": len(s), "fnv1a": orc.fnv1a(s), "crc32": zlib.crc32(s),
            "final_nb": chunk_count(s, hdr),
            "decoded_crc32": zlib.crc32(dec),
            "lossless": dec == c["data"].tobytes(),
        }
        if c["kind"] in ("dct", "hadamard"):
            v = orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"])
            e["prdn"] = None if v != v else v  # NaN (degenerate denominators) -> null
        if c["store"] == "full" and len(s) <= FULL_LIMIT:
            e["stream"] = s.hex()
        out["packers"][c["name"]] = e
        print("%-28s %-10s size %8d nb %d->%d fnv %08x %s" % (c["name"], c["kind"], len(s), c["nb"], e["final_nb"], e["fnv1a"], "prdn %s" % e["prdn"] if "prdn" in e else ""))
        pk.close()
    out["dct_big"] = {}
    for c in cases.dct_big_cases():
        pk = ref.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
        s = pk.compress(c["data"])
        dec, used, rc = pk.decompress(s)
        assert used == len(s) and rc == 0
        out["dct_big"][c["name"]] = {
            "kind": c["kind"], "bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "nb": c["nb"],
            "in_crc32": zlib.crc32(c["data"].tobytes()), "size": len(s), "fnv1a": orc.fnv1a(s), "crc32": zlib.crc32(s),
            "decoded_crc32": zlib.crc32(dec), "prdn": orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"]),
            "stream": s.hex(),  # the whole stream: the GPU's FFT path is compared coefficient by coefficient
        }
        print("%-28s %-10s size %8d prdn %s" % (c["name"], c["kind"], len(s), out["dct_big"][c["name"]]["prdn"]))
        pk.close()
    out["dct_dense_big"] = {}
    for c in cases.dct_dense_big_cases():
        pk = ref.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
        s = pk.compress(c["data"])
        dec, used, rc = pk.decompress(s)
        assert used == len(s) and rc == 0
        out["dct_dense_big"][c["name"]] = {
            "kind": c["kind"], "bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "nb": c["nb"],
            "in_crc32": zlib.crc32(c["data"].tobytes()), "size": len(s), "fnv1a": orc.fnv1a(s), "crc32": zlib.crc32(s),
            "decoded_crc32": zlib.crc32(dec), "prdn": orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"]),
            "stream": s.hex(),
        }
        print("%-28s %-10s size %8d prdn %s" % (c["name"], c["kind"], len(s), out["dct_dense_big"][c["name"]]["prdn"]))
        pk.close()
    out["hadamard_big"] = {}
    for c in cases.hadamard_big_cases():  # (needs `ulimit -s unlimited`: fwht.c keeps two n-point arrays on the stack)
        pk = ref.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
        s = pk.compress(c["data"])
        dec, used, rc = pk.decompress(s)
        assert used == len(s) and rc == 0
        out["hadamard_big"][c["name"]] = {
            "kind": c["kind"], "bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "nb": c["nb"],
            "in_crc32": zlib.crc32(c["data"].tobytes()), "size": len(s), "fnv1a": orc.fnv1a(s), "crc32": zlib.crc32(s),
            "decoded_crc32": zlib.crc32(dec), "prdn": orc.prdn(c["data"], dec, c["ns"], c["nch"], c["bps"]),
        }
        print("%-28s %-10s size %8d prdn %s" % (c["name"], c["kind"], len(s), out["hadamard_big"][c["name"]]["prdn"]))
        pk.close()
    out["iir"] = {}
    for c in cases.iir_cases():
        filt = ref.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"])
        # and what the harness does next (rspt_test.cpp:141-142): xdelta_hzr on the filtered block
        pk = ref.packer("xdelta_hzr", c["bps"], c["nch"], c["ns"], 3)
        s = pk.compress(np.frombuffer(filt, dtype=np.uint8))
        pk.close()
        out["iir"][c["name"]] = {
            "bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "n": c["n"], "d": c["d"], "init": c["init"],
            "in_crc32": zlib.crc32(c["data"].tobytes()), "filtered_crc32": zlib.crc32(filt), "filtered_fnv1a": orc.fnv1a(filt),
            "xdelta_size": len(s), "xdelta_fnv1a": orc.fnv1a(s),
        }
        print("%-28s filtered crc %08x xdelta %d" % (c["name"], zlib.crc32(filt), len(s)))
    # the pre-filter at full size (CRCs only): the harness's shared filter object (what the reference runs), a fresh filter per
    # channel (the reference run channel by channel on one-channel blocks), each with and without the history initialisation
    # (without it the GPU takes the one-thread-per-channel kernel)
    c = cases.iir_big_case()
    data = cases.iir_big_data(c)
    x = data.view(np.int32).reshape(c["ns"], c["nch"])
    out["iir_big"] = {c["name"]: {"bps": c["bps"], "nch": c["nch"], "ns": c["ns"], "n": c["n"], "d": c["d"], "block": c["block"],
                                  "in_crc32": zlib.crc32(data.tobytes()), "modes": {}}}
    for init in (c["init"], 0):
        shared = ref.iir_prefilter(data, c["bps"], c["nch"], c["ns"], c["n"], c["d"], init)
        per = np.empty_like(x)
        for ch in range(c["nch"]):
            one = np.ascontiguousarray(x[:, ch]).view(np.uint8)
            per[:, ch] = np.frombuffer(ref.iir_prefilter(one, c["bps"], 1, c["ns"], c["n"], c["d"], init), dtype=np.int32)
        per = per.tobytes()
        for mode, buf in (("shared", shared), ("per_channel", per)):
            out["iir_big"][c["name"]]["modes"]["%s_init%d" % (mode, init)] = {"crc32": zlib.crc32(buf), "fnv1a": orc.fnv1a(buf)}
            print("%-28s %-18s crc %08x" % (c["name"], "%s init %d" % (mode, init), zlib.crc32(buf)))
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
