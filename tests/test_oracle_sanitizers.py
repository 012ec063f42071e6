"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference itself has
undefined behaviour -- signed shifts, int overflow in the FWHT, a one-byte over-read -- the restatement must not).
oracle/Makefile builds librspt_oracle_asan.so; the known-answer inputs, a packer case of every kind and the IIR stage run
through it in a child process with the sanitizer runtime preloaded.  CPU build only (no GPU sanitizers on this pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import cases
from oracle.oracle import Oracle
o = Oracle(path=%(so)r)
n = 0
for name, data in cases.hzr_kat_inputs().items():
    s = o.hzr_encode(data)
    assert o.hzr_decode(s, data.size)[0] == data.tobytes(), name
    n += 1
want = {"xdelta_hzr", "hzr", "hadamard", "dct"}
for c in cases.packer_cases():
    if c["data"].size > 200000 or (c["kind"] == "dct" and c["ns"] > 1024):
        continue
    pk = o.packer(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    s = pk.compress(c["data"])
    dec, used, rc = pk.decompress(s)
    assert used == len(s) and rc == 0, c["name"]
    want.discard(c["kind"])
    n += 1
assert not want, want
for c in cases.iir_cases():
    if c["ns"] > 5000:
        continue
    o.iir_prefilter(c["data"], c["bps"], c["nch"], c["ns"], c["n"], c["d"], c["init"])
    n += 1
print("sanitized cases:", n)
"""


def test_restatement_is_clean_under_asan_ubsan():
    so = os.path.join(ROOT, "oracle", "librspt_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "librspt_oracle_asan.so"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": so}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "sanitized cases:" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
