"""Static check of the hand-scheduled loads of k_tile_stream (no GPU needed: hipcc cross-compiles).

The streaming front end issues its loads from inline asm and waits for them with explicit counts, behind the compiler's
back.  That is only sound if the generated code never touches a destination register between the load and
the wait that covers it -- e.g. through a copy the register allocator inserts, or by parking another value there after
the loop.  tools/check_stream_regs.py walks the control-flow graph of the device assembly for exactly that; all twelve
instantiations must come out clean."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_register_is_touched_between_a_hand_issued_load_and_its_wait(tmp_path):
    asm = str(tmp_path / "rspt.s")
    subprocess.check_call(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value", "-w",
         "-I" + os.path.join(ROOT, "include"), "-o", asm, os.path.join(ROOT, "rspt_amd", "csrc", "rspt_hip.hip")]
    )
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_stream_regs.py"), asm], capture_output=True, text=True).stdout
    summary = {}
    for line in out.splitlines():
        if "suspicious accesses" in line:
            name, rest = line.split(":", 1)
            states, bad = [int(t) for t in rest.replace(",", " ").split() if t.isdigit()]
            summary[name.strip()] = (states, bad)
    # every instantiation (int32 / int24 / int16, xdelta and plain, whole and ragged tiles) was found, its control-flow graph
    # was walked (hundreds of ring states each) and no instruction touches a register with a hand-issued load in flight
    assert len(summary) == 12, summary
    assert all(v[0] >= 300 for v in summary.values()), summary
    for name, (states, bad) in summary.items():
        assert bad == 0, (name, bad, out[-3000:])
