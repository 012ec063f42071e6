"""Static check of the hand-scheduled loads of k_tile_stream (no GPU needed: hipcc cross-compiles).

The streaming front end issues its loads from inline asm and waits for them with explicit counts, behind the compiler's
back.  That is only sound if the generated code never touches a destination register between the load and
the wait that covers it -- e.g. through a copy the register allocator inserts.  tools/check_stream_regs.py scans the
device assembly for exactly that; the headline instantiations must come out clean."""
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
        if "hand-issued loads" in line:
            name, rest = line.split(":", 1)
            loads, waits, bad = [int(t) for t in rest.replace(",", " ").split() if t.isdigit()]
            summary[name.strip()] = (loads, waits, bad)
    # every instantiation was found and carries hand-issued loads and waits
    assert len(summary) == 12, summary
    assert all(v[0] >= 96 and v[1] >= 4 for v in summary.values()), summary
    # the scan is linear (no control-flow graph): instantiations whose slow paths are laid out behind the loop can show
    # false positives, the int32 / int16 xdelta kernels -- the headline path -- are straight enough to come out clean
    for name, (loads, waits, bad) in summary.items():
        if "streamILi4ELb1E" in name or "streamILi2ELb1E" in name:
            assert bad == 0, (name, bad, out[-2000:])
