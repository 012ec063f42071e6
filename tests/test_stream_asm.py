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


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    asm = str(tmp_path_factory.mktemp("asm") / "rspt.s")
    subprocess.check_call(
        [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value", "-w",
         "-I" + os.path.join(ROOT, "include"), "-o", asm, os.path.join(ROOT, "rspt_amd", "csrc", "rspt_hip.hip")]
    )
    return asm


def test_no_register_is_touched_between_a_hand_issued_load_and_its_wait(device_asm):
    asm = device_asm
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


def test_every_barrier_is_reached_with_the_lds_stores_published(device_asm):
    """No kernel reaches an s_barrier with an LDS store of its own in flight (tools/check_barrier_waits.py walks every
    control-flow graph).  hipcc was seen to leave the `s_waitcnt lgkmcnt(0)` of a __syncthreads() out where a thread-0-only block
    of LDS stores reached it over a loop back-edge: a one-in-a-few-launches corruption at full batch size only."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_barrier_waits.py"), device_asm], capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if " barriers, " in ln]
    names = " ".join(lines)
    for k in ("k_dec_block", "k_hist", "k_encode", "k_tile_stream", "k_fwht64k", "k_iir_pipe"):
        assert k in names, (k, names[:400])
    bad = [ln for ln in lines if " 0 unpublished" not in ln]
    assert not bad and r.returncode == 0, bad


def test_the_barrier_check_sees_the_hazard(tmp_path):
    """the pattern that was miscompiled, spelled out: LDS stores under a mask, back-edge, barrier at the loop header"""
    src = """_Z1kv:                                  ; @_Z1kv
; %bb.0:
	s_mov_b32 s0, 0
.LBB0_1:                                ; =>This Loop Header
	s_barrier
	ds_read_b32 v1, v0
	s_waitcnt lgkmcnt(0)
	s_cbranch_scc1 .LBB0_3
	s_and_saveexec_b64 s[2:3], vcc
	ds_write_b32 v0, v1
	s_or_b64 exec, exec, s[2:3]
	s_branch .LBB0_1
.LBB0_3:
	s_endpgm
.Lfunc_end0:
"""
    good = src.replace("\ts_or_b64 exec, exec, s[2:3]\n\ts_branch", "\ts_or_b64 exec, exec, s[2:3]\n\ts_waitcnt lgkmcnt(0)\n\ts_branch")
    for text, want in ((src, 1), (good, 0)):
        f = tmp_path / ("k%d.s" % want)
        f.write_text(text)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_barrier_waits.py"), str(f)], capture_output=True, text=True)
        assert r.returncode == want, (want, r.stdout)
        assert ("1 barriers, %d unpublished" % want) in r.stdout, r.stdout
