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
    # every instantiation (int32 / int24 / int16 / int8, xdelta and plain, whole and ragged tiles) was found, its control-flow graph
    # was walked (hundreds of ring states each) and no instruction touches a register with a hand-issued load in flight
    assert len(summary) == 16, summary
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


def _kernel_bodies(asm_path):
    """mangled kernel name -> its instruction lines (from the symbol's label to its .Lfunc_end)"""
    import re

    bodies, name, cur = {}, None, []
    for ln in open(asm_path):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m and name is None:
            name, cur = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                bodies[name] = cur
                name = None
            else:
                cur.append(ln.strip())
    return bodies


def test_no_fused_multiply_add_where_the_reference_rounds_twice(device_asm):
    """The IIR pre-filter and the dct's table path restate the reference's arithmetic operation for operation: every product
    and every sum is rounded on its own (iir_filter.cpp:46-116, signal_packer_dct.cpp:76-87).  hipcc contracts a * b + c into
    one FMA by default and HIP's __dmul_rn / __dadd_rn do not stop it: round 2's k_iir held 83 v_fma_f64 and moved one output
    count per ~2 M samples -- on no fixture, only on the full-size batch.  The ISA of these kernels must hold no fp FMA."""
    import re

    bodies = _kernel_bodies(device_asm)
    fma = re.compile(r"^v_(fma|fmac|mad|mac|pk_fma)_(f64|f32|legacy_f32)\b")
    iir = [k for k in bodies if "k_iir" in k]
    dct = [k for k in bodies if re.search(r"5k_dctILb[01]E", k)]  # rspt::k_dct<true|false>: the dense-table transform
    assert len(iir) >= 64 and len(dct) == 2, (len(iir), dct)  # (k_iir / k_iir_pipe x sample width x order x mode)
    for k in iir + dct:
        hits = [ln for ln in bodies[k] if fma.match(ln)]
        assert not hits, (k, hits[:4])
        # and the arithmetic is there at all: separate fp64 multiplies and adds
        assert any(ln.startswith("v_mul_f64") for ln in bodies[k]) or "k_dct" in k, k
        assert any(ln.startswith("v_add_f64") for ln in bodies[k]), k
