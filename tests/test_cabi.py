"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports
every symbol include/rspt_hip.h declares.  No compute calls without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "rspt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rspt_hip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from rspt_amd import api

    L = api.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "librspt_hip.so does not export %s" % n
    assert sorted(api.C_ABI_SYMBOLS) == names, "rspt_amd/api.py binding list drifted from include/rspt_hip.h"


def test_cxx_factories_are_exported():
    """include/signal_packer.h: the i_signal_packer statics must link from the .so."""
    import subprocess

    from rspt_amd import build

    out = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB]).decode()
    for f in ("new_xdelta_hzr", "delete_xdelta_hzr", "new_hzr", "delete_hzr", "new_dct", "delete_dct", "new_hadamard", "delete_hadamard"):
        assert re.search(r"_ZN15i_signal_packer\d+%sE" % f, out), f


def test_reference_style_program_compiles_against_our_header(tmp_path):
    """A program written like the reference's README example (README.md:49-83)
    compiles and links against include/signal_packer.h + librspt_hip.so."""
    import subprocess

    from rspt_amd import build

    src = tmp_path / "readme_example.cpp"
    src.write_text(
        """
#include <cstdint>
#include <cmath>
#include <iostream>
#include "signal_packer.h"
int main() {
    const int bytes_per_sample = 4, nr_samples = 8192, nr_channels = 1;
    static int32_t data_stream[nr_samples];
    for (int i = 0; i < nr_samples; ++i) data_stream[i] = sin(i / 100.0) * 1000.0;
    i_signal_packer* c = i_signal_packer::new_xdelta_hzr(bytes_per_sample, nr_channels, nr_samples, 3);
    size_t dst_max_len = nr_samples * nr_channels * bytes_per_sample * 2;
    static unsigned char dst[8192 * 4 * 2];
    size_t compressed_size = 0;
    c->compress((uint8_t*)data_stream, dst, dst_max_len, compressed_size);
    std::cout << "compressed_size: " << compressed_size << std::endl;
    i_signal_packer::delete_xdelta_hzr(c);
    return compressed_size == 2028 ? 0 : 1;
}
"""
    )
    exe = tmp_path / "readme_example"
    lib_dir = os.path.dirname(build.LIB)
    cmd = ["g++", "-std=c++11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + lib_dir, "-lrspt_hip",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    assert exe.exists()


def test_create_fails_loudly_without_device():
    from rspt_amd import api

    if api.lib().rspt_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.RsptHipError) as e:
        api.new_xdelta_hzr(4, 1, 8192, 3)
    assert e.value.status == -2  # RSPT_HIP_ERR_NO_DEVICE: no CPU fallback exists


def test_bad_arguments_are_rejected_before_touching_the_device():
    from rspt_amd import api

    import ctypes as C

    L = api.lib()
    h = C.c_void_p()
    for args in [(9, 4, 1, 16, 3), (1, 5, 1, 16, 3), (1, 4, 0, 16, 3), (1, 4, 1, 0, 3), (1, 4, 1, 16, 0), (1, 4, 1, 16, 5), (3, 4, 2, 100, 3)]:
        assert L.rspt_hip_packer_create(C.byref(h), *args, 0) == -1, args
