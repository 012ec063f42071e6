"""PCIe-inclusive rate of the host-pointer entry points (rspt_hip_compress / rspt_hip_decompress): one 64ch x 65536
x int32 block per call, buffers in pageable host memory, as the i_signal_packer drop-in sees it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rspt_amd import api, synth

nch, ns = 64, 65536
x = synth.synth_native(nch, ns, block_index=1).numpy().reshape(-1)
pk = api.new_xdelta_hzr(4, nch, ns, 3)
out = pk.compress(x)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    out = pk.compress(x); n += 1
dt = time.perf_counter() - t0
print("compress  : %.1f MSamples/s (%.2f ms per 16 MiB block, %d B out)" % (n * nch * ns / dt / 1e6, dt / n * 1e3, len(out)))
dec, used = pk.decompress(out)
assert dec == x.tobytes() and used == len(out)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 3.0:
    pk.decompress(out); n += 1
dt = time.perf_counter() - t0
print("decompress: %.1f MSamples/s (%.2f ms per block)" % (n * nch * ns / dt / 1e6, dt / n * 1e3))
