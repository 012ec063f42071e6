"""PCIe-inclusive rate of the host-pointer entry points (rspt_hip_compress / rspt_hip_decompress): one 64ch x 65536
x int32 block per call, as the i_signal_packer drop-in sees it -- with the buffers in pageable host memory (what a caller of
the reference has today) and in page-locked memory from rspt_hip_host_alloc (DMA at link rate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rspt_amd import api, synth

nch, ns = 64, 65536
x = synth.synth_native(nch, ns, block_index=1).numpy().reshape(-1)
pk = api.new_xdelta_hzr(4, nch, ns, 3)


def rate(label, src, dst, back):
    n = pk.compress_into(src, dst)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 3.0:
        n = pk.compress_into(src, dst); k += 1
    dt = time.perf_counter() - t0
    print("%-12s compress  : %8.1f MSamples/s (%.3f ms per 16 MiB block, %d B out)" % (label, k * nch * ns / dt / 1e6, dt / k * 1e3, n))
    used = pk.decompress_into(dst, back)
    assert used == n and back.tobytes() == x.tobytes()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 3.0:
        pk.decompress_into(dst, back); k += 1
    dt = time.perf_counter() - t0
    print("%-12s decompress: %8.1f MSamples/s (%.3f ms per block)" % (label, k * nch * ns / dt / 1e6, dt / k * 1e3))


rate("pageable", x.copy(), np.empty(2 * x.size, dtype=np.uint8), np.empty(x.size, dtype=np.uint8))
hs, hd, hb = api.HostBuffer(x.size), api.HostBuffer(2 * x.size), api.HostBuffer(x.size)
hs.a[:] = x
rate("page-locked", hs.a, hd.a, hb.a)


def rate_many(label, src, out, n):
    lens = pk.compress_many(src, out)
    hold = None if label == "pageable" else api.HostBuffer(src.size)  # (kept alive: the array is a view of its memory)
    back = np.empty(src.size, dtype=np.uint8) if hold is None else hold.a
    assert (pk.decompress_many(out, back, lengths=lens) == lens).all() and back.tobytes() == src.tobytes()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 3.0:
        pk.decompress_many(out, back, lengths=lens); k += 1
    dt = time.perf_counter() - t0
    print("%-12s decompress_many, %d blocks per call: %8.1f MSamples/s (%.3f ms per block, %.1f GB/s of samples downloaded)"
          % (label, n, k * n * nch * ns / dt / 1e6, dt / k / n * 1e3, k * n * x.size / dt / 1e9))
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 3.0:
        lens = pk.compress_many(src, out); k += 1
    dt = time.perf_counter() - t0
    print("%-12s compress_many, %d blocks per call: %8.1f MSamples/s (%.3f ms per 16 MiB block, %.1f GB/s of samples uploaded)"
          % (label, n, k * n * nch * ns / dt / 1e6, dt / k / n * 1e3, k * n * x.size / dt / 1e9))


n = 32
stride = (pk.max_compressed_size + 255) // 256 * 256
xs = np.concatenate([synth.synth_native(nch, ns, block_index=i).numpy().reshape(-1) for i in range(n)])
rate_many("pageable", xs, np.empty((n, stride), dtype=np.uint8), n)
hs2, hd2 = api.HostBuffer(xs.size), api.HostBuffer(n * stride)
hs2.a[:] = xs
rate_many("page-locked", hs2.a, hd2.a.reshape(n, stride), n)
