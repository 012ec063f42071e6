"""Device-resident batched decompress rate (64 blocks of 64ch x 65536 x int32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rspt_amd import api, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "xdelta_hzr"
B, nch, ns = (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 64, 65536
dev = torch.device("cuda", 0)
d_src = synth.synth_batch_native(B, nch, ns, device=dev)
pk = api.SignalPacker(kind, 4, nch, ns, 3)
stride = (pk.max_compressed_size + 255) // 256 * 256
dst = torch.empty((B, stride), dtype=torch.uint8, device=dev)
sz = torch.empty(B, dtype=torch.int64, device=dev)
pk.compress_batch(d_src, dst, sz, stride)
torch.cuda.synchronize()
out = torch.empty_like(d_src)
used = torch.empty(B, dtype=torch.int64, device=dev)
pk.decompress_batch(dst, B, stride, out, used)
torch.cuda.synchronize()
ok = torch.equal(out, d_src) and torch.equal(used, sz); print(kind, "exact:", ok, "(lossy packers: False is expected)")
t0 = time.perf_counter(); n = 5
for _ in range(n):
    pk.decompress_batch(dst, B, stride, out, used)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("decompress_batch: %.2f ms per %d blocks = %.1f MSamples/s" % (dt * 1e3, B, B * nch * ns / dt / 1e6))
