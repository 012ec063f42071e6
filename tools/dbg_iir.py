import sys, numpy as np, torch
sys.path.insert(0, '.')
from rspt_amd import api, synth
from oracle.oracle import Oracle
import bench
orc = Oracle()
for (nch, ns, B) in ((64, 65536, 2), (64, 4096, 2), (12, 8192, 3), (64, 64, 2)):
    pk = api.SignalPacker("xdelta_hzr", 4, nch, ns, 3)
    src = synth.synth_batch_native(B, nch, ns, first_block=0, bps=4, ecg=True, device="cuda")
    for pc in (True, False):
        work = src.clone()
        pk.iir_prefilter_batch(work, bench.IIR_N, bench.IIR_D, bench.IIR_INIT, per_channel=pc)
        torch.cuda.synchronize()
        for b in range(B):
            want = np.frombuffer(orc.iir_prefilter(src[b].cpu().numpy(), 4, nch, ns, bench.IIR_N, bench.IIR_D, bench.IIR_INIT, shared_state=not pc), dtype=np.int32).reshape(ns, nch)
            got = work[b].cpu().numpy().view(np.int32).reshape(ns, nch)
            bad = np.argwhere(want != got)
            print(nch, ns, "per_channel" if pc else "shared", "block", b, "mismatches", len(bad), bad[:3].tolist(), [(int(want[tuple(i)]), int(got[tuple(i)])) for i in bad[:3]])
    pk.close()
