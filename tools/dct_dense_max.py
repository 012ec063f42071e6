"""one-off check: dct at the largest dense-table size (ns = 32767, the reach of the reference's int table index) against the C
restatement (4.3 GB table, 10^9 cosines: a minute of host time)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rspt_amd import api, synth
from oracle.oracle import Oracle
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 32767
x = np.ascontiguousarray(synth.synth_native(1, ns, block_index=5, ecg=True).numpy().reshape(-1))
t = time.time(); pk = api.new_dct(4, 1, ns); t1 = time.time() - t
got = pk.compress(x)
t = time.time(); po = Oracle().packer("dct", 4, 1, ns); t2 = time.time() - t
want = po.compress(x)
print("ns %d: gpu create %.1f s, oracle create %.1f s, stream %d bytes, identical %s" % (ns, t1, t2, len(got), got == want))
dec, used = pk.decompress(want)
print("decode of the oracle's stream identical:", bytes(po.decompress(want)[0]) == dec)
