import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np
import cases
from rspt_amd import api
from oracle.oracle import Oracle
o = Oracle()
cs = {c["name"]: c for c in cases.packer_cases()}
c = cs[sys.argv[1] if len(sys.argv) > 1 else "ecg12x4096_dct"]
pk = api.new_dct(c["bps"], c["nch"], c["ns"])
s = pk.compress(c["data"])
N = c["nch"] * c["ns"]
planar = pk.debug_read(1, N * 4).view(np.int32)
planar2 = pk.debug_read(2, N * 4).view(np.int32)
want = o.native_to_i32(c["data"], c["ns"], c["nch"], c["bps"]).reshape(-1)
print("planar == deinterleaved:", (planar == want).all(), planar[:8], want[:8])
print("planar2 (dct out) first", planar2[:16], "nonzero", np.count_nonzero(planar2))
print("means", pk.debug_read(6, 3 * c["nch"]))
po = o.packer("dct", c["bps"], c["nch"], c["ns"]); w = po.compress(c["data"])
print(len(s), len(w))
