import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cases
from rspt_amd import api
pc = {c["name"]: c for c in cases.packer_cases()}
for name in ["readme_sine_xdelta_nb3", "readme_sine_xdelta_nb1"]:
    c = pc[name]
    pk = api.SignalPacker(c["kind"], c["bps"], c["nch"], c["ns"], c["nb"])
    try:
        got = pk.compress(c["data"])
        print(name, len(got))
    except Exception as e:
        print(name, "ERR", e)
