for k in 1 2 3; do timeout -k 10 100 python tools/soak.py 0.001 1731 2>&1 | grep -v amdgpu.ids | cut -c1-900; done
