export GPU_MAX_HW_QUEUES=16
timeout -k 10 200 python tools/overlap2.py 0:0:0
OV_GRAPH=0 OV_NCH=64 OV_NS=65536 timeout -k 10 300 python tools/overlap_small.py 64 16
