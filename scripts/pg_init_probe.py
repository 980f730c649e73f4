"""How long does the one-rank process group take to come up in a fresh process?  python scripts/pg_init_probe.py [ifname]"""
import os
import sys
import time
t0 = time.perf_counter()
import torch
import torch.distributed as dist
t1 = time.perf_counter()
if len(sys.argv) > 1:
    os.environ["GLOO_SOCKET_IFNAME"] = sys.argv[1]
dist.init_process_group(backend="gloo", store=dist.HashStore(), rank=0, world_size=1)
t2 = time.perf_counter()
dist.barrier()
t3 = time.perf_counter()
print(f"ifname={os.environ.get('GLOO_SOCKET_IFNAME')} import {t1 - t0:.3f} s, init {t2 - t1:.3f} s, first barrier {t3 - t2:.3f} s")
dist.destroy_process_group()
