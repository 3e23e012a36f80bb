#!/usr/bin/env python3
"""Developer probe (GPU box): resident blocks per CU of the plain pooled variant against its dynamic LDS, as the runtime's
occupancy calculator sees it.  usage: tools/occ_probe.py <lib.so> [...]"""
import ctypes as C, sys, torch
torch.cuda.init()
for path in sys.argv[1:]:
    lib = C.CDLL(path)
    f = lib.rtdev_pool_blocks_per_cu
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 4 + [C.c_size_t]
    last = None
    for dyn in range(0, 16384, 128):
        n = f(0, 0, 0, 0, dyn)
        if n != last:
            print(path.split('/')[-1], "dyn", dyn, "blocks", n)
            last = n
