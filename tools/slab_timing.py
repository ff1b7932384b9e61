#!/usr/bin/env python3
"""Local operators of the headline in pieces (record cap of the context): do the records stay in the Infinity Cache when the pre-pass
and the cooperative kernel alternate over slabs?   tools/slab_timing.py [N cd fd]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import proton_amd as pa
N, cd, fd = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 3, 2)))
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
asm.generate_mesh(N, N)
out = None
for cap_mb in (4096, 384, 256, 192, 128, 96, 64, 48, 4096, 256, 128, 64):
    asm.ctx.set_record_cap(cap_mb << 20)
    asm.ctx.trim()
    for _ in range(3):
        out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",), out=out)
    asm.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",), out=out)
    asm.synchronize()
    print("record cap %4d MiB: %.3f ms per pass of the local operators" % (cap_mb, (time.perf_counter() - t0) / K * 1e3))
