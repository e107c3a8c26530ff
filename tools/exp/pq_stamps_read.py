#!/usr/bin/env python3
"""Reads the in-kernel stamps of the diagnostic build (tools/exp/pq_stamps_patch.py): per-phase cycles of gemm_pq_kernel, workgroups 0 and 129.
   MIRROR_HIP_LIB=_exp_lib/libmirror_exp2.so python tools/exp/pq_stamps_read.py [N] [K]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mirror_amd import kernels as K
from mirror_amd import _lib
from mirror_amd._lib import MH_BF16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Kd = int(sys.argv[2]) if len(sys.argv) > 2 else 512
M, NS = 65536, 640
dev, bf = "cuda", torch.bfloat16
a = (torch.randn(M, Kd, device=dev) * .5).to(bf); w = (torch.randn(N, Kd, device=dev) * .5).to(bf)
out = torch.empty(M, N, device=dev, dtype=bf)
for _ in range(20): K.gemm(a, w.t(), out=out, mma=MH_BF16)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(2 * 8 * NS, dtype=np.uint32)
rc = lib.mh_exp_pq_stamps(buf.ctypes.data_as(C.c_void_p)); assert rc == 0, rc
nt = Kd // 64
d = lambda x, y: int((int(x) - int(y)) & 0xffffffff)
for blk in range(2):
    st = buf[blk * 8 * NS:(blk + 1) * 8 * NS].reshape(8, NS).astype(np.int64)
    print(f"== workgroup {0 if blk == 0 else 129}: [{M} x {Kd}] x [{Kd} x {N}], {nt} K-tiles per unit")
    for wv in (0, 4):
        s = st[wv]; n = int((s != 0).sum())
        # per unit: segments x (start, waited, past barrier 1, mfmas issued); loop end; packed; 2 halves x (written, past barrier, read, past barrier, stores issued); unit end
        pu = nt * 8 + 13
        units = n // pu
        print(f" wave {wv}: {n} stamps, {units} units of {pu}")
        for u in range(1, min(units, 4)):
            b = s[u * pu:(u + 1) * pu]; prev_end = s[u * pu - 1]
            seg = b[:nt * 8].reshape(nt * 2, 4)
            load = [d(x[1], x[0]) for x in seg]; bar1 = [d(x[2], x[1]) for x in seg]; mf = [d(x[3], x[2]) for x in seg]
            bar2 = [d(seg[i + 1][0], seg[i][3]) for i in range(nt * 2 - 1)] + [d(b[nt * 8], seg[-1][3])]
            e = b[nt * 8:]
            print(f"  unit {u}: total {d(b[-1], prev_end)} cyc  (previous unit's end -> first segment {d(b[0], prev_end)})")
            print(f"    load+wait {load}")
            print(f"    barrier1  {bar1}")
            print(f"    mfma      {mf}")
            print(f"    barrier2  {bar2}")
            names = ["level+pack", "write0", "sync", "read0", "sync", "stores0", "write1", "sync", "read1", "sync", "stores1", "sync+switch"]
            print(f"    K loop {d(e[0], b[0])}  epilogue: " + "  ".join(f"{nm} {d(e[i + 1], e[i])}" for i, nm in enumerate(names)))
