import torch, sys
sys.path.insert(0,'/root/repo')
from mirror_amd import kernels as K, functional as Fn
from mirror_amd._lib import MH_BF16
bf=torch.bfloat16
def run(B,T,R,Kd,N,kc,shared):
    g=torch.Generator().manual_seed(1)
    r0=T-R
    a=(torch.randn(B,T,Kd,generator=g)).cuda().to(bf)
    if kc: w=(torch.randn(N,Kd,generator=g)*0.05).cuda().to(bf); b2=w.t()
    else:  w=(torch.randn(Kd,N,generator=g)*0.05).cuda().to(bf); b2=w
    out=torch.full((B,T,N+256),7.0,device='cuda',dtype=bf)
    o3=out[...,:N]
    K.shared_chip=shared
    Fn._rows_window(a,b2,o3,r0,R,mma=MH_BF16, wt=(w.t().contiguous() if not kc else None))
    K.shared_chip=False
    torch.cuda.synchronize()
    ref=(a[:,r0:].float()@b2.float()).to(bf)
    err=(o3[:,r0:].float()-ref.float()).abs().max().item()
    untouched=bool((o3[:,:r0]==7).all()) and bool((out[...,N:]==7).all())
    # per-row error to find misplaced rows
    rowerr=(o3[:,r0:].float()-ref.float()).abs().amax(-1)
    bad=(rowerr>0.1).nonzero()
    print(f"B{B} T{T} R{R} K{Kd} N{N} kc{kc} shared{shared}: max err {err:.4f} untouched {untouched} bad rows {bad.shape[0]} first {bad[:5].tolist()}")
for shared in (False,True):
    run(2,1280,1025,512,1024,1,shared)
    run(2,1280,1025,1536,512,0,shared)
    run(16,4352,4097,512,1024,1,shared)
    run(16,4352,4097,1536,512,0,shared)
