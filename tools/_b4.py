import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
dev='cuda'; bf=torch.bfloat16
def timeit(name, fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1)/reps*1e3:9.1f} us")
for m in (256, 128):
  for BH in (32, 128, 256):
    it=6
    x=(torch.randn(BH,m,m,device=dev)*2).softmax(-1)
    st=K.pinv_absmax(x); z0=K.pinv_z0(x,st)
    saved=torch.zeros((it,4,BH,m,m),device=dev,dtype=bf); K.cast(z0,bf,out=saved[0,0])
    zf=torch.empty((BH,m,m),device=dev,dtype=bf); xb=K.cast(x,bf)
    timeit(f"chain fwd m={m} BH={BH} (24 products)", lambda: K.pinv_chain_fwd(xb,saved,zf,it))
    work=torch.empty_like(saved); dX=torch.empty((BH,m,m),device=dev); dz0=torch.empty_like(dX)
    timeit(f"chain bwd m={m} BH={BH} (54 products)", lambda: K.pinv_chain_bwd(xb,saved,zf,work,dX,dz0,it))
