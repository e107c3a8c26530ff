import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mirror_amd import kernels as K
from mirror_amd._lib import MH_BF16, MH_F32
dev='cuda'; bf=torch.bfloat16; f32=torch.float32
def timeit(name, fn, flops, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    print(f"{name:56s} {ms*1e3:8.1f} us {flops/ms/1e9:8.1f} TF/s")
B,h,m=16,8,256
fl=2*B*h*m**3
xb=(torch.randn(B,h,m,m,device=dev)*.1).to(bf); yb=(torch.randn(B,h,m,m,device=dev)*.1).to(bf)
xf=xb.float(); yf=yb.float()
tr=lambda t:t.transpose(-1,-2)
timeit("bf16 x bf16 -> bf16 NN", lambda: K.gemm(xb,yb,mma=MH_BF16), fl)
timeit("bf16 x bf16 -> bf16 NN + diag + R", lambda: K.gemm(xb,xb,diag=15.,R=xb,rcoef=-7.,mma=MH_BF16), fl)
timeit("bf16 x bf16 -> f32  NN", lambda: K.gemm(xb,yb,mma=MH_BF16,out_dtype=f32), fl)
timeit("f32  x bf16 -> bf16 NN (a2.z)", lambda: K.gemm(xf,yb,mma=MH_BF16,out_dtype=bf), fl)
timeit("f32  x f32  -> f32  NN (old)", lambda: K.gemm(xf,yf,mma=MH_BF16), fl)
timeit("bf16^T x f32 -> bf16 TN (z^T dz)", lambda: K.gemm(tr(xb),yf,mma=MH_BF16,out_dtype=bf), fl)
timeit("f32 x bf16^T -> f32 NT (dz T3^T)", lambda: K.gemm(xf,tr(yb),mma=MH_BF16,out_dtype=f32), fl)
o=torch.zeros(B,h,m,m,device=dev)
timeit("f32 x bf16^T -> f32 NT accumulate + R", lambda: K.gemm(xf,tr(yb),out=o,accumulate=True,R=xf,rcoef=-7.,mma=MH_BF16), fl)
timeit("bf16^T x f32 -> f32 TN accumulate", lambda: K.gemm(tr(xb),yf,out=o,accumulate=True,mma=MH_BF16), fl)
timeit("f32^T x f32 -> f32 TN accumulate", lambda: K.gemm(tr(xf),yf,out=o,accumulate=True,mma=MH_BF16), fl)
