import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mirror_amd.functional as Fn
from mirror_amd import kernels as K
x = torch.randn(16, 512, device="cuda", requires_grad=True)
def bench(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    t1=time.perf_counter(); torch.cuda.synchronize()
    return (t1-t)/n*1e6
xd = x.detach()
print("K.gelu_fwd direct      %.1f us" % bench(lambda: K.gelu_fwd(xd)))
print("Fn.gelu apply (fwd)    %.1f us" % bench(lambda: Fn.gelu(x)))
def fb():
    y = Fn.gelu(x); y.backward(xd)
print("Fn.gelu fwd+bwd        %.1f us" % bench(fb, 1000))
def fb3():
    y = Fn.gelu(Fn.gelu(Fn.gelu(x))); y.backward(xd)
print("3x chained fwd+bwd     %.1f us" % bench(fb3, 1000))
print("torch.empty            %.1f us" % bench(lambda: torch.empty(16,512,device='cuda')))
