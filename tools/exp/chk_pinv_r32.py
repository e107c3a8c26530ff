import torch, sys
sys.path.insert(0, '/root/repo')
from mirror_amd import functional as Fn
gen = torch.Generator().manual_seed(5)
m, BH, iters = 384, 6, 6
DEV = 'cuda'
logits = torch.randn(BH, m, m, generator=gen) + 4.0 * torch.eye(m)
a2 = torch.softmax(logits, dim=-1).to(DEV).contiguous()
dZ = (torch.randn(BH, m, m, generator=gen) * 0.1).to(DEV)
z_t, saved_t, st_t = Fn.pinv_forward_tile(a2, iters)
a = Fn.pinv_backward_tile(a2, saved_t, st_t, dZ)
b = Fn.pinv_backward_tile(a2, saved_t, st_t, dZ)
Fn._PINV_R32 = False
c = Fn.pinv_backward_tile(a2, saved_t, st_t, dZ)
d = Fn.pinv_backward_tile(a2, saved_t, st_t, dZ)
print('new vs new', float((a - b).abs().max()), 'old vs old', float((c - d).abs().max()), 'new vs old', float((a - c).abs().max()), 'scale', float(a.abs().max()))
