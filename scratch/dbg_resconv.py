import torch, torch.nn.functional as F, sys
sys.path.insert(0, '.')
from mirror_amd import kernels as K
torch.manual_seed(0)
for (B, dh, n_p, zero) in [(2, 8, 256, True), (2, 8, 256, False), (1, 8, 256, False), (2, 8, 192, False), (2, 8, 320, False), (2,32,384,False)]:
    h, taps = 8, 33
    D = h * dh
    qkv = torch.randn(B, n_p, 3 * D)
    dout = torch.randn(B, n_p, D)
    if zero: dout[:, :30] = 0
    w = torch.randn(h, 1, taps, 1, requires_grad=True)
    v = qkv[..., 2 * D:].reshape(B, n_p, h, dh).permute(0, 2, 1, 3).contiguous()
    ref = F.conv2d(v, w, padding=(taps // 2, 0), groups=h)
    ref.backward(dout.reshape(B, n_p, h, dh).permute(0, 2, 1, 3))
    # independent reference
    vp = F.pad(v, (0, 0, 16, 16))
    do = dout.reshape(B, n_p, h, dh).permute(0, 2, 1, 3)
    man = torch.stack([(do * vp[:, :, j:j + n_p]).sum((0, 2, 3)) for j in range(taps)], dim=1)
    dw = torch.zeros(h, taps, device='cuda')
    K.resconv_wgrad(qkv.cuda()[..., 2 * D:], dout.cuda(), dw, h)
    e1 = (dw.cpu() - w.grad.reshape(h, taps)).abs().max()
    e2 = (dw.cpu() - man).abs().max()
    e3 = (man - w.grad.reshape(h, taps)).abs().max()
    print(B, dh, n_p, zero, 'hip-vs-torch', float(e1), 'hip-vs-manual', float(e2), 'manual-vs-torch', float(e3))
