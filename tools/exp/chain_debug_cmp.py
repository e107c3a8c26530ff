import torch, sys
a = torch.load(sys.argv[1]); b = torch.load(sys.argv[2])      # a = new, b = old
m = 256
it = torch.arange(m * m // 8); lane, T, jblk = it & 63, (it >> 6) & 15, it >> 10
e = torch.arange(8)
I = (16 * T + 4 * (lane >> 5))[:, None] + (e & 3) + 8 * (e >> 2)
J = (32 * jblk + (lane & 31))[:, None].expand(-1, 8)
def unpn(P):            # [BH, m*m] panel native -> [BH, m, m]
    M = torch.zeros(P.shape[0], m, m)
    M[:, I, J] = P.reshape(P.shape[0], -1, 8).float()
    return M
def rel(x, y): return float((x.float() - y.float()).abs().max() / (y.float().abs().max() + 1e-30))
print("zfT", rel(a["zfT"], b["zfT"]), "dX", rel(a["dX"], b["dX"]), "dz0", rel(a["dz0"], b["dz0"]))
for k in range(5, -1, -1):
    print(k, "Z", rel(a["saved"][k, 0], b["saved"][k, 0]), " V3", rel(a["work"][k, 0], b["work"][k, 0]), " 4W", rel(a["work"][k, 2], b["work"][k, 2]),
          " Un", rel(a["work"][k, 3], b["work"][k, 3]), " |V3| new/old", float(a["work"][k,0].float().norm()), float(b["work"][k,0].float().norm()))
# reference for the last iteration from the OLD run's own buffers: Un(k) = U'(k); U for k = 4 is old work[5, 3]
Z4 = unpn(b["saved"][4, 0].reshape(2, -1)); U4 = unpn(b["work"][5, 3].reshape(2, -1))
V3ref = 0.25 * U4 @ Z4
print("k=4: V3 old vs 1/4 U Z", rel(unpn(b["work"][4, 0].reshape(2, -1)), V3ref), "  new vs ref (with the NEW run's U)",
      rel(unpn(a["work"][4, 0].reshape(2, -1)), 0.25 * unpn(a["work"][5, 3].reshape(2, -1)) @ unpn(a["saved"][4, 0].reshape(2, -1))))
Un = unpn(a["work"][5, 3].reshape(2, -1)); Zn = unpn(a["saved"][4, 0].reshape(2, -1)); V = unpn(a["work"][4, 0].reshape(2, -1))
T = lambda x: x.transpose(-1, -2)
for name, c in (("U Z", Un @ Zn), ("Z U", Zn @ Un), ("U^T Z", T(Un) @ Zn), ("U Z^T", Un @ T(Zn)), ("(U Z)^T", T(Un @ Zn)), ("Z^T U", T(Zn) @ Un)):
    print(name, rel(V, 0.25 * c))
# k = 5 uses dzf: U5 = G^T packed; not dumped.  Check the forward-like first product of iteration 5 instead through Un(5) = U'(5)
X = None

if "G" in a:
    Z5 = unpn(a["saved"][5, 0].reshape(2, -1)); U5 = T(a["G"].float())
    print("k=5 new V3 vs 1/4 G^T Z5:", rel(unpn(a["work"][5, 0].reshape(2, -1)), 0.25 * U5 @ Z5), " old:", rel(unpn(b["work"][5, 0].reshape(2, -1)), 0.25 * U5 @ Z5))
