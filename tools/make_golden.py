#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python tools/make_golden.py [--use-installed] [--out tests/golden]

* `losses.MIRRORLoss`, `losses.mirror_loss.ClipLoss`, `losses.InfoNCE` are imported from
  /root/reference unmodified (pure torch) -> golden_losses.npz, golden_infonce.npz.
* `models/mirror.py` is loaded from /root/reference by path, with `timm` and
  `nystrom_attention` (absent from this image) replaced by the stand-ins of
  tools/oracle_shims.py -> golden_model_<cfg>.npz.  State-dicts/inputs come from
  oracle/synth.py (seeded); the four random draws of MIRROR.forward are injected by
  patching `torch.rand` / `_standard_normal` for the duration of the call, so the recorded
  outputs correspond to known noise tensors.

Nothing from /root/reference is copied: fixtures hold inputs, seeds and expected outputs.
"""
from __future__ import annotations

import argparse
import contextlib
import importlib.util
import os
import sys
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import synth  # noqa: E402
from oracle.mirror_oracle import Cfg, OUTPUT_NAMES, LOSS_NAMES  # noqa: E402
from tools import oracle_shims  # noqa: E402

TEMPLATE_W = dict(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15,
                  rna_retention_loss_weight=0.15, style_loss_weight=0.1, cluster_loss_weight=0.1)

MODEL_CASES = {
    # name: (cfg, batch, (wsi_ratio, rna_ratio), store_full)
    "tiny": (Cfg(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=32, wsi_num_tokens=20,
                 rna_encoder_depth=2, rna_num_heads=8, style_mlp_hidden_dim=64,
                 style_mlp_out_dim=32, style_latent_dim=16, num_prototypes=30), 3, (0.6, 0.4), True),
    "h12": (Cfg(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=96, wsi_num_tokens=64,
                rna_encoder_depth=2, rna_num_heads=12, num_prototypes=100), 4, (0.75, 0.75), False),
    "mid": (Cfg(wsi_embed_dim=96, rna_embed_dim=80, embed_dim=64, wsi_num_tokens=200,
                rna_encoder_depth=1, rna_num_heads=8, rna_mlp_ratio=4.0, num_prototypes=300,
                wsi_retention_decoder_depth=2, rna_retention_decoder_depth=2), 2, (0.75, 0.75), False),
    "c1": (Cfg(wsi_embed_dim=1024, rna_embed_dim=512, embed_dim=256, wsi_num_tokens=256,
               rna_encoder_depth=2, rna_num_heads=8, rna_mlp_ratio=4.0), 8, (0.75, 0.75), False),
}


def load_reference_model_module(use_installed: bool):
    used = oracle_shims.install(use_installed)
    spec = importlib.util.spec_from_file_location("ref_models_mirror", os.path.join(REF, "models", "mirror.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, used


def load_reference_losses():
    sys.path.insert(0, REF)
    try:
        import losses as ref_losses
        from losses.mirror_loss import ClipLoss
    finally:
        sys.path.remove(REF)
    return ref_losses, ClipLoss


@contextlib.contextmanager
def injected_noise(rands, normals):
    """Feed queued tensors to torch.rand (models/mirror.py:630, :516) and to the
    reparameterisation draw of torch.distributions.Normal.rsample (:832-833)."""
    import torch.distributions.normal as tdn
    rq, nq = list(rands), list(normals)
    orig_rand, orig_sn = torch.rand, tdn._standard_normal

    def fake_rand(*shape, **kw):
        t = rq.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone()

    def fake_sn(shape, dtype, device):
        t = nq.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone().to(dtype)

    torch.rand, tdn._standard_normal = fake_rand, fake_sn
    try:
        yield
    finally:
        torch.rand, tdn._standard_normal = orig_rand, orig_sn
    assert not rq and not nq, "reference consumed fewer random draws than expected"


def subsample(t: torch.Tensor, limit: int = 8192):
    flat = t.detach().flatten()
    if flat.numel() <= limit:
        return np.arange(flat.numel()), flat.numpy().copy()
    idx = np.linspace(0, flat.numel() - 1, limit).astype(np.int64)
    return idx, flat[idx].numpy().copy()


def gen_model_case(mod, ref_losses, name, cfg: Cfg, batch, ratios, store_full, out_dir, used):
    torch.manual_seed(0)
    seed = {"tiny": 11, "h12": 12, "mid": 13, "c1": 14}[name]
    orig_rna = mod.TransFormerHybrid
    if cfg.rna_num_heads != 12:  # reference hard-wires 12 heads
        mod.TransFormerHybrid = partial(orig_rna, num_heads=cfg.rna_num_heads)
    try:
        model = mod.mirror(
            wsi_embed_dim=cfg.wsi_embed_dim, rna_embed_dim=cfg.rna_embed_dim, embed_dim=cfg.embed_dim,
            wsi_num_tokens=cfg.wsi_num_tokens, wsi_retention_decoder_depth=cfg.wsi_retention_decoder_depth,
            rna_encoder_depth=cfg.rna_encoder_depth, rna_mlp_ratio=cfg.rna_mlp_ratio,
            rna_norm_layer="layernorm", rna_act_layer="gelu",
            rna_retention_decoder_depth=cfg.rna_retention_decoder_depth,
            style_mlp_hidden_dim=cfg.style_mlp_hidden_dim, style_mlp_out_dim=cfg.style_mlp_out_dim,
            style_latent_dim=cfg.style_latent_dim, num_prototypes=cfg.num_prototypes)
    finally:
        mod.TransFormerHybrid = orig_rna
    ref_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    shapes = synth.param_shapes(cfg)
    assert sorted(ref_shapes) == sorted(shapes), (
        "state-dict contract mismatch", set(ref_shapes) ^ set(shapes))
    sd = synth.synth_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    model.eval()  # dropout off; masking/eps draws still fire
    wsi, rna, noise = synth.synth_batch(cfg, batch, seed + 1000)

    with injected_noise([noise["wsi_mask"], noise["rna_mask"]], [noise["wsi_eps"], noise["rna_eps"]]):
        outs = model(wsi, rna, wsi_mask_ratio=ratios[0], rna_mask_ratio=ratios[1])
    assert len(outs) == 15
    loss_t = ref_losses.MIRRORLoss(**TEMPLATE_W)(*outs)
    loss_d = ref_losses.MIRRORLoss()(*[o.detach() for o in outs])
    model.zero_grad()
    loss_t[0].backward()

    rec = {
        "cfg_json": np.array(str(cfg.__dict__)),
        "batch": np.array(batch), "seed": np.array(seed), "ratios": np.array(ratios),
        "shim_timm": np.array(used["timm"]), "shim_nystrom": np.array(used["nystrom_attention"]),
        "sd_checksum": np.array(synth.checksum([sd[k] for k, _ in shapes])),
        "in_checksum": np.array(synth.checksum([wsi, rna] + [noise[k] for k in sorted(noise)])),
        "loss_template": np.array([float(x) for x in loss_t], dtype=np.float64),
        "loss_default": np.array([float(x) for x in loss_d], dtype=np.float64),
        "keys": np.array([k for k, _ in shapes]),
    }
    for nm, o in zip(OUTPUT_NAMES, outs):
        idx, val = subsample(o)
        rec[f"out_idx/{nm}"], rec[f"out_val/{nm}"] = idx, val
        rec[f"out_sum/{nm}"] = np.array([float(o.double().sum()), float(o.double().abs().sum())])
    params = dict(model.named_parameters())
    gn = []
    for k, _ in shapes:
        g = params[k].grad
        gn.append(0.0 if g is None else float(g.double().norm()))
        if g is not None and g.numel() <= 2048:
            rec[f"grad/{k}"] = g.numpy().copy()
    rec["grad_norm"] = np.array(gn, dtype=np.float64)
    if store_full:
        for k, _ in shapes:
            rec[f"sd/{k}"] = sd[k].numpy()
        rec["in/wsi"], rec["in/rna"] = wsi.numpy(), rna.numpy()
        for k, v in noise.items():
            rec[f"noise/{k}"] = v.numpy()
    path = os.path.join(out_dir, f"golden_model_{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: losses(template)={rec['loss_template']}  -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


CLS_CFG = Cfg(wsi_embed_dim=128, rna_embed_dim=96, embed_dim=96, wsi_num_tokens=50, rna_encoder_depth=2, rna_num_heads=12)


def gen_classifier(mod, out_dir, used):
    """Downstream model: MIRRORClassifier with both fusions and without RNA, eval mode."""
    from oracle import mirror_oracle as O
    rec = {"shim_timm": np.array(used["timm"]), "shim_nystrom": np.array(used["nystrom_attention"])}
    cfg, ncls, batch = CLS_CFG, 5, 3
    wsi, rna, _ = synth.synth_batch(cfg, batch, 4242)
    rec["in/wsi"], rec["in/rna"] = wsi.numpy(), rna.numpy()
    for fusion in ("concat", "add"):
        torch.manual_seed(0)
        model = mod.mirror_classifier(wsi_embed_dim=cfg.wsi_embed_dim, rna_embed_dim=cfg.rna_embed_dim, embed_dim=cfg.embed_dim,
                                      num_classes=ncls, rna_encoder_depth=cfg.rna_encoder_depth, rna_mlp_ratio=cfg.rna_mlp_ratio,
                                      rna_norm_layer="layernorm", rna_act_layer="gelu", fusion=fusion)
        shapes = synth.classifier_param_shapes(cfg, ncls, fusion)
        ref_shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        assert sorted(ref_shapes) == sorted(shapes), ("classifier state-dict mismatch", set(ref_shapes) ^ set(shapes))
        sd = synth.synth_state_dict(shapes, 77)
        model.load_state_dict(sd, strict=True)
        model.eval()
        pred = model(wsi, rna)
        pred_w = model(wsi, None) if fusion == "add" else None
        want = O.classifier_forward(sd, cfg, wsi, rna, fusion)
        assert torch.allclose(pred, want, rtol=1e-5, atol=1e-6), "oracle restatement differs from the reference classifier"
        rec[f"pred/{fusion}"] = pred.detach().numpy()
        if pred_w is not None:
            rec["pred/add_wsi_only"] = pred_w.detach().numpy()
        model.zero_grad()
        pred.square().sum().backward()
        params = dict(model.named_parameters())
        rec[f"keys/{fusion}"] = np.array([k for k, _ in shapes])
        rec[f"grad_norm/{fusion}"] = np.array([float(params[k].grad.double().norm()) for k, _ in shapes], dtype=np.float64)
        for k, _ in shapes:       # the generator seeds by key position: only the head differs between the two fusions
            if fusion == "concat" or k.startswith("head."):
                rec[f"sd/{fusion}/{k}"] = sd[k].numpy()
    path = os.path.join(out_dir, "golden_classifier.npz")
    np.savez_compressed(path, **rec)
    print(f"classifier: pred(concat)[0]={rec['pred/concat'][0]} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def gen_datafeed(out_dir):
    """Data feed: the reference dataset's __getitem__ (datasets/dataset_pretrain.py:150-167) on a tiny
    on-disk bank: a long slide (sampled without replacement) and a short one (with replacement), numpy's global RNG seeded."""
    import importlib.util
    import tempfile
    import pandas as pd
    from oracle import mirror_oracle as O
    spec = importlib.util.spec_from_file_location("ref_dataset_pretrain", os.path.join(REF, "datasets", "dataset_pretrain.py"))
    dmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dmod)
    g = torch.Generator().manual_seed(99)
    ids = ["TCGA-AA-0001-01Z-00-DX1", "TCGA-BB-0002-01Z-00-DX1", "TCGA-CC-0003-01Z-00-DX1"]
    lens, Fd, G, N = [37, 5, 12], 8, 6, 12
    slides = [torch.randn(n, Fd, generator=g) for n in lens]
    rna = torch.randn(len(ids), G, generator=g)
    rec = {"num_tokens": np.array(N), "seed": np.array(2024)}
    with tempfile.TemporaryDirectory() as d:
        fdir = os.path.join(d, "feat")
        os.makedirs(fdir)
        for sid, sl in zip(ids, slides):
            torch.save(sl, os.path.join(fdir, sid + ".pt"))
        pd.DataFrame(rna.numpy().astype(np.float64), index=[s[:15] for s in ids], columns=[f"g{j}" for j in range(G)]).to_csv(os.path.join(d, "rna.csv"))
        ds = dmod.TCGAWSIRNAPretrainDataset(fdir, os.path.join(d, "rna.csv"), N)
        order = list(ds.used_feature_ids)                 # os.listdir order
        np.random.seed(2024)
        items = [ds[i] for i in range(len(ds))] + [ds[0]]
    by_id = dict(zip(ids, range(len(ids))))
    rec["order"] = np.array([by_id[s] for s in order] + [by_id[order[0]]])
    np.random.seed(2024)
    for j, (w, r) in enumerate(items):
        k = int(rec["order"][j])
        ow, orr, idx = O.dataset_getitem(slides[k], rna[k].double().numpy(), N)
        assert torch.equal(ow, w) and torch.equal(orr, r), "oracle restatement differs from the reference dataset"
        rec[f"out/{j}/wsi"], rec[f"out/{j}/rna"], rec[f"out/{j}/idx"] = w.numpy(), r.numpy(), idx
    for k, sl in enumerate(slides):
        rec[f"slide/{k}"] = sl.numpy()
    rec["rna"] = rna.numpy()
    path = os.path.join(out_dir, "golden_datafeed.npz")
    np.savez_compressed(path, **rec)
    print(f"datafeed: {len(items)} items, lens {lens}, N={N} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def gen_losses(ref_losses, ClipLoss, out_dir):
    g = torch.Generator().manual_seed(2024)
    b, n, d, p, lat = 8, 16, 32, 30, 12

    def rn(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).requires_grad_(True)

    mask_w = (torch.rand(b, n, generator=g) > 0.3).float()
    mask_r = (torch.rand(b, d, generator=g) > 0.5).float()
    ins = [rn(b, d), rn(b, n, d), rn(b, n, d), mask_w, rn(b, p), rn(b, lat, scale=0.5), rn(b, lat, scale=0.5),
           rn(b, d), rn(b, d), rn(b, d), mask_r, rn(b, p), rn(b, lat, scale=0.5), rn(b, lat, scale=0.5),
           torch.tensor(14.2857, requires_grad=True)]
    rec = {}
    for tag, kw in (("default", {}), ("template", TEMPLATE_W)):
        for t in ins:
            t.grad = None
        out = ref_losses.MIRRORLoss(**kw)(*ins)
        out[0].backward()
        rec[f"loss_{tag}"] = np.array([float(x) for x in out], dtype=np.float64)
        for nm, t in zip(OUTPUT_NAMES, ins):
            if t.grad is not None:
                rec[f"grad_{tag}/{nm}"] = t.grad.numpy().copy()
    for nm, t in zip(OUTPUT_NAMES, ins):
        rec[f"in/{nm}"] = t.detach().numpy()
    cl = ClipLoss()(ins[0].detach(), ins[7].detach(), ins[14].detach())
    rec["clip_loss"] = np.array(float(cl))
    np.savez_compressed(os.path.join(out_dir, "golden_losses.npz"), **rec)
    print("losses:", rec["loss_default"], rec["loss_template"])

    g = torch.Generator().manual_seed(77)
    q = torch.randn(32, 128, generator=g)
    k = torch.randn(32, 128, generator=g) + 0.5 * q
    rec = {"q": q.numpy(), "k": k.numpy()}
    for sym in (False, True):
        for red in ("mean", "sum", "none"):
            for tau in (0.1, 0.07):
                qq, kk = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
                out = ref_losses.InfoNCE(temperature=tau, reduction=red, symmetric=sym)(qq, kk)
                tag = f"sym{int(sym)}_{red}_{tau}"
                rec[f"loss/{tag}"] = out.detach().numpy()
                out.sum().backward()
                rec[f"gq/{tag}"], rec[f"gk/{tag}"] = qq.grad.numpy(), kk.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "golden_infonce.npz"), **rec)
    print("infonce cases:", len([k for k in rec if k.startswith('loss/')]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--use-installed", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    # torch 2.10 CPU: the oneDNN conv2d weight-gradient is WRONG for some depthwise shapes (e.g. the 33-tap
    # res_conv at n_p=256: error O(10) vs an explicit sum; forward and input-gradient are fine).  Recording the
    # reference with oneDNN off uses ATen's native convolution, which agrees with the explicit sum.
    torch.backends.mkldnn.enabled = False
    if not a.only or "datafeed" in a.only.split(","):
        gen_datafeed(a.out)
    if a.only == "datafeed":
        return
    ref_losses, ClipLoss = load_reference_losses()
    gen_losses(ref_losses, ClipLoss, a.out)
    mod, used = load_reference_model_module(a.use_installed)
    print("third-party packages:", used)
    for name, (cfg, batch, ratios, full) in MODEL_CASES.items():
        if a.only and name not in a.only.split(","):
            continue
        gen_model_case(mod, ref_losses, name, cfg, batch, ratios, full, a.out, used)
    if not a.only or "cls" in a.only.split(","):
        gen_classifier(mod, a.out, used)


if __name__ == "__main__":
    main()
