"""CPU (no GPU): the C-ABI library loads and exports every symbol the header declares; host-side logic
(state-dict contract, operand-view -> GEMM descriptor mapping, bucket planning, loud failure without a GPU)."""
import os
import re

import pytest
import torch

import mirror_amd
from mirror_amd import _lib, kernels as K
from mirror_amd.engine import plan_buckets
from oracle import synth
from oracle.mirror_oracle import Cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mirror_hip.h")).read()
    declared = set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", header))
    declared -= {"mh_gemm_desc", "mh_stream"}
    assert len(declared) >= 48
    lib = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.mh_version() >= 100


def test_gemm_desc_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "mirror_hip.h")).read()
    body = header[header.index("typedef struct {"):header.index("} mh_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(void|float|int32_t|int64_t)\s*\*?", "", decl)
        names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
    names = [re.sub(r"^(const\s+)?(void|float)\s*\*\s*", "", n) for n in names]
    assert names == [f[0] for f in _lib.GemmDesc._fields_], names


def test_state_dict_contract_matches_reference_keys():
    import mirror_amd.models as M
    cfg = Cfg(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=96, wsi_num_tokens=20, rna_encoder_depth=3,
              wsi_retention_decoder_depth=2, rna_retention_decoder_depth=2, num_prototypes=30)
    m = M.mirror(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=96, wsi_num_tokens=20, rna_encoder_depth=3,
                 wsi_retention_decoder_depth=2, rna_retention_decoder_depth=2, num_prototypes=30, pretrained_cfg=None)
    got = sorted((k, tuple(v.shape)) for k, v in m.state_dict().items())
    assert got == sorted(synth.param_shapes(cfg))      # synth.param_shapes is checked against the reference in make_golden
    with pytest.raises(AssertionError):
        M.mirror(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=256, wsi_num_tokens=20)   # 256 % 12 != 0, as the reference
    M.mirror(wsi_embed_dim=64, rna_embed_dim=48, embed_dim=256, wsi_num_tokens=20, rna_num_heads=8, num_prototypes=10)


def test_no_cpu_fallback():
    import mirror_amd.models as M
    m = M.mirror(wsi_embed_dim=16, rna_embed_dim=8, embed_dim=32, wsi_num_tokens=4, rna_num_heads=8, num_prototypes=5)
    with pytest.raises(mirror_amd.MirrorHipError):
        m(torch.randn(1, 4, 16), torch.randn(1, 8))
    with pytest.raises(mirror_amd.MirrorHipError):
        K.gemm(torch.randn(4, 4), torch.randn(4, 4))
    from mirror_amd.losses import InfoNCE, MIRRORLoss
    with pytest.raises(ValueError):
        InfoNCE()(torch.randn(4), torch.randn(4, 3))
    with pytest.raises(mirror_amd.MirrorHipError):
        InfoNCE()(torch.randn(4, 3), torch.randn(4, 3))


def test_operand_views_map_to_strides():
    B, n, h, dh = 2, 12, 4, 8
    qkv = torch.zeros(B, n, 3 * h * dh)
    q = qkv.view(B, n, 3, h, dh)[:, :, 0].permute(0, 2, 1, 3)
    t4, rowmajor, ld, s1, s2 = K._mat(q)
    assert rowmajor and ld == 3 * h * dh and (s1, s2) == (n * 3 * h * dh, dh)
    t4, rowmajor, ld, s1, s2 = K._mat(q.transpose(-1, -2))
    assert (not rowmajor) and ld == 3 * h * dh
    w = torch.zeros(7, 5)
    assert K._mat(w.t())[1] is False and K._mat(w.t())[2] == 5
    with pytest.raises(mirror_amd.MirrorHipError):
        K._mat(torch.zeros(4, 6, 8)[:, ::2, ::2])


def test_bucket_plan_covers_arena_once():
    sizes = [5, 1000, 3, 64, 4096, 7, 7, 900]
    buckets, owner = plan_buckets(sizes, cap_elems=1024)
    assert buckets[0][0] == 0 and all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))
    assert sum(b[2] for b in buckets) == len(sizes) and owner == sorted(owner)
    padded = sum((s + 7) // 8 * 8 for s in sizes)
    assert buckets[-1][1] == padded
    assert all(b[1] - b[0] >= 1024 for b in buckets[:-1])


def test_checkpoint_saver_keeps_best_k(tmp_path):
    """Best-k selection of train_mirror.py:1053-1062 (timm CheckpointSaver semantics, decreasing loss metric)."""
    from mirror_amd.checkpoint import CheckpointSaver, resume_checkpoint
    model = torch.nn.Linear(3, 2)
    saver = CheckpointSaver(model, checkpoint_dir=str(tmp_path), decreasing=True, max_history=2)
    seen = []
    for epoch, metric in enumerate([5.0, 3.0, 4.0, 1.0, 2.0]):
        with torch.no_grad():
            model.weight.fill_(float(epoch))
        seen.append(saver.save_checkpoint(epoch, metric))
    assert seen == [(5.0, 0), (3.0, 1), (3.0, 1), (1.0, 3), (1.0, 3)]
    kept = sorted(f for f in os.listdir(tmp_path) if f.startswith("checkpoint-"))
    assert kept == ["checkpoint-3.pth.tar", "checkpoint-4.pth.tar"]             # the two lowest losses: epochs 3 and 4
    assert [m for _, m in saver.files] == [1.0, 2.0]
    other = torch.nn.Linear(3, 2)
    assert resume_checkpoint(other, os.path.join(tmp_path, "model_best.pth.tar")) == 4
    assert float(other.weight[0, 0]) == 3.0
    assert resume_checkpoint(other, os.path.join(tmp_path, "last.pth.tar")) == 5
    assert float(other.weight[0, 0]) == 4.0


def test_sample_indices_follows_the_reference_replace_rule():
    """datasets/dataset_pretrain.py:157-161: replace only when the slide is shorter than num_wsi_feature_tokens."""
    from mirror_amd.data import sample_indices
    g = torch.Generator().manual_seed(3)
    long = sample_indices(50, 20, g)
    assert long.shape == (20,) and len(set(long.tolist())) == 20 and int(long.max()) < 50 and int(long.min()) >= 0
    exact = sample_indices(20, 20, g)
    assert sorted(exact.tolist()) == list(range(20))
    short = sample_indices(5, 20, g)
    assert short.shape == (20,) and int(short.max()) < 5 and len(set(short.tolist())) <= 5
    with pytest.raises(ValueError):
        sample_indices(0, 4, g)
