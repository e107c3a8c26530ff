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
    assert lib.mh_version() == _lib.ABI_VERSION


def test_binding_signatures_restate_the_header_argument_lists():
    """mirror_amd/_lib.py restates every entry point's argument list by hand (_SIGS, ctypes has no header parser): a drifted list calls the
    library with shifted arguments.  Parse include/mirror_hip.h and compare, position by position, the CLASS of every parameter (pointer,
    int, int64, uint64, float) with the ctypes type of the binding, plus the trailing stream."""
    import ctypes as C
    header = open(os.path.join(ROOT, "include", "mirror_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decls = re.findall(r"^(?:int|int64_t|const char\*|void)\s+(mh_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", header, flags=re.M | re.S)
    assert len(decls) >= 100

    def cls(param):
        param = " ".join(param.split())
        if "*" in param or param.startswith("mh_stream"):
            return C.c_void_p
        t = param.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"int": C.c_int, "unsigned": C.c_int, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float}[t]

    checked, bad = 0, []
    for name, params in decls:
        if name not in _lib._SIGS:
            continue
        want = [cls(q) for q in params.split(",") if q.strip() and q.strip() != "void"]
        sig = list(_lib._SIGS[name])
        if sig and isinstance(sig[0], type) and issubclass(sig[0], C._Pointer):      # descriptor entry points: (struct *, stream)
            sig[0] = C.c_void_p
        got = sig + [C.c_void_p]
        checked += 1
        if len(got) != len(want) or any(g is not w for g, w in zip(got, want)):
            bad.append((name, [t.__name__ for t in got], [t.__name__ for t in want]))
    assert checked >= 90 and not bad, bad[:3]


def test_load_refuses_a_library_of_another_abi_generation(monkeypatch):
    """_SIGS restates the header's argument lists by hand, so a stale libmirror_hip.so (or one named by MIRROR_HIP_LIB)
    must be an error at load time, not shifted arguments at call time."""
    lib = _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", lib.mh_version() + 1)
    with pytest.raises(_lib.MirrorHipError, match="ABI v"):
        _lib.load()
    monkeypatch.setattr(_lib, "ABI_VERSION", lib.mh_version())
    assert _lib.load().mh_version() == lib.mh_version()


def _struct_fields(header: str, name: str):
    """Field names of `typedef struct { ... } name;` in declaration order."""
    end = header.index("} %s;" % name)
    body = header[header.rindex("typedef struct {", 0, end):end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s*", "", decl)          # the declaration's type
        names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
    return names


def test_gemm_desc_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "mirror_hip.h")).read()
    assert _struct_fields(header, "mh_gemm_desc") == [f[0] for f in _lib.GemmDesc._fields_]
    assert _struct_fields(header, "mh_gemm_epi") == [f[0] for f in _lib.GemmEpi._fields_]


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


def test_reference_import_names_resolve_through_the_alias():
    """after install_aliases() the reference trainers' own imports (`import models`, train_mirror.py:43;
    `from losses import MIRRORLoss`, :889-891; `from losses import InfoNCE`, train_pretrain.py:43) land in this build."""
    import importlib
    import sys
    saved = {k: sys.modules.get(k) for k in ("models", "models.mirror", "losses", "losses.mirror_loss", "losses.info_nce")}
    try:
        mirror_amd.install_aliases()
        models = importlib.import_module("models")
        from models import mirror, mirror_classifier  # noqa: F401
        from models.mirror import MIRROR, FeatureTransMILHybrid, TransFormerHybrid  # noqa: F401
        from losses import CrossEntropySurvLoss, InfoNCE, MIRRORLoss, NLLSurvLoss  # noqa: F401
        from losses.mirror_loss import ClipLoss  # noqa: F401
        from losses.info_nce import InfoNCE as I2
        import mirror_amd.models as M
        import mirror_amd.losses as L
        assert models is M and mirror is M.mirror and MIRRORLoss is L.MIRRORLoss and I2 is InfoNCE is L.InfoNCE
        assert sorted(importlib.import_module("losses").__all__) == ["CrossEntropySurvLoss", "InfoNCE", "MIRRORLoss", "NLLSurvLoss"]
        m = models.create_model("mirror", wsi_embed_dim=16, rna_embed_dim=8, embed_dim=96, wsi_num_tokens=4, num_prototypes=5,
                                pretrained_cfg=None)                 # timm injects pretrained*: filtered with a warning
        assert isinstance(m, MIRROR) and m.rna_encoder.num_heads == 12
        assert MIRRORLoss(alignment_loss_weight=0.5, cluster_loss_weight=0.1).cluster_loss_weight == 0.1
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_checkpoint_with_args_round_trips_and_reference_namespace_opens(tmp_path):
    """The reference always checkpoints `args` (train_mirror.py:920-930, an argparse.Namespace).  This build stores it as a
    plain dict (opens under torch.load's weights_only default), and a reference-style file that pickles the Namespace itself
    opens through the allow-list in load_checkpoint_file."""
    import argparse
    from mirror_amd.checkpoint import CheckpointSaver, load_checkpoint_file, resume_checkpoint
    model = torch.nn.Linear(3, 2)
    args = argparse.Namespace(model="mirror", lr=2e-5, model_kwargs={"embed_dim": 512}, amp=True)
    saver = CheckpointSaver(model, args=args, checkpoint_dir=str(tmp_path), max_history=1)
    saver.save_checkpoint(3, metric=1.5)
    other = torch.nn.Linear(3, 2)
    assert resume_checkpoint(other, os.path.join(tmp_path, "last.pth.tar")) == 4
    assert torch.equal(other.weight, model.weight)
    ck = load_checkpoint_file(os.path.join(tmp_path, "last.pth.tar"))
    assert ck["args"] == vars(args) and ck["metric"] == 1.5
    ref_style = os.path.join(tmp_path, "ref.pth.tar")
    torch.save({"epoch": 7, "arch": "mirror", "state_dict": model.state_dict(), "args": args, "version": 2}, ref_style)
    assert resume_checkpoint(other, ref_style) == 8
    assert load_checkpoint_file(ref_style)["args"].lr == 2e-5


def test_bench_gpus_n_without_launcher_refuses_when_devices_are_missing():
    """`python bench.py --gpus N` launches its N ranks itself; with fewer than N visible GPUs it must exit non-zero instead
    of silently benchmarking one rank (--gpus was parsed and never read).  No GPU here: 0 devices < 2."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MIRROR_BENCH_DIST")}
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but only" in r.stderr, (r.returncode, r.stderr[-300:])
    assert '"n_gpus"' not in r.stdout


def test_rna_block_ok_mirrors_the_kernel_limits():
    """mh_rna_block_fwd rejects D > 2048 and operand images above 158 KiB (csrc/rna_block.hip): the host-side admission test
    must say no to the same shapes so that Block.forward falls back to the composed ops instead of raising."""
    import torch
    from mirror_amd import kernels as K
    assert K.rna_block_ok(torch.zeros(16, 512), 512, 2048, 8)            # c2
    assert K.rna_block_ok(torch.zeros(16, 768), 768, 1984, 12)
    assert not K.rna_block_ok(torch.zeros(32, 1024), 1024, 4096, 8)     # 2 x 16 x 4104 x 2 B = 262 KiB operand image
    assert not K.rna_block_ok(torch.zeros(8, 4096), 4096, 4096, 8)      # D > 2048
    assert K.rna_block_ok(torch.zeros(16, 1024), 1024, 4096, 8)         # one row tile: 131 KiB


def test_pmc_summary_reports_the_replayed_step_on_its_own(tmp_path):
    """tools/pmc_summary.py cuts a PMC run into steps at the Adam launch and reports the step that repeats (the graph replays) apart
    from the eager steps and the one-time work around them: the launch and byte counts bench.py prints (roofline.step_launches /
    step_hbm_bytes) are those of a replayed step, not a run average."""
    import csv
    import json
    import subprocess
    import sys

    def write(fn, per_kernel_value):
        rows, did = [], 0
        # first step: one-time work (5 extra copies); two replays of 3 launches; one eager step of 4 launches
        plan = [["__amd_rocclr_copyBuffer"] * 5 + ["k_a", "k_b", "adam_kernel"], ["k_a", "k_b", "adam_kernel"], ["k_a", "k_b", "adam_kernel"],
                ["k_a", "k_b", "k_c", "adam_kernel"]]
        for step in plan:
            for k in step:
                did += 1
                rows.append({"Dispatch_Id": did, "Kernel_Name": f"void (anonymous namespace)::{k}(float*)", "Counter_Value": per_kernel_value.get(k, 0.0)})
        rows.reverse()          # the tool sorts by dispatch id
        with open(fn, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Value"])
            w.writeheader()
            w.writerows(rows)

    f, wv = tmp_path / "f.csv", tmp_path / "w.csv"
    write(f, {"k_a": 100.0, "k_b": 50.0, "adam_kernel": 10.0, "__amd_rocclr_copyBuffer": 1.0, "k_c": 7.0})     # KiB, counted at half
    write(wv, {"k_a": 20.0, "k_b": 0.0, "adam_kernel": 10.0, "k_c": 1.0})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), str(f), str(wv)], capture_output=True, text=True, check=True)
    d = json.loads(out.stdout)
    assert [p["launches"] for p in d["per_step"]] == [8, 3, 3, 4]
    rp = d["replayed_step"]
    assert rp["launches"] == 3 and rp["steps_averaged"] == 2
    assert rp["hbm_bytes"] == int((2 * (100 + 50 + 10) + (20 + 0 + 10)) * 1024)
    assert d["kernels"]["k_a"]["bytes_per_launch"] == int((2 * 100 + 20) * 1024) and d["kernels"]["k_a"]["launches"] == 4


def test_step_listing_picks_a_replayed_step_as_the_fastest(tmp_path):
    """tools/prof_step_listing.py <trace> fastest: the shortest step of a traced bench run (a graph replay), not the eager re-runs of
    the roofline leg at its end."""
    import csv
    import subprocess
    import sys
    rows, t = [], 0
    for step, dur in enumerate([900, 300, 310, 800]):          # eager, replay, replay, eager (ns per kernel)
        # (the tool takes Adam launches more than 50 dispatches apart as step ends: 60 kernels per step, two more in an eager one)
        for k in ["k_a"] * 60 + (["k_c"] * 2 if dur > 500 else []) + ["adam_kernel"]:
            rows.append({"Kernel_Name": f"void {k}(float*)", "Start_Timestamp": t, "End_Timestamp": t + dur, "Queue_Id": 1,
                         "Grid_Size_X": 256, "Grid_Size_Y": 1, "Grid_Size_Z": 1, "Workgroup_Size_X": 256})
            t += dur + 10
    fn = tmp_path / "trace.csv"
    with open(fn, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "prof_step_listing.py"), str(fn), "fastest"], capture_output=True, text=True, check=True)
    head = out.stdout.splitlines()[0]
    assert head.startswith("step: 61 launches"), head          # a replay (61), not an eager step (63)


def test_gbuf_n_uses_the_sink_slot_only_when_it_has_exactly_the_elements():
    """functional._gbuf_n (gradients of tokens / positional tables that reach their Function reshaped): the arena slot when it holds
    exactly the requested elements, a buffer of its own (returned through autograd) for a slice of a longer table or without a sink."""
    from mirror_amd import functional as Fn

    class Sink:
        def __init__(self, p):
            self.p, self.g, self.finished = p, torch.zeros_like(p), []

        def slot(self, t):
            return self.g if t.data_ptr() == self.p.data_ptr() else None

        def done(self, t):
            self.finished.append(t.data_ptr())

    pos = torch.randn(1, 9, 4)
    sink = Sink(pos)
    Fn.set_grad_sink(sink)
    try:
        buf, sunk = Fn._gbuf_n(pos, (36,))
        assert sunk and buf.data_ptr() == sink.g.data_ptr() and tuple(buf.shape) == (36,)
        assert Fn._gret(pos, buf, sunk) is None and sink.finished == [pos.data_ptr()]
        part, sunk2 = Fn._gbuf_n(pos[:, :5], (20,))           # same data_ptr, fewer elements: not the slot
        assert not sunk2 and part.data_ptr() != sink.g.data_ptr() and float(part.abs().sum()) == 0.0
        assert Fn._gret(pos, part, sunk2) is part
        other, sunk3 = Fn._gbuf_n(torch.randn(3), (3,))       # a tensor the sink does not know
        assert not sunk3
    finally:
        Fn.set_grad_sink(None)
    free, sunk4 = Fn._gbuf_n(pos, (36,))                      # no sink installed
    assert not sunk4 and tuple(free.shape) == (36,)


def test_loss_out_hands_over_views_of_one_small_tensor_without_a_copy():
    """TrainEngine._loss_out: six f32 scalars that are views of ONE small fresh tensor (MirrorLossTermsFn's result) are returned as
    they are; anything else (slices of a larger buffer such as the step's zero arena, separate tensors) is copied into a new tensor."""
    from mirror_amd.engine import TrainEngine
    out = torch.arange(8, dtype=torch.float32)
    views = tuple(out[i].reshape(()) for i in range(6))
    got = TrainEngine._loss_out(views)
    assert all(g.untyped_storage().data_ptr() == out.untyped_storage().data_ptr() for g in got)
    assert [float(g) for g in got] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    arena = torch.arange(4096, dtype=torch.float32)
    got2 = TrainEngine._loss_out(tuple(arena[64 * i].reshape(()) for i in range(6)))
    assert all(g.untyped_storage().data_ptr() != arena.untyped_storage().data_ptr() for g in got2)
    assert [float(g) for g in got2] == [0.0, 64.0, 128.0, 192.0, 256.0, 320.0]
    sep = tuple(torch.tensor(float(i)) for i in range(6))
    got3 = TrainEngine._loss_out(sep)
    assert len({g.untyped_storage().data_ptr() for g in got3}) == 1 and [float(g) for g in got3] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]


def test_dropout_step_take_hands_the_offset_to_the_caller_once():
    from mirror_amd import functional as Fn
    st = Fn._dropout_state
    saved = dict(st)
    try:
        st["base"], st["offset"] = torch.zeros(1, dtype=torch.int64), 24
        base, used = Fn.dropout_step_take()
        assert base is st["base"] and used == 24 and st["offset"] == 0
        assert Fn.dropout_step_take() == (None, 0)            # nothing consumed since: no counter for the caller to advance
        st["base"], st["offset"] = None, 8
        assert Fn.dropout_step_take() == (None, 0) and st["offset"] == 0
    finally:
        st.clear()
        st.update(saved)


def test_pending_landmark_merges_are_cleared_per_step_and_a_leftover_is_an_error():
    """entries of functional._pending_lm_merge pin their buffers through a closure and are keyed by a raw address; the engine
    clears the table at the start / end of a step and raises behind backward() when a merge was never consumed."""
    from mirror_amd import functional as Fn
    Fn._pending_lm_merge.clear()
    Fn._pending_lm_merge[1234] = (5678, lambda: None)
    Fn.pending_lm_merge_reset("test (lenient)")
    assert not Fn._pending_lm_merge
    Fn._pending_lm_merge[1234] = (5678, lambda: None)
    with pytest.raises(mirror_amd.MirrorHipError, match="never run"):
        Fn.pending_lm_merge_reset("test (strict)", strict=True)
    assert not Fn._pending_lm_merge          # cleared even when it raises: the next step starts clean


def test_split_k_partials_policy_switch_is_validated(monkeypatch):
    """kernels.SPLITK_PARTIALS (env MIRROR_SPLITK_PARTIALS) is 'bf16' (workspace + fold, the bf16 training policy's default) or 'f32'
    (f32 atomics); anything else is refused at import."""
    import importlib
    assert K.SPLITK_PARTIALS in ("bf16", "f32")
    monkeypatch.setenv("MIRROR_SPLITK_PARTIALS", "fp16")
    with pytest.raises(mirror_amd.MirrorHipError):
        importlib.reload(K)
    monkeypatch.setenv("MIRROR_SPLITK_PARTIALS", "f32")
    assert importlib.reload(K).SPLITK_PARTIALS == "f32"
    monkeypatch.delenv("MIRROR_SPLITK_PARTIALS")
    assert importlib.reload(K).SPLITK_PARTIALS == "bf16"
