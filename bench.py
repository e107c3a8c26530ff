#!/usr/bin/env python3
"""MIRROR pre-training step benchmark on MI355X (BASELINE.json metric: SSL samples/s, slide+RNA pairs).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under an external launcher (`python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N`, the reference's scripts/run_train_mirror.sh:62-71) the ranks read RANK / LOCAL_RANK / WORLD_SIZE from
the environment; without one (`python bench.py --gpus N` on its own) this process starts that launcher itself as a child
BEFORE it touches the GPU and exits with its code.

A step = prototype renorm -> forward -> MIRRORLoss -> backward (+ RCCL gradient all-reduce) -> Adam -> logit clamp
on one synthetic batch per GPU (BASELINE config 2: B x [4096 patch tokens x 1024-d] + [B x 2048 genes], D=512,
6-layer RNA encoder, bf16 MFMA, dropout on).  Weak scaling: per-GPU batch fixed.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, MI355X_MICROARCH.md §Chip-level parameters
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA peak (same guide)


def model_flops_fwd(N, F, D, G, L, P=3000, mlp_ratio=4.0, style=(512, 256, 128)):
    """Canonical (re-associated) forward GEMM/conv FLOPs per sample, SURVEY.md §8d."""
    h, m = 8, D // 2
    side = math.ceil(math.sqrt(N))
    nsq = side * side
    n, nr = nsq + 1, N + 1

    def layer(length):
        p = math.ceil(length / m) * m
        return (6 * p * D * D + 2 * p * m * D + 2 * m * m * D + 2 * m * p * D + 48 * h * m ** 3 + 2 * m * p * D
                + 2 * m * m * D + 2 * p * m * D + 66 * p * D + 2 * p * D * D)
    hh = int(D * mlp_ratio)
    rna = 4 * G * D + 4 * D * D + (L + 1) * (8 * D * D + 4 * D * hh) + 6 * D * D
    sty = 4 * (style[0] * D + style[0] * style[1] + 2 * style[1] * style[2] + style[2] * D + D * P)
    return 2 * N * F * D + 2 * layer(n) + 166 * nsq * D + 4 * (N + 1) * D * D + layer(nr) + 2 * D * D + rna + sty


def csrc_digest() -> str:
    """sha256 over the kernel sources: the key that ties profiles/pmc_traffic.json to the code it was measured on."""
    import hashlib
    d = os.path.join(ROOT, "mirror_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run N ranks through torch.distributed.run as a CHILD process (this
    process has made no GPU call yet and makes none: device_count() does not initialise HIP)."""
    import socket
    import subprocess
    dry = os.environ.get("MIRROR_BENCH_DIST", "") == "gloo:shared"
    have = torch.cuda.device_count()
    if have < (1 if dry else n):
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode


CONFIGS = {
    # BASELINE.json configs[1] (the metric's config) and configs[0]; "template" = the reference's own operating point,
    # configs/pretrain/mirror.template.yaml:16-46 (2048 Phikon tokens x 768-d, 10234 genes, D = 768 -> dh = 96, m = 384,
    # RNA depth 2 / 12 heads / mlp_ratio 2.572): a second bench line, not the headline
    "c2": dict(N=4096, F=1024, G=2048, D=512, L=6, heads=8, mlp=4.0),
    "c1": dict(N=256, F=1024, G=512, D=256, L=2, heads=8, mlp=4.0),
    "template": dict(N=2048, F=768, G=10234, D=768, L=2, heads=12, mlp=2.572),
    # BASELINE.json configs[3]: Phikon-dim features (768-d), 8192 patch tokens per slide, per-sample valid length ~U[2048, 8192] (seeded),
    # padded + bool key-padding mask through every Nystrom layer; a separate bench line, B = 8 unless --batch says otherwise
    "c4": dict(N=8192, F=768, G=2048, D=512, L=6, heads=8, mlp=4.0, mask=True, batch=8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 16 (the reference template's batch_size), 8 for --config c4")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16_pinv32", "fp32", "fp8"])
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="rank-local InfoNCE even when N>1 (reference behaviour)")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="wire format of the gradient buckets at N > 1 (config 5: bf16)")
    ap.add_argument("--feed", default="resident", choices=["resident", "host", "host-bf16"],
                    help="resident (default, the metric): the batch is in HBM when the timed region starts; host / host-bf16: every "
                         "step's batch comes from pinned host memory (f32 / bf16 patch features) through mirror_amd.data.HostFeeder, "
                         "copied on a side stream under the previous step — the PCIe-inclusive rate, reported as a separate line")
    a = ap.parse_args()

    # a timing tool must not be one environment variable away from a fake number: the MH_EXP_* switches make entry points
    # return without launching (results garbage) and only exist in `make EXP=1` builds of the library
    bad = sorted(k for k in os.environ if k.startswith("MH_EXP_"))
    if bad:
        raise SystemExit(f"bench.py: refusing to run with timing-experiment switches set: {', '.join(bad)}")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MIRROR_BENCH_DIST=gloo:shared runs the N > 1 code path with every rank on GPU 0 over gloo: a dry run of this file's
    # multi-rank branch on a one-GPU box (tools/bench_world2_dryrun.sh); the driver's runs use RCCL, one GPU per rank.
    dry = os.environ.get("MIRROR_BENCH_DIST", "") == "gloo:shared"
    if dry:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from mirror_amd import _lib, kernels as K
    import mirror_amd.models as M
    from mirror_amd.engine import TrainEngine
    from mirror_amd.losses import MIRRORLoss
    from mirror_amd import functional as Fn

    lib = _lib.load()
    if lib.mh_device_ok() <= 0:
        raise SystemExit("bench.py needs an MI355X: " + lib.mh_last_error().decode())

    shp = CONFIGS[a.config]
    if a.batch is None:
        a.batch = shp.get("batch", 16)
    torch.manual_seed(42)   # configs/pretrain/mirror.template.yaml:123
    model = M.mirror(wsi_embed_dim=shp["F"], rna_embed_dim=shp["G"], embed_dim=shp["D"], wsi_num_tokens=shp["N"],
                     rna_encoder_depth=shp["L"], rna_mlp_ratio=shp["mlp"], rna_norm_layer="layernorm", rna_act_layer="gelu",
                     rna_num_heads=shp["heads"]).to(dev).train()
    loss_fn = MIRRORLoss(alignment_loss_weight=0.5, wsi_retention_loss_weight=0.15, rna_retention_loss_weight=0.15,
                         style_loss_weight=0.1, cluster_loss_weight=0.1,
                         gather_distributed=(world > 1 and not a.no_gather))
    eng = TrainEngine(model, loss_fn, lr=2e-5, precision=a.precision, grad_reduce_dtype=a.grad_dtype)
    Fn.manual_seed(1234 + rank)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    in_dtype = torch.float32 if a.precision == "fp32" else torch.bfloat16
    wsi = torch.randn(a.batch, shp["N"], shp["F"], device=dev, generator=g).to(in_dtype)
    rna = torch.randn(a.batch, shp["G"], device=dev, generator=g)
    kw, lens = {}, None
    if shp.get("mask"):       # config 4: padded slides + key-padding mask (a static input of the captured step, refreshed like the batch)
        lens = torch.randint(2048, shp["N"] + 1, (a.batch,), device=dev, generator=g)
        kmask = torch.arange(shp["N"], device=dev)[None, :] < lens[:, None]
        wsi = wsi * kmask[..., None]                  # padded rows are zeros, as a collate function would leave them
        kw = {"wsi_key_padding_mask": kmask}
        if a.feed != "resident":
            raise SystemExit("bench.py: --feed host* is measured on c2 (the HostFeeder does not carry masks)")

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up; one warm-up step (the second when there is one) times every MFMA kernel launch (all GEMM template
    #      instances + the fused pinv chains) with HIP events to find the dominant kernel
    # The engine records its HIP graphs during its third step (the whole step at N = 1, the RNA branch at N > 1) after two
    # eager ones: at least three untimed steps run before the timed region whatever --warmup says, so that no capture
    # (~0.3 s, once) is ever timed; `warmup` in the JSON line is the flag, `config.untimed_steps` what actually ran.
    nwarm = max(3, a.warmup)
    summ = {}
    for i in range(nwarm):
        if i == min(1, nwarm - 1):
            torch.cuda.synchronize()
            K.gemm_profiler = K.GemmProfiler()
            eng.step(wsi, rna, **kw)
            torch.cuda.synchronize()
            summ = K.gemm_profiler.summary()
            K.gemm_profiler = None
        else:
            eng.step(wsi, rna, **kw)
    dominant = max(summ, key=lambda v: summ[v]["total_ms"])

    # ---- timed region: exactly K steps, barrier + synchronize on both sides.  On one GPU every step is ONE HIP-graph
    #      launch (TrainEngine captures the step after two eager ones), so no host code runs between its kernels.
    feed = None
    if a.feed != "resident":
        from mirror_amd.data import HostFeeder
        hdt = torch.float32 if a.feed == "host" else torch.bfloat16
        # two distinct host batches, alternated, in pinned memory as a DataLoader(pin_memory=True) hands them over
        host = [(wsi.float().cpu().to(hdt).roll(i, 0).pin_memory(), rna.cpu().roll(i, 0).pin_memory()) for i in range(2)]
        feed = HostFeeder((host[i % 2] for i in range(a.steps + 2)), dev, wsi_dtype=in_dtype, depth=3)
        it = iter(feed)
        for _ in range(2):                        # untimed: pinned buffers allocated, pipeline primed
            w_, r_ = next(it)
            eng.step(w_, r_)
    sync_all()
    t0 = time.perf_counter()
    host_us = []                                  # host time inside each step() call (MIRROR_BENCH_HOSTTIME=1 prints it to stderr)
    if feed is None:
        for _ in range(a.steps):
            h0 = time.perf_counter()
            losses = eng.step(wsi, rna, **kw)
            host_us.append((time.perf_counter() - h0) * 1e6)
    else:
        for w_, r_ in it:
            losses = eng.step(w_, r_)
    sync_all()
    dt = time.perf_counter() - t0
    if os.environ.get("MIRROR_BENCH_HOSTTIME") and rank == 0 and host_us:
        print("host us per step() call: " + " ".join(f"{u:.0f}" for u in host_us), file=sys.stderr)
    loss_vals = [float(x) for x in losses]
    # ---- roofline leg: the dominant kernel's launches carry HIP event pairs on their launch stream.  Events cannot be
    #      recorded inside a graph replay, so the same steps are run once more eagerly right here (same process, same
    #      inputs, min(K, 5) steps); profiles/ holds the rocprofv3 per-kernel averages of the graph run for comparison.
    graphed = getattr(eng, "_graph", None) is not None
    eng._use_graph = False
    K.gemm_profiler = K.GemmProfiler(only=dominant)
    for _ in range(min(a.steps, 5)):
        eng.step(wsi, rna, **kw)
    torch.cuda.synchronize()
    prof = K.gemm_profiler.summary().get(dominant, {"launches": 0, "total_ms": 0.0, "flops": 0.0})
    K.gemm_profiler = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)

    if rank == 0:
        samples = a.batch * world * a.steps
        value = samples / dt
        fl_fwd = model_flops_fwd(shp["N"], shp["F"], shp["D"], shp["G"], shp["L"], mlp_ratio=shp["mlp"])
        step_tflops = 3 * fl_fwd * a.batch * world * a.steps / dt / 1e12
        is_f32 = dominant.startswith("gemm_kernel<0")
        peak = PEAK_F32_TFLOPS if is_f32 else PEAK_BF16_TFLOPS
        avg_ms = prof["total_ms"] / max(prof["launches"], 1)
        # HBM-side bytes per launch of the dominant kernel: PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, gfx950
        # correction applied) summarised by tools/pmc_summary.py into profiles/pmc_traffic.json; null when not collected
        # the file records the digest of the kernel sources it was measured on: an entry from other code is refused (null)
        traffic, traffic_note = None, "no profiles/pmc_traffic.json entry for this kernel"
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pm = json.load(fh)
            ent = pm["kernels"].get(dominant.replace(" ", ""))
            if ent and pm.get("csrc_sha256") == csrc_digest() and pm.get("config", "c2") == a.config:
                traffic, traffic_note = ent["bytes_per_launch"], f"PMC passes on csrc {pm['csrc_sha256']} (tools/collect_profiles.sh)"
            elif ent:
                traffic_note = (f"stale: profiles/pmc_traffic.json was measured on csrc {pm.get('csrc_sha256')}, "
                                f"this run is {csrc_digest()} / config {a.config}")
        except (OSError, ValueError, KeyError):
            pass
        # step-level HBM traffic (sum over every kernel of the same PMC passes) and the dominant kernel's MFMA-pipe utilisation
        # (profiles/pmc_mfma_busy.json, its own PMC pass): both only when measured on exactly these kernel sources
        step_bytes, mfma_busy, mfma_note = None, None, "no profiles/pmc_mfma_busy.json entry for this kernel"
        step_launches = None
        try:
            if pm.get("csrc_sha256") == csrc_digest() and pm.get("config", "c2") == a.config:
                steps_prof = float(pm.get("steps", 9))
                step_bytes = sum(v["bytes_per_launch"] * v["launches"] for v in pm["kernels"].values()) / steps_prof
                if pm.get("replayed_step"):          # the replayed step alone (what is timed), not the 9-step average
                    step_bytes = float(pm["replayed_step"]["hbm_bytes"])
                    step_launches = int(pm["replayed_step"]["launches"])
        except (NameError, KeyError, TypeError):
            pass
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_mfma_busy.json")) as fh:
                mb = json.load(fh)
            key = {k.replace(" ", "").replace("unsignedshort", "bf16"): v for k, v in mb["kernels"].items()}
            ent = key.get(dominant.replace(" ", ""))
            if ent and mb.get("csrc_sha256") == csrc_digest():
                mfma_busy, mfma_note = ent["mfma_pipe_utilisation_of_chip"], f"SQ_VALU_MFMA_BUSY_CYCLES pass on csrc {mb['csrc_sha256']} (tools/pmc_mfma.sh)"
            elif ent:
                mfma_note = f"stale: measured on csrc {mb.get('csrc_sha256')}, this run is {csrc_digest()}"
        except (OSError, ValueError, KeyError):
            pass
        ach = (prof["flops"] / max(prof["launches"], 1)) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out = {
            "metric": "SSL samples/sec (slide+RNA pairs)", "value": round(value, 3), "unit": "samples/s",
            "n_gpus": world, "world_size_reported": (dist.get_world_size() if world > 1 else 1), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "feed": a.feed,
            "dtype": {"bf16": "bf16", "bf16_pinv32": "bf16", "fp32": "f32", "fp8": "fp8-fwd/bf16"}[a.precision], "data": "synthetic",
            "config": {"workload": f"{'reference template' if a.config == 'template' else 'BASELINE ' + a.config}: B={a.batch}/GPU x [{shp['N']} patch tokens x {shp['F']}-d] + "
                                   f"[{shp['G']} genes], D={shp['D']}, RNA depth {shp['L']} / {shp['heads']} heads, train mode, "
                                   f"{'global' if (world > 1 and not a.no_gather) else 'local'}-batch InfoNCE",
                       "precision_policy": a.precision, "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "untimed_steps": nwarm, "grad_bucket_dtype": a.grad_dtype,
                       **({"valid_lengths": lens.tolist(), "mask": "key-padding mask through every Nystrom layer (static input of the captured step)"} if lens is not None else {}),
                       "step_launch": "hip_graph" if graphed else (
                           "eager + graphed RNA branch" if getattr(eng, "_rna_branch_state", "") == "on" else "eager")},
            "model_tflops_per_s": round(step_tflops, 2),
            "model_flops_frac_of_bf16_peak": round(step_tflops / world / PEAK_BF16_TFLOPS, 4),
            "losses": [round(x, 5) for x in loss_vals],
            "roofline": {"bound": "mfma", "kernel": dominant, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "bound_note": ("`bound` names what limits the DOMINANT KERNEL (a chain of dependent 256^3 products on half of the "
                                        "CUs: MFMA latency, not throughput); the STEP as a whole moves step_hbm_bytes through HBM and "
                                        "is closer to the HBM roof than to the MFMA roof, see step_hbm_frac_of_6.3TBps"),
                         "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_note,
                         "step_hbm_bytes": None if step_bytes is None else int(step_bytes),
                         "step_launches": step_launches,      # dispatches of one replayed step in the same PMC passes
                         "step_hbm_frac_of_6.3TBps": None if step_bytes is None else round(step_bytes / (dt / a.steps) / 6.3e12, 4),
                         "launches_timed": prof["launches"],
                         "timed_in": ("eager re-run of min(K,5) steps right after the timed region (the timed region "
                                      "replays one HIP graph per step)") if graphed else "the timed region",
                         "avg_launch_ms": round(avg_ms, 5),
                         "share_of_mfma_kernel_time_in_profiled_step": round(summ[dominant]["total_ms"] / max(sum(s["total_ms"] for s in summ.values()), 1e-9), 3),
                         "flops_per_launch": round(prof["flops"] / max(prof["launches"], 1) / 1e9, 3),
                         "flops_unit": "GFLOP (algorithmic: 2*M*N*K*batch per GEMM; 2*m^3 per chain product)",
                         "runner_up": sorted(((round(v["total_ms"], 3), k) for k, v in summ.items()), reverse=True)[1:4]},
        }
        if world == 1 and not a.no_cpu_baseline and not shp.get("mask"):      # (the CPU leg times the unmasked path; c4 is reported without it)
            from oracle import mirror_oracle as O
            from oracle.cpu_step import time_cpu_steps
            cfg = O.Cfg(wsi_embed_dim=shp["F"], rna_embed_dim=shp["G"], embed_dim=shp["D"], wsi_num_tokens=shp["N"],
                        rna_encoder_depth=shp["L"], rna_mlp_ratio=shp["mlp"], rna_num_heads=shp["heads"])
            cb = 8 if a.config == "c1" else 2
            r = time_cpu_steps(cfg, batch=cb, budget_s=25.0)
            out["cpu_baseline"] = {"value": round(r["samples_per_s"], 4), "unit": "samples/s", "cores": r["cores"],
                                   "kind": "port",
                                   "sample": f"oracle (torch fp32 CPU restatement of the reference), same shapes, B={cb}, "
                                             f"{r['steps']} timed step(s) of fwd+loss+bwd+Adam after {r['warmup']} warm-up, "
                                             f"~25 s budget ({r['s_per_step']:.2f} s/step, {r['cores']} threads)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
