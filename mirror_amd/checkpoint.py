"""Best-k checkpoint selection for the pre-training loop.

Mirrors what train_mirror.py does after every epoch (:1053-1062): `saver.save_checkpoint(epoch, metric=latest_metric)` with
timm's CheckpointSaver built at :920-930 (`decreasing=True` for a loss metric, `max_history=args.checkpoint_hist`): the latest
state always goes to `last.pth.tar`, the `max_history` best epochs are kept as `checkpoint-<epoch>.pth.tar`, the best one
is also copied to `model_best.pth.tar`, and (best_metric, best_epoch) comes back.  Host logic only — no kernels.
"""
from __future__ import annotations

import operator
import os
import shutil
from typing import Callable, List, Optional, Tuple

import torch


class CheckpointSaver:
    def __init__(self, model: torch.nn.Module, engine=None, *, args=None, checkpoint_dir: str = "", decreasing: bool = True,
                 max_history: int = 10, checkpoint_prefix: str = "checkpoint", extension: str = ".pth.tar"):
        self.model, self.engine, self.args = model, engine, args
        self.checkpoint_dir, self.prefix, self.ext = checkpoint_dir, checkpoint_prefix, extension
        self.decreasing, self.max_history = decreasing, max_history
        self.files: List[Tuple[str, float]] = []          # (path, metric), best first
        self.best_epoch: Optional[int] = None
        self.best_metric: Optional[float] = None
        self.cmp: Callable = operator.lt if decreasing else operator.gt
        assert max_history >= 1

    def _state(self, epoch: int, metric: Optional[float]) -> dict:
        state = {"epoch": epoch, "arch": type(self.model).__name__.lower(),
                 "state_dict": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                 "version": 2}
        if self.engine is not None:
            state["optimizer"] = self.engine.state_dict()
        if self.args is not None:
            # timm pickles the argparse.Namespace itself (train_mirror.py:920-930); a plain dict holds the same information and
            # opens under torch.load's default weights_only=True
            state["args"] = dict(vars(self.args)) if hasattr(self.args, "__dict__") else self.args
        if metric is not None:
            state["metric"] = metric
        return state

    def save_checkpoint(self, epoch: int, metric: Optional[float] = None) -> Tuple[Optional[float], Optional[int]]:
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        tmp = os.path.join(self.checkpoint_dir, "tmp" + self.ext)
        last = os.path.join(self.checkpoint_dir, "last" + self.ext)
        torch.save(self._state(epoch, metric), tmp)
        os.replace(tmp, last)
        worst = self.files[-1] if self.files else None
        if len(self.files) < self.max_history or metric is None or self.cmp(metric, worst[1]):
            if len(self.files) >= self.max_history:
                self._cleanup(1)
            path = os.path.join(self.checkpoint_dir, f"{self.prefix}-{epoch}{self.ext}")
            shutil.copyfile(last, path)
            self.files.append((path, metric))
            self.files.sort(key=lambda x: x[1] if x[1] is not None else float("inf"), reverse=not self.decreasing)
            if metric is not None and (self.best_metric is None or self.cmp(metric, self.best_metric)):
                self.best_epoch, self.best_metric = epoch, metric
                shutil.copyfile(last, os.path.join(self.checkpoint_dir, "model_best" + self.ext))
        return (None, None) if self.best_metric is None else (self.best_metric, self.best_epoch)

    def _cleanup(self, trim: int = 0) -> None:
        keep = max(self.max_history - trim, 0)
        for path, _ in self.files[keep:]:
            try:
                os.remove(path)
            except OSError:
                pass
        self.files = self.files[:keep]


def load_checkpoint_file(path: str, trusted: bool = False):
    """torch.load for a training checkpoint.  Files written by this package hold tensors and plain containers only; the
    reference's (timm CheckpointSaver) also pickle `args` as an argparse.Namespace, which torch >= 2.6 refuses under the
    default weights_only=True — it is allow-listed here.  `trusted=True` is the explicit opt-in to full unpickling
    (weights_only=False) for files that carry other Python objects: only for files you wrote yourself."""
    if trusted:
        return torch.load(path, map_location="cpu", weights_only=False)
    import argparse
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(path, map_location="cpu", weights_only=True)


def resume_checkpoint(model: torch.nn.Module, path: str, engine=None, trusted: bool = False) -> Optional[int]:
    """timm.models.resume_checkpoint as used at train_mirror.py:772-780: weights (+ optimizer state) back in, returns the
    epoch to resume from (saved epoch + 1) or None for a bare state_dict file."""
    ckpt = load_checkpoint_file(path, trusted)
    if isinstance(ckpt, dict) and "state_dict" in ckpt:
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in ckpt["state_dict"].items()}
        model.load_state_dict(sd)
        if engine is not None:
            if "optimizer" in ckpt:
                engine.load_state_dict(ckpt["optimizer"])
            else:
                engine.sync_shadows()
        return ckpt["epoch"] + 1 if "epoch" in ckpt else None
    model.load_state_dict(ckpt)
    if engine is not None:
        engine.sync_shadows()
    return None
