"""Load tests/golden/*.npz cases (produced by tools/make_golden.py from the reference)."""
import ast
import os

import numpy as np
import torch

from oracle import synth
from oracle.mirror_oracle import Cfg, OUTPUT_NAMES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TEMPLATE_W = (0.5, 0.15, 0.15, 0.1, 0.1)   # configs/pretrain/mirror.template.yaml:104-110
DEFAULT_W = (0.5, 0.1, 0.1, 0.1, 0.2)      # losses/mirror_loss.py:59-63


class ModelCase:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f"golden_model_{name}.npz"), allow_pickle=False)
        self.z = z
        self.name = name
        self.cfg = Cfg(**ast.literal_eval(str(z["cfg_json"])))
        self.batch = int(z["batch"])
        self.seed = int(z["seed"])
        self.ratios = tuple(float(x) for x in z["ratios"])
        shapes = synth.param_shapes(self.cfg)
        self.keys = [k for k, _ in shapes]
        if f"sd/{self.keys[0]}" in z.files:
            self.sd = {k: torch.from_numpy(z[f"sd/{k}"]) for k in self.keys}
            self.wsi, self.rna = torch.from_numpy(z["in/wsi"]), torch.from_numpy(z["in/rna"])
            self.noise = {k: torch.from_numpy(z[f"noise/{k}"]) for k in ("wsi_mask", "rna_mask", "wsi_eps", "rna_eps")}
        else:
            self.sd = synth.synth_state_dict(shapes, self.seed)
            self.wsi, self.rna, self.noise = synth.synth_batch(self.cfg, self.batch, self.seed + 1000)
        # the regenerated tensors must be the ones the golden outputs were recorded for
        cs = synth.checksum([self.sd[k] for k in self.keys])
        assert abs(cs - float(z["sd_checksum"])) <= 1e-6 * max(1.0, abs(cs)), "state-dict regeneration drifted"
        ci = synth.checksum([self.wsi, self.rna] + [self.noise[k] for k in sorted(self.noise)])
        assert abs(ci - float(z["in_checksum"])) <= 1e-6 * max(1.0, abs(ci)), "input regeneration drifted"

    def expected_output(self, nm):
        return self.z[f"out_idx/{nm}"], self.z[f"out_val/{nm}"], self.z[f"out_sum/{nm}"]

    def check_outputs(self, outs, rtol, atol_scale=1.0):
        """outs: 15 tensors (any device). Returns dict name -> max scaled error."""
        errs = {}
        for nm, o in zip(OUTPUT_NAMES, outs):
            idx, val, _ = self.expected_output(nm)
            got = o.detach().float().cpu().flatten().numpy()[idx]
            scale = max(float(np.abs(val).max()), 1e-6)
            err = float(np.abs(got - val).max()) / scale
            errs[nm] = err
            assert err <= rtol * atol_scale, f"{self.name}:{nm} max-abs err {err:.3e} (rel. to max |ref| {scale:.3e}) > {rtol}"
        return errs
