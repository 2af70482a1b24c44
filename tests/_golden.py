"""Loader for the committed golden vectors (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

from oracle import weights as W

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DYN_CASES = ["dyn_small_cond", "dyn_small_joint", "dyn_small_variant", "dyn_ca_cond",
             "dyn_fullatom_cond", "dyn_fullatom_joint"]


class Case:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.z = z
        self.cfg = json.loads(str(z["cfg_json"])) if "cfg_json" in z else None
        self.ddpm = json.loads(str(z["ddpm_json"])) if "ddpm_json" in z else None
        if self.ddpm is not None:
            self.ddpm["norm_values"] = tuple(self.ddpm["norm_values"])

    def t(self, key, dtype=None):
        v = torch.from_numpy(np.asarray(self.z[key]))
        return v if dtype is None else v.to(dtype)

    def has(self, key):
        return key in self.z

    def state_dict(self):
        """Weights: stored in the file if present, else regenerated from the
        seed and verified against the stored checksum."""
        keys = [k for k in self.z.files if k.startswith("w:")]
        if keys:
            sd = {k[2:]: torch.from_numpy(self.z[k]) for k in keys}
        else:
            sd = W.random_state_dict(self.cfg, seed=int(self.z["seed"]))
        chk = W.state_dict_checksum(sd)
        assert chk == str(self.z["checksum"]), \
            f"{self.name}: weight checksum {chk} != golden {self.z['checksum']}"
        return sd

    def noise(self, prefix="noise_", count_key="n_draws"):
        n = int(self.z[count_key])
        return [torch.from_numpy(self.z[f"{prefix}{i}"]) for i in range(n)]

    def steps(self):
        n = int(self.z["n_steps"])
        return [{k: torch.from_numpy(self.z[f"step{i}_{k}"]) for k in ("s", "t", "zt", "pt", "zs", "ps")}
                for i in range(n)]

    def pocket(self, prefix="pocket_"):
        return {"x": self.t(prefix + "x"), "one_hot": self.t(prefix + "one_hot"),
                "size": self.t(prefix + "size"), "mask": self.t(prefix + "mask")}
