#!/usr/bin/env python
"""Golden vectors for the molecule post-processing row, generated
from the REAL reference (build container only):

    python tests/golden/make_golden_chem.py

  chem_tables.npz    the reference's per-dataset bond-length matrices, margins and
                     vocabularies (constants.py:17, :95-183)
  chem_bonds.npz     random "molecules" (atoms placed at typical bonded / non-bonded
                     distances) with the reference's own get_bond_order_batch /
                     make_mol_edm bond matrices (analysis/molecule_builder.py:30-55,
                     :101-118)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_shim import import_reference_chem  # noqa: E402
from oracle import chem_oracle  # noqa: E402

const, mb = import_reference_chem()


def random_molecule(rng, n, n_types):
    """Self-avoiding random walk with bond-like steps (1.1-1.6 A) plus a few branch
    atoms: gives singles, doubles, triples and plenty of non-bonded pairs."""
    pos = [np.zeros(3)]
    while len(pos) < n:
        base = pos[rng.integers(len(pos))]
        step = rng.normal(size=3)
        step *= rng.uniform(1.05, 1.65) / np.linalg.norm(step)
        cand = base + step
        if min(np.linalg.norm(cand - p) for p in pos) > 1.0:
            pos.append(cand)
    types = rng.choice(n_types, size=n, p=np.asarray([0.55, 0.15, 0.15, 0.04, 0.01, 0.02, 0.03, 0.02, 0.01, 0.02][:n_types])
                       / sum([0.55, 0.15, 0.15, 0.04, 0.01, 0.02, 0.03, 0.02, 0.01, 0.02][:n_types]))
    return np.asarray(pos, np.float32), types.astype(np.int64)


def main():
    out = {"margins": np.asarray([const.margin1, const.margin2, const.margin3], np.float32)}
    for name, p in const.dataset_params.items():
        for k in ("bonds1", "bonds2", "bonds3"):
            out[f"{name}_{k}"] = np.asarray(p[k], np.float32)
        out[f"{name}_atom_decoder"] = np.asarray(p["atom_decoder"])
        out[f"{name}_aa_decoder"] = np.asarray(p["aa_decoder"])
    np.savez_compressed(os.path.join(HERE, "chem_tables.npz"), **out)
    print("wrote chem_tables.npz")

    rng = np.random.default_rng(7)
    info_ref = const.dataset_params["crossdock"]
    info = {k: np.asarray(info_ref[k], np.float32) for k in ("bonds1", "bonds2", "bonds3")}
    info["margins"] = (const.margin1, const.margin2, const.margin3)
    sizes = np.asarray([23, 5, 31, 1, 12, 40, 2, 18], np.int64)
    xs, ts, Es = [], [], []
    n_max = int(sizes.max())
    E_all = np.zeros((len(sizes), n_max, n_max), np.int8)
    for b, n in enumerate(sizes):
        x, t = random_molecule(rng, int(n), 10)
        xs.append(x)
        ts.append(t)
        # the reference's (X, A, E) step, molecule_builder.py:110-114
        pos = torch.from_numpy(x).unsqueeze(0)
        dists = torch.cdist(pos, pos, p=2).squeeze(0).view(-1)
        tt = torch.from_numpy(t)
        a1, a2 = torch.cartesian_prod(tt, tt).T
        E_full = mb.get_bond_order_batch(a1, a2, dists, info_ref).view(int(n), int(n))
        E_all[b, :n, :n] = torch.tril(E_full, diagonal=-1).numpy().astype(np.int8)
    x = np.concatenate(xs)
    t = np.concatenate(ts)
    mine = chem_oracle.bond_orders_dense(x, t, sizes, info, n_max)
    assert np.array_equal(mine, E_all), "oracle disagrees with the reference"
    counts = [int((E_all == k).sum()) for k in (1, 2, 3)]
    print("bond counts single/double/triple:", counts)
    assert all(c > 0 for c in counts)
    np.savez_compressed(os.path.join(HERE, "chem_bonds.npz"), x=x, atom_type=t, sizes=sizes, order=E_all)
    print("wrote chem_bonds.npz")


if __name__ == "__main__":
    main()
