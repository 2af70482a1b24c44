"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float32) of the distance
based bond assignment the reference uses to turn sampled atoms into a molecular
graph:

  get_bond_order_batch   /root/reference/analysis/molecule_builder.py:30-55
  make_mol_edm (X, A, E) /root/reference/analysis/molecule_builder.py:101-118

Pinned by tests/golden/chem_*.npz, generated from the reference itself by
tests/golden/make_golden_chem.py.  Only tests/ and __graft_entry__.smoke() may
import this module; the product path is diffsbdd_amd/molecules.py (HIP).
"""
import numpy as np


def pair_distance(xi, xj):
    """float32 Euclidean distance with the operation order of the HIP kernel:
    sqrt((dx*dx + dy*dy) + dz*dz), every step rounded to float32."""
    d = (xi - xj).astype(np.float32)
    s = (d[..., 0] * d[..., 0]).astype(np.float32)
    s = (s + (d[..., 1] * d[..., 1]).astype(np.float32)).astype(np.float32)
    s = (s + (d[..., 2] * d[..., 2]).astype(np.float32)).astype(np.float32)
    return np.sqrt(s).astype(np.float32)


def bond_order(t1, t2, dist, bonds1, bonds2, bonds3, margins):
    """molecule_builder.py:30-55: distances in Angstrom are rescaled by 100 (pm);
    single, then double, then triple overwrite each other."""
    d = (np.float32(100.0) * dist.astype(np.float32)).astype(np.float32)
    m1, m2, m3 = (np.float32(m) for m in margins)
    out = np.zeros(d.shape, dtype=np.int8)
    out[d < (bonds1[t1, t2] + m1)] = 1
    out[d < (bonds2[t1, t2] + m2)] = 2
    out[d < (bonds3[t1, t2] + m3)] = 3
    return out


def bond_orders_dense(x, atom_type, sizes, info, n_max=None):
    """Per molecule b the strictly lower triangle E[b, i, j] (i > j) of the bond
    order matrix (molecule_builder.py:110-114), zero elsewhere / beyond size."""
    x = np.asarray(x, np.float32)
    atom_type = np.asarray(atom_type, np.int64)
    sizes = np.asarray(sizes, np.int64)
    n_max = int(sizes.max()) if n_max is None else n_max
    out = np.zeros((len(sizes), n_max, n_max), dtype=np.int8)
    off = 0
    b1, b2, b3 = (np.asarray(info[k], np.float32) for k in ("bonds1", "bonds2", "bonds3"))
    for b, n in enumerate(sizes):
        xs, ts = x[off:off + n], atom_type[off:off + n]
        dist = pair_distance(xs[:, None, :], xs[None, :, :])
        full = bond_order(ts[:, None], ts[None, :], dist, b1, b2, b3, info["margins"])
        out[b, :n, :n] = np.tril(full, -1)
        off += n
    return out
