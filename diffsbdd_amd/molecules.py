"""Finished point clouds -> molecular graphs, without RDKit / OpenBabel
(SURVEY.md §8f-2).

The reference builds one RDKit molecule per sample on the host, by default
through an OpenBabel temp-file round trip (analysis/molecule_builder.py:58-98);
its distance-table path (`make_mol_edm`, :101-137, `get_bond_order_batch`,
:30-55) is the batchable one and is what this module implements: bond orders of
the whole batch in one HIP launch (`dsbdd_bond_orders`), one small device->host
copy, then plain-Python graph objects that can be written as V2000 SDF or, when
RDKit is installed, converted with `Molecule.to_rdkit()` for the reference's
`process_molecule` filters (:162-214).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .chem_tables import MAX_VALENCE, dataset_info


@dataclass
class Molecule:
    positions: np.ndarray                 # [n, 3] float32, Angstrom
    symbols: list                         # element symbols
    bonds: list = field(default_factory=list)   # (i, j, order) with i > j

    @property
    def num_atoms(self):
        return len(self.symbols)

    def fragments(self):
        """Connected components (lists of atom indices), largest first; ties
        keep the component with the smallest atom index first."""
        parent = list(range(self.num_atoms))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for i, j, _ in self.bonds:
            ri, rj = find(i), find(j)
            if ri != rj:
                parent[max(ri, rj)] = min(ri, rj)
        comp = {}
        for a in range(self.num_atoms):
            comp.setdefault(find(a), []).append(a)
        return sorted(comp.values(), key=lambda c: (-len(c), c[0]))

    def largest_fragment(self):
        """What `process_molecule(largest_frag=True)` keeps (molecule_builder.py:188-196)."""
        frags = self.fragments()
        if len(frags) <= 1:
            return self
        keep = frags[0]
        new_index = {a: k for k, a in enumerate(keep)}
        bonds = [(new_index[i], new_index[j], o) for i, j, o in self.bonds
                 if i in new_index and j in new_index]
        m = Molecule(self.positions[keep], [self.symbols[a] for a in keep], bonds)
        m.n_generated_atoms = self.num_atoms          # size of the sample the fragment was cut from
        return m

    def valences(self):
        """Sum of bond orders per atom."""
        v = [0] * self.num_atoms
        for i, j, o in self.bonds:
            v[i] += o
            v[j] += o
        return v

    def valence_violations(self):
        """Atoms that carry more bonds than their element allows (`allowed_bonds`, constants.py:19-22) -- what makes
        RDKit's sanitisation reject a molecule built from the distance tables (process_molecule(sanitize=True),
        analysis/molecule_builder.py:176-181).  Elements without an entry are not checked."""
        return [a for a, (s, v) in enumerate(zip(self.symbols, self.valences()))
                if s in MAX_VALENCE and v > MAX_VALENCE[s]]

    def to_sdf_block(self, name=""):
        """V2000 mol block + `$$$$` record separator."""
        lines = [name, "  diffsbdd_amd", ""]
        lines.append(f"{self.num_atoms:3d}{len(self.bonds):3d}  0  0  0  0  0  0  0  0999 V2000")
        for (x, y, z), s in zip(self.positions, self.symbols):
            lines.append(f"{x:10.4f}{y:10.4f}{z:10.4f} {s:<3s} 0  0  0  0  0  0  0  0  0  0  0  0")
        for i, j, o in self.bonds:
            lines.append(f"{j + 1:3d}{i + 1:3d}{o:3d}  0")
        lines += ["M  END", "$$$$"]
        return "\n".join(lines) + "\n"

    def to_rdkit(self):
        """RDKit molecule with a conformer (the object the reference's
        `build_molecule(..., add_coords=True)` returns); needs RDKit."""
        from rdkit import Chem  # noqa: WPS433  (optional dependency)
        order = {1: Chem.rdchem.BondType.SINGLE, 2: Chem.rdchem.BondType.DOUBLE,
                 3: Chem.rdchem.BondType.TRIPLE}
        mol = Chem.RWMol()
        for s in self.symbols:
            mol.AddAtom(Chem.Atom(s))
        for i, j, o in self.bonds:
            mol.AddBond(int(i), int(j), order[int(o)])
        conf = Chem.Conformer(mol.GetNumAtoms())
        for a, (x, y, z) in enumerate(self.positions):
            conf.SetAtomPosition(a, (float(x), float(y), float(z)))
        mol.AddConformer(conf)
        return mol


# What of the reference's molecule post-processing (analysis/molecule_builder.py:58-214) this package does -- printed by the
# generation / test-set CLIs and written next to their outputs, so that a run's record says it:
PROCESS_MOLECULE_COVERAGE = (
    "molecule post-processing vs the reference (analysis/molecule_builder.py): "
    "bond perception = the reference's distance-table path (make_mol_edm / get_bond_order_batch, :30-55,101-137: bit-exact, "
    "GPU kernel) -- NOT its default OpenBabel path (make_mol_openbabel, :58-98: third-party, absent here); "
    "largest_frag = exact (connected components of the bond graph); "
    "sanitize = APPROXIMATED by a valence-table + connectivity filter (RDKit's SanitizeMol is absent); "
    "relax_iter (UFF) = ABSENT (refused); add_hydrogens = ABSENT (the samplers' callers pass False). "
    "Molecule.to_rdkit() hands the same atoms / bonds to RDKit where it is installed.")


def write_sdf(path, molecules, names=None):
    with open(path, "w") as f:
        for k, m in enumerate(molecules):
            f.write(m.to_sdf_block(names[k] if names else f"mol_{k}"))


def _tables_on(device, info):
    key = (info["name"], str(device))
    cache = _tables_on.cache
    if key not in cache:
        cache[key] = tuple(torch.as_tensor(np.ascontiguousarray(info[k]), dtype=torch.float32, device=device)
                           for k in ("bonds1", "bonds2", "bonds3"))
    return cache[key]


_tables_on.cache = {}


def bond_orders(x, atom_type, lig_mask, info, n_max=None, batch=None):
    """Bond-order matrices of a batch on the GPU.

    x [N,3] float32 (Angstrom), atom_type [N] integer class ids, lig_mask [N]
    sorted sample ids (as returned by the samplers).  Returns (orders int8
    [B, n_max, n_max] strictly lower triangular, sizes int64 [B]) on x's device.
    """
    if not x.is_cuda:
        raise _lib.HipLibraryError("bond_orders needs device tensors (HIP path; no CPU fallback)")
    lib = _lib.load()
    dev = x.device
    x = x.contiguous().float()
    at = atom_type.to(device=dev, dtype=torch.int32).contiguous()
    B = int(batch) if batch is not None else (int(lig_mask.max().item()) + 1 if lig_mask.numel() else 0)
    sizes = torch.bincount(lig_mask.to(dev), minlength=B)      # (`batch`: samples that lost every atom stay in the list)
    off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(sizes, 0).to(torch.int32)
    if n_max is None:
        n_max = max(int(sizes.max().item()) if B else 0, 1)
    b1, b2, b3 = _tables_on(dev, info)
    m1, m2, m3 = info["margins"]
    out = torch.empty((B, n_max, n_max), dtype=torch.int8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.dsbdd_bond_orders(stream, x.data_ptr(), at.data_ptr(), off.data_ptr(), B, b1.shape[0],
                               b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), m1, m2, m3, n_max,
                               out.data_ptr())
    _lib.check(rc, "dsbdd_bond_orders")
    return out, sizes


def build_molecules(x, atom_type, lig_mask, info, largest_frag=False, batch=None):
    """Batch version of `build_molecule(..., use_openbabel=False)` +
    `process_molecule(largest_frag=...)` (molecule_builder.py:140-160, 162-214):
    list of `Molecule`, one per sample, in sample order."""
    if isinstance(info, str):
        info = dataset_info(info)
    orders, sizes = bond_orders(x, atom_type, lig_mask, info, batch=batch)
    orders = orders.cpu().numpy()
    sizes = sizes.cpu().numpy()
    xs = x.detach().float().cpu().numpy()
    ts = atom_type.detach().cpu().numpy()
    dec = info["atom_decoder"]
    mols, o = [], 0
    for b, n in enumerate(sizes):
        n = int(n)
        ii, jj = np.nonzero(orders[b, :n, :n])
        bonds = [(int(i), int(j), int(orders[b, i, j])) for i, j in zip(ii, jj)]
        m = Molecule(xs[o:o + n].copy(), [dec[int(t)] for t in ts[o:o + n]], bonds)
        mols.append(m.largest_fragment() if largest_frag else m)
        o += n
    return mols


def is_valid_molecule(mol, min_atoms=1, max_fragments=None, total_atoms=None, min_fragment_fraction=0.0):
    """The acceptance filter of the test-set driver when RDKit is absent (testset.TestSetDriver; the reference accepts
    what `process_molecule` returns, test.py:99-135, analysis/molecule_builder.py:162-214): a molecule built from the
    GPU bond-order matrix passes when
      * it exists and has at least `min_atoms` atoms,
      * no atom exceeds its element's valence (the failure RDKit's sanitisation reports for distance-table bonds),
      * it is connected -- after `largest_frag` it is by construction; `min_fragment_fraction` additionally asks the
        kept fragment to hold that share of the `total_atoms` the sample generated (0 = the reference's behaviour:
        any largest fragment is accepted),
      * (`max_fragments`) it has at most that many fragments when the fragments were kept (`--all_frags`)."""
    if mol is None or mol.num_atoms < min_atoms:
        return False
    if mol.valence_violations():
        return False
    if max_fragments is not None and len(mol.fragments()) > max_fragments:
        return False
    if min_fragment_fraction > 0.0:
        n_all = total_atoms if total_atoms is not None else getattr(mol, "n_generated_atoms", mol.num_atoms)
        if mol.num_atoms < min_fragment_fraction * n_all:
            return False
    return True
