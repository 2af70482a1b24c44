"""TEST INFRASTRUCTURE ONLY -- import the *reference* DiffSBDD modules from
/root/reference inside this (GPU-less) container so that

  * oracle/ (our CPU restatement) can be validated against the real thing, and
  * tests/golden/make_golden.py can generate the committed golden vectors.

/root/reference does not exist on the GPU box, so nothing in `-m gpu` tests or
smoke() may import this module; bench.py's `cpu_baseline` leg uses it with
DIFFSBDD_REFERENCE = oracle/_ref/reference_path.zip (oracle/make_ref.py: the
reference's own modules of this path, shipped with the push) to time the
reference itself on the GPU host.  It copies no reference source; it
only provides stand-ins for third-party modules that are missing in this image
(SURVEY.md §8c):

  torch_scatter.scatter_add / scatter_mean   (en_diffusion.py:8, conditional_model.py:6)
  rdkit, Bio, networkx, openbabel             (utils.py:6-9, import-time only)
"""
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("DIFFSBDD_REFERENCE", "/root/reference")


def reference_available() -> bool:
    if REF_ROOT.endswith(".zip"):      # oracle/_ref/reference_path.zip (oracle/make_ref.py): imported through zipimport
        import zipfile
        try:
            with zipfile.ZipFile(REF_ROOT) as z:
                return "equivariant_diffusion/egnn_new.py" in z.namelist()
        except (OSError, zipfile.BadZipFile):
            return False
    return os.path.isfile(os.path.join(REF_ROOT, "equivariant_diffusion", "egnn_new.py"))


def _scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return res.index_add_(0, index, src)


def _scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    s = _scatter_add(src, index, dim=0, dim_size=dim_size)
    cnt = torch.zeros(s.shape[0], dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return s / cnt.view((-1,) + (1,) * (s.dim() - 1))


def _install_stubs():
    if "torch_scatter" not in sys.modules:
        m = types.ModuleType("torch_scatter")
        m.scatter_add = _scatter_add
        m.scatter_mean = _scatter_mean
        sys.modules["torch_scatter"] = m

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    class _Dummy:  # placeholder for names imported at module import time
        def __init__(self, *a, **k):
            pass

    rd = stub("rdkit")
    chem = stub("rdkit.Chem", BondType=types.SimpleNamespace(
        SINGLE=1, DOUBLE=2, TRIPLE=3, AROMATIC=4))
    rd.Chem = chem
    try:
        import networkx  # noqa: F401  (present in this image via torch deps)
        import networkx.algorithms  # noqa: F401
    except Exception:
        nx = stub("networkx")
        alg = stub("networkx.algorithms", isomorphism=types.SimpleNamespace())
        nx.algorithms = alg
    bio = stub("Bio")
    pdb = stub("Bio.PDB")
    poly = stub("Bio.PDB.Polypeptide", is_aa=lambda *a, **k: True,
                three_to_one=lambda x: x)
    bio.PDB = pdb
    pdb.Polypeptide = poly
    pdb.PDBParser = _Dummy


def import_reference():
    """Returns (dynamics_mod, en_diffusion_mod, conditional_model_mod, egnn_mod)
    of the reference.  Never writes into /root/reference (no bytecode)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    # The repo root contains a drop-in package with the same name
    # (`equivariant_diffusion`).  Make sure the reference one wins here.
    for name in [n for n in sys.modules if n == "equivariant_diffusion"
                 or n.startswith("equivariant_diffusion.") or n == "utils"]:
        del sys.modules[name]
    # The reference's `equivariant_diffusion` directory has no __init__.py (a
    # namespace package), so a regular package of the same name anywhere on
    # sys.path (ours) would win.  Pin the package object to the reference dir.
    pkg = types.ModuleType("equivariant_diffusion")
    pkg.__path__ = [os.path.join(REF_ROOT, "equivariant_diffusion")]
    sys.modules["equivariant_diffusion"] = pkg
    sys.path.insert(0, REF_ROOT)
    try:
        import equivariant_diffusion.egnn_new as egnn_mod
        import equivariant_diffusion.en_diffusion as en_mod
        import equivariant_diffusion.dynamics as dyn_mod
        import equivariant_diffusion.conditional_model as cond_mod
    finally:
        sys.path.remove(REF_ROOT)
    # keep them importable under private names, free the public names again
    mods = (dyn_mod, en_mod, cond_mod, egnn_mod)
    for name in [n for n in sys.modules if n == "equivariant_diffusion"
                 or n.startswith("equivariant_diffusion.") or n == "utils"]:
        sys.modules["_ref_" + name] = sys.modules.pop(name)
    return mods


def import_reference_chem():
    """Returns (constants_mod, molecule_builder_mod) of the reference
    (/root/reference/constants.py, analysis/molecule_builder.py) for the golden
    vectors of the molecule post-processing row.  RDKit /
    OpenBabel are absent here: only the pure-torch functions
    (get_bond_order_batch, the tables) are usable."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    chem = sys.modules["rdkit.Chem"]
    bt = types.SimpleNamespace(SINGLE=1, DOUBLE=2, TRIPLE=3, AROMATIC=12)
    if not hasattr(chem, "rdchem"):
        chem.rdchem = types.SimpleNamespace(BondType=bt)
    if "rdkit.Chem.rdForceFieldHelpers" not in sys.modules:
        ff = types.ModuleType("rdkit.Chem.rdForceFieldHelpers")
        ff.UFFOptimizeMolecule = lambda *a, **k: 0
        ff.UFFHasAllMoleculeParams = lambda *a, **k: True
        sys.modules["rdkit.Chem.rdForceFieldHelpers"] = ff
        chem.rdForceFieldHelpers = ff
    if "openbabel" not in sys.modules:
        sys.modules["openbabel"] = types.ModuleType("openbabel")
    import importlib.util
    saved = {n: sys.modules.pop(n) for n in ("utils", "constants") if n in sys.modules}
    sys.path.insert(0, REF_ROOT)
    try:
        import constants as const_mod
        spec = importlib.util.spec_from_file_location(
            "_ref_molecule_builder", os.path.join(REF_ROOT, "analysis", "molecule_builder.py"))
        mb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mb)
    finally:
        sys.path.remove(REF_ROOT)
        for n in ("utils", "constants"):
            if n in sys.modules:
                sys.modules["_ref_" + n] = sys.modules.pop(n)
        sys.modules.update(saved)
    return const_mod, mb
