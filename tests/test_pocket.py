"""CPU: the BioPython-free pocket featuriser against the
fixture derived from the reference's example/3rfm.pdb and a hand-written PDB."""
import os

import numpy as np
import torch

from diffsbdd_amd import pocket as P
from tests._golden import GOLDEN_DIR

PDB = """\
HEADER    TEST
ATOM      1  N   ALA A   1       0.000   0.000   0.000  1.00  0.00           N
ATOM      2  CA  ALA A   1       1.458   0.000   0.000  1.00  0.00           C
ATOM      3  C   ALA A   1       2.009   1.420   0.000  1.00  0.00           C
ATOM      4  O   ALA A   1       1.251   2.390   0.000  1.00  0.00           O
ATOM      5  HA  ALA A   1       1.800  -0.500   0.900  1.00  0.00           H
ATOM      6  N   CYS A   2       3.332   1.540   0.000  1.00  0.00           N
ATOM      7  CA ACYS A   2       3.970   2.850   0.000  0.50  0.00           C
ATOM      8  CA BCYS A   2       3.990   2.870   0.010  0.50  0.00           C
ATOM      9  SG  CYS A   2       5.700   2.700   0.300  1.00  0.00           S
ATOM     10  CA  GLY A   3      40.000  40.000  40.000  1.00  0.00           C
HETATM   11  O   HOH A 101       2.000   2.000   2.000  1.00  0.00           O
ATOM     12  CA  UNK A   4       2.500   2.500   0.500  1.00  0.00           C
END
"""


def test_fixed_column_pdb_reader_and_pocket_selection(tmp_path):
    f = tmp_path / "t.pdb"
    f.write_text(PDB)
    res = P.read_pdb_residues(str(f))
    assert [r["resname"] for r in res] == ["ALA", "CYS", "GLY", "UNK"]       # HETATM skipped
    assert len(res[1]["atoms"]) == 3                                          # altloc B dropped
    lig = np.array([[2.0, 1.0, 0.0]], dtype=np.float32)
    sel = P.pocket_residues_from_ligand(res, lig, dist_cutoff=8.0)
    assert [r["resname"] for r in sel] == ["ALA", "CYS"]                     # GLY too far, UNK not an amino acid
    x, t, n = P.featurize_pocket(sel, "CA")
    assert x.shape == (2, 3) and n == 20 and t.tolist() == [P.AA_ENCODER["A"], P.AA_ENCODER["C"]]
    x, t, n = P.featurize_pocket(sel, "full-atom")
    assert x.shape == (7, 3) and n == 10                                      # hydrogens dropped
    assert t.tolist() == [1, 0, 0, 2, 1, 0, 3]                                # N C C O | N C S
    pk = P.prepare_pocket(x, t, n, repeats=3)
    assert pk["x"].shape == (21, 3) and pk["one_hot"].shape == (21, 10)
    assert pk["size"].tolist() == [7, 7, 7] and pk["mask"].tolist() == [0] * 7 + [1] * 7 + [2] * 7
    assert pk["mask"].dtype == torch.int64 and pk["x"].dtype == torch.float32


def test_benchmark_fixture_sizes():
    """3rfm pocket = 36 residues -> 36 CA nodes / 286 heavy atoms; 5ndu 33 / 287."""
    z = np.load(os.path.join(GOLDEN_DIR, "pocket_3rfm.npz"))
    assert int(z["n_residues"]) == 36 and z["ca_x"].shape == (36, 3) and z["fa_x"].shape == (286, 3)
    assert set(z["fa_types"].tolist()) <= {0, 1, 2, 3}
    z5 = np.load(os.path.join(GOLDEN_DIR, "pocket_5ndu.npz"))
    assert int(z5["n_residues"]) == 33 and z5["fa_x"].shape == (287, 3)
    pkg = np.load(os.path.join(os.path.dirname(GOLDEN_DIR), "..", "diffsbdd_amd", "data", "pocket_3rfm.npz"))
    np.testing.assert_array_equal(pkg["fa_x"], z["fa_x"])                    # bench fixture == golden fixture
