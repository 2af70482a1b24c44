"""Chemistry tables of the sampling front/back end: atom / residue
vocabularies of the three shipped datasets and the distance tables that turn a
generated point cloud into bonds.

The numbers are the ones the reference ships in /root/reference/constants.py
(:17 margins, :28-69 bond lengths in pm, :95-183 per-dataset encoders); they are
stored here once, per unordered element pair, and expanded to the dense
`bonds1/2/3` matrices of a dataset's atom decoder on demand (the reference
stores the expanded matrices per dataset).  tests/test_chem.py checks the
expansion against the reference's own matrices (golden file).
"""
from __future__ import annotations

import numpy as np

# single / double / triple bond length in pm per unordered pair (0 = no such bond)
_PAIR_PM = """
C C 154 134 120 | C N 147 129 116 | C O 143 120 113 | C S 182 160 0 | C Br 194 0 0
C Cl 177 0 0    | C P 184 0 0     | C I 214 0 0     | C F 135 0 0
N N 145 125 110 | N O 140 121 0   | N S 168 0 0     | N Br 214 0 0   | N Cl 175 0 0
N P 177 0 0     | N I 222 0 0     | N F 136 0 0
O O 148 121 0   | O S 151 0 0     | O Br 172 0 0    | O Cl 164 0 0   | O P 163 150 0
O I 194 0 0     | O F 142 0 0
S S 204 0 0     | S Br 225 0 0    | S Cl 207 0 0    | S P 210 186 0  | S I 234 0 0 | S F 158 0 0
B Cl 175 0 0
Br Br 228 0 0   | Br Cl 214 0 0   | Br P 222 0 0    | Br F 178 0 0
Cl Cl 199 0 0   | Cl P 203 0 0    | Cl F 166 0 0
P P 221 0 0     | P F 156 0 0
I I 266 0 0     | I F 187 0 0
F F 142 0 0
"""
MARGINS_PM = (3.0, 2.0, 1.0)   # margin1, margin2, margin3 (constants.py:17)

# highest number of covalent bonds (sum of bond orders) an element takes: the reference's `allowed_bonds`
# (constants.py:19-22; where it lists alternatives -- P, Hg, Bi -- the largest one)
MAX_VALENCE = {"H": 1, "C": 4, "N": 3, "O": 2, "F": 1, "B": 3, "Al": 3, "Si": 4, "P": 5, "S": 4, "Cl": 1, "As": 3,
               "Br": 1, "I": 1, "Hg": 2, "Bi": 5}

_LIGAND_ELEMENTS = ["C", "N", "O", "S", "B", "Br", "Cl", "P", "I", "F"]
_AMINO_ACIDS = list("ACDEFGHIKLMNPQRSTVWY")
# dataset -> (ligand atom decoder, pocket decoder for the CA representation)
_DATASETS = {
    "crossdock": (_LIGAND_ELEMENTS, _AMINO_ACIDS),
    "bindingmoad": (_LIGAND_ELEMENTS, _AMINO_ACIDS),
    # full-atom CrossDocked variant with an explicit rest class; its "aa" vocabulary is atoms too
    "crossdock_full": (_LIGAND_ELEMENTS + ["others"], _LIGAND_ELEMENTS + ["others"]),
}


def _pair_table():
    out = {}
    for item in _PAIR_PM.replace("\n", "|").split("|"):
        tok = item.split()
        if not tok:
            continue
        a, b = tok[0], tok[1]
        out[frozenset((a, b))] = tuple(float(v) for v in tok[2:5])
    return out


_PAIRS = _pair_table()


def bond_matrices(decoder):
    """Dense [A][A] float32 single/double/triple tables (pm) for an atom decoder."""
    n = len(decoder)
    m = np.zeros((3, n, n), dtype=np.float32)
    for i, a in enumerate(decoder):
        for j, b in enumerate(decoder):
            v = _PAIRS.get(frozenset((a, b)))
            if v is not None:
                m[:, i, j] = v
    return m[0], m[1], m[2]


def dataset_info(name):
    """The subset of the reference's `dataset_params[name]` that sampling needs."""
    if name not in _DATASETS:
        raise KeyError(f"unknown dataset {name!r} (have {sorted(_DATASETS)})")
    atoms, aas = _DATASETS[name]
    b1, b2, b3 = bond_matrices(atoms)
    return {
        "name": name,
        "atom_decoder": list(atoms), "atom_encoder": {a: i for i, a in enumerate(atoms)},
        "aa_decoder": list(aas), "aa_encoder": {a: i for i, a in enumerate(aas)},
        "bonds1": b1, "bonds2": b2, "bonds3": b3,
        "margins": MARGINS_PM,
    }
