"""Pocket featuriser without BioPython / RDKit.

Host-side input plumbing for the sampling hot path: PDB file -> the
`{'x','one_hot','size','mask'}` pocket dict the DDPM classes consume.
Mirrors what the reference does with BioPython in
  utils.get_pocket_from_ligand      (/root/reference/utils.py:103-128)
  LigandPocketDDPM.prepare_pocket   (/root/reference/lightning_modules.py:714-752)
but with a fixed-column PDB reader and a V2000 SDF coordinate reader.
"""
from __future__ import annotations

import numpy as np
import torch

# residue-type / atom-type vocabularies (integers are all that reach the hot
# path): /root/reference/constants.py:154-183
AA3_TO_1 = {
    "ALA": "A", "CYS": "C", "ASP": "D", "GLU": "E", "PHE": "F", "GLY": "G", "HIS": "H",
    "ILE": "I", "LYS": "K", "LEU": "L", "MET": "M", "ASN": "N", "PRO": "P", "GLN": "Q",
    "ARG": "R", "SER": "S", "THR": "T", "VAL": "V", "TRP": "W", "TYR": "Y",
}
AA_ENCODER = {a: i for i, a in enumerate("ACDEFGHIKLMNPQRSTVWY")}
ATOM_ENCODER = {a: i for i, a in enumerate(
    ["C", "N", "O", "S", "B", "Br", "Cl", "P", "I", "F"])}


def read_sdf_coords(path):
    """Coordinates of the first molecule of a V2000 SDF file -> float32 [n,3]."""
    with open(path) as f:
        lines = f.read().splitlines()
    counts = lines[3]
    n_atoms = int(counts[0:3])
    xyz = [[float(l[0:10]), float(l[10:20]), float(l[20:30])] for l in lines[4:4 + n_atoms]]
    return np.asarray(xyz, dtype=np.float32)


def read_pdb_residues(path, model=0, hetero=False):
    """Fixed-column PDB reader (first model, ATOM records, altloc ' '/'A').
    Returns a list of residues in file order; each residue is a dict
    {'chain','resseq','icode','resname','atoms': [(name, element, xyz)]}.
    hetero=True also returns HETATM groups (flagged 'hetero': True) so that a
    ligand can be addressed as <chain>:<resi> (utils.py:111-116)."""
    residues, index = [], {}
    with open(path) as f:
        for line in f:
            rec = line[0:6]
            if rec.startswith("ENDMDL"):
                break
            het = rec.startswith("HETATM")
            if not (rec.startswith("ATOM") or (hetero and het)):
                continue
            if line[16] not in (" ", "A"):
                continue
            name = line[12:16].strip()
            resname = line[17:20].strip()
            chain = line[21]
            resseq = int(line[22:26])
            icode = line[26]
            xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            elem = line[76:78].strip() if len(line) >= 78 else ""
            if not elem:
                elem = "".join(c for c in name if c.isalpha())[:1]
            key = (chain, resseq, icode, het)
            if key not in index:
                index[key] = len(residues)
                residues.append(dict(chain=chain, resseq=resseq, icode=icode,
                                     resname=resname, atoms=[], hetero=het))
            residues[index[key]]["atoms"].append((name, elem.capitalize(), xyz))
    return residues


def pocket_residues_from_ligand(residues, ligand_xyz, dist_cutoff=8.0):
    """Standard amino-acid residues with any atom closer than `dist_cutoff`
    to any ligand atom (utils.py:118-126)."""
    lig = np.asarray(ligand_xyz, dtype=np.float32)
    out = []
    for res in residues:
        if res["resname"] not in AA3_TO_1:
            continue
        xyz = np.asarray([a[2] for a in res["atoms"]], dtype=np.float32)
        d = np.sqrt(((xyz[:, None, :] - lig[None, :, :]) ** 2).sum(-1))
        if d.min() < dist_cutoff:
            out.append(res)
    return out


def featurize_pocket(residues, representation="CA", atom_encoder=None):
    """-> (coords float32 [n,3], type ids int64 [n], n_types)
    (lightning_modules.py:716-733)."""
    if representation == "CA":
        coords, types = [], []
        for res in residues:
            ca = [a for a in res["atoms"] if a[0] == "CA"]
            if not ca:
                continue
            coords.append(ca[0][2])
            types.append(AA_ENCODER[AA3_TO_1[res["resname"]]])
        return (np.asarray(coords, np.float32), np.asarray(types, np.int64), len(AA_ENCODER))
    enc = ATOM_ENCODER if atom_encoder is None else atom_encoder
    coords, types = [], []
    for res in residues:
        for _, elem, xyz in res["atoms"]:
            if elem in enc:
                coords.append(xyz)
                types.append(enc[elem])
            elif elem != "H":
                # the reference would raise KeyError here as well
                raise KeyError(f"pocket atom type {elem!r} not in encoder")
    return np.asarray(coords, np.float32), np.asarray(types, np.int64), len(enc)


def prepare_pocket(coords, types, n_types, repeats=1, device="cpu"):
    """Pocket dict for `repeats` identical copies
    (lightning_modules.py:735-752)."""
    coords = torch.as_tensor(coords, dtype=torch.float32, device=device)
    types = torch.as_tensor(types, dtype=torch.int64, device=device)
    n = coords.shape[0]
    one_hot = torch.nn.functional.one_hot(types, num_classes=n_types)
    return {
        "x": coords.repeat(repeats, 1),
        "one_hot": one_hot.repeat(repeats, 1),
        "size": torch.full((repeats,), n, dtype=torch.int64, device=device),
        "mask": torch.repeat_interleave(
            torch.arange(repeats, dtype=torch.int64, device=device), n),
    }


def pocket_from_files(pdb_file, ligand_sdf, representation="CA", repeats=1, device="cpu",
                      dist_cutoff=8.0):
    residues = read_pdb_residues(pdb_file)
    lig = read_sdf_coords(ligand_sdf)
    sel = pocket_residues_from_ligand(residues, lig, dist_cutoff)
    coords, types, nt = featurize_pocket(sel, representation)
    return prepare_pocket(coords, types, nt, repeats, device)
