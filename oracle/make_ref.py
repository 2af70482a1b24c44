"""TEST / MEASUREMENT INFRASTRUCTURE -- recipe for `oracle/_ref/` (git-ignored, travels to the GPU box with the push like
the built .so; VERDICT r3 "missing" #5).

The reference is pure Python: there is nothing to compile.  What `bench.py`'s `cpu_baseline` leg needs in order to time
THE REFERENCE ITSELF (kind "reference") on the GPU host -- where /root/reference does not exist -- is the reference's
own modules for this path, bit for bit.  This recipe copies them, unmodified, from where they lie under /root/reference
into ONE archive, oracle/_ref/reference_path.zip, at `__graft_entry__.build()` time (in the build container):

    equivariant_diffusion/{egnn_new,dynamics,en_diffusion,conditional_model}.py   utils.py   MANIFEST.json (SHA-256)
    + (round 5) lightning_modules.py constants.py dataset.py analysis/molecule_builder.py example/3rfm.{pdb,sdf} LICENSE

Nothing under oracle/_ref/ is committed (`.gitignore`), nothing in the product path reads it; `oracle/ref_shim.py`
imports the modules straight from the archive (zipimport; third-party stubs only) when DIFFSBDD_REFERENCE points at it.
"""
import hashlib
import json
import os
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(DEST, "reference_path.zip")
SRC = os.environ.get("DIFFSBDD_REFERENCE_SRC", "/root/reference")
FILES = ["equivariant_diffusion/egnn_new.py", "equivariant_diffusion/dynamics.py",
         "equivariant_diffusion/en_diffusion.py", "equivariant_diffusion/conditional_model.py", "utils.py"]
# round 5: the reference's CALLERS of the path and what they import, so that the GPU suite can run
# `LigandPocketDDPM.generate_ligands` / `training_step` -- the reference's own code, unchanged -- against the drop-in
# package on the GPU box (tests/test_gpu_reference_caller.py through oracle/ref_caller_shim.py); the two example inputs
# they are run on; the reference's licence (MIT) travels with the copy.
CALLER_FILES = ["lightning_modules.py", "constants.py", "dataset.py", "analysis/molecule_builder.py",
                "example/3rfm.pdb", "example/3rfm_B_CFF.sdf", "LICENSE"]


def available():
    if not os.path.isfile(ARCHIVE):
        return False
    try:
        with zipfile.ZipFile(ARCHIVE) as z:
            return all(f in z.namelist() for f in FILES)
    except zipfile.BadZipFile:
        return False


def callers_available():
    if not os.path.isfile(ARCHIVE):
        return False
    try:
        with zipfile.ZipFile(ARCHIVE) as z:
            return all(f in z.namelist() for f in FILES + CALLER_FILES)
    except zipfile.BadZipFile:
        return False


def extract_examples(dest):
    """The example inputs (3rfm.pdb, 3rfm_B_CFF.sdf) out of the archive into `dest`; returns their paths."""
    out = []
    with zipfile.ZipFile(ARCHIVE) as z:
        for f in ("example/3rfm.pdb", "example/3rfm_B_CFF.sdf"):
            path = os.path.join(dest, os.path.basename(f))
            with open(path, "wb") as fh:
                fh.write(z.read(f))
            out.append(path)
    return out


def make(verbose=True):
    if not os.path.isfile(os.path.join(SRC, FILES[0])):
        if verbose:
            print(f"[make_ref] {SRC} not present: oracle/_ref left as it is ({'complete' if available() else 'absent'})")
        return available()
    os.makedirs(DEST, exist_ok=True)
    manifest = {}
    with zipfile.ZipFile(ARCHIVE, "w", zipfile.ZIP_DEFLATED) as z:
        for f in FILES + CALLER_FILES:
            data = open(os.path.join(SRC, f), "rb").read()
            manifest[f] = hashlib.sha256(data).hexdigest()
            z.writestr(zipfile.ZipInfo(f, date_time=(2020, 1, 1, 0, 0, 0)), data, zipfile.ZIP_DEFLATED)
        z.writestr("MANIFEST.json", json.dumps({"source": SRC, "upstream": "github.com/arneschneuing/DiffSBDD (MIT licence, see LICENSE)",
                                                "note": "unmodified copies; test / measurement infrastructure only, git-ignored",
                                                "sha256": manifest}, indent=1))
    if verbose:
        print(f"[make_ref] {len(FILES) + len(CALLER_FILES)} reference files -> {ARCHIVE}")
    return True


if __name__ == "__main__":
    make()
