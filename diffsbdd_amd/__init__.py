"""diffsbdd_amd -- MI355X-native implementation of DiffSBDD's DDPM denoising hot
path (EGNN dynamics + reverse sampling loops).  See DESIGN.md.

  csrc/            hand-written gfx950 kernels + the C-ABI (include/diffsbdd_hip.h)
  _lib.py          ctypes binding (no fallback: fails loudly if the .so is missing)
  engine.py        weight packing, workspace, launch of dsbdd_dynamics_forward
  dynamics.py      EGNNDynamics            (reference: equivariant_diffusion/dynamics.py)
  en_diffusion.py  EnVariationalDiffusion  (reference: equivariant_diffusion/en_diffusion.py)
  conditional_model.py  ConditionalDDPM, SimpleConditionalDDPM
  sharding.py      one-process-per-GPU pocket sharding + RCCL gather
  pocket.py        PDB/SDF -> pocket tensors without BioPython/RDKit
"""
__version__ = "0.1.0"
