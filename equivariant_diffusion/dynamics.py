"""Drop-in for equivariant_diffusion/dynamics.py of the reference."""
from diffsbdd_amd.dynamics import EGNNDynamics  # noqa: F401
