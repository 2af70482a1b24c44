"""Drop-in for equivariant_diffusion/conditional_model.py of the reference."""
from diffsbdd_amd.conditional_model import ConditionalDDPM, SimpleConditionalDDPM  # noqa: F401
