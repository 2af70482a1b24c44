"""Drop-in for equivariant_diffusion/en_diffusion.py of the reference."""
from diffsbdd_amd.en_diffusion import (  # noqa: F401
    DistributionNodes, EnVariationalDiffusion, PredefinedNoiseSchedule)
