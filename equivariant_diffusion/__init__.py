"""Drop-in replacement package for the reference's `equivariant_diffusion`
(same module and class names, so `lightning_modules.py:18-21` imports resolve
here unchanged).  The implementation lives in `diffsbdd_amd`."""
