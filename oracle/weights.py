"""ORACLE support: the synthetic-weight generator lives in the package
(diffsbdd_amd/synthetic.py -- plain data generation, no arithmetic of the hot
path) so that bench.py's GPU leg does not import from oracle/."""
from diffsbdd_amd.synthetic import (arch_cfg, dynamics_param_shapes, random_state_dict,  # noqa: F401
                                    state_dict_checksum)
