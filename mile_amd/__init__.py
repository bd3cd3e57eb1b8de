"""mile_amd -- MI355X-native Microcanonical Langevin Ensemble sampler (hot path of MILE).

Importing the package never touches the GPU; the HIP library is loaded on first use and
there is no CPU fallback.
"""
from mile_amd.spec import LeNetSpec, ModelSpec  # noqa: F401

__version__ = '0.1.0'
