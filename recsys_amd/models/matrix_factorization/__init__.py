"""Drop-ins for recsys/models/matrix_factorization (same class names and constructor kwargs)."""
from .mf import BiasMF, SideFeaturesMF, TemporalMF, VanillaMF  # noqa: F401
from .fm import FactorizationMachine  # noqa: F401
