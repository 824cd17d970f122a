"""Drop-ins for recsys/models/neuralcf (same class names and constructor kwargs)."""
from .ncf import GMF, MLP, NeuMF  # noqa: F401
from .wide_deep import DeepFactorizationMachine, WidenDeep  # noqa: F401
