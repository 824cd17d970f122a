from . import matrix_factorization, neuralcf  # noqa: F401
