"""Mirror of the reference package cuda_supported_gear/quant (new_pack.py, matmul.py) on HIP kernels."""
from . import matmul, new_pack  # noqa: F401
