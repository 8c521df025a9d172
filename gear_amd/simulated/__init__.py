"""Mirror of GenerationBench/GenerationTest/GEARLM/Simulated/{compress_function,compress_config}.py on HIP kernels."""
from .compress_config import CompressionConfig  # noqa: F401
from .compress_function import compress_insert_function  # noqa: F401
