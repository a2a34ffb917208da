"""differt_amd -- MI355X-native ray-tracing core behind DiffeRT's operator signatures.

Only the hot path is here: ray/triangle operators, the image-method tracer
with its VJP, and path-candidate enumeration; arithmetic lives in ``csrc/*.hip`` behind the C ABI
``include/differt_amd.h``.  There is no CPU fallback: without the HIP library every call raises.
"""

__version__ = "0.1.0"
