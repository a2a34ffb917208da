"""Physical constants (CODATA values used by the reference, em/_constants.py:1-11)."""

c: float = 299792458.0
mu_0: float = 1.25663706212e-06
epsilon_0: float = 8.8541878128e-12
z_0: float = 376.73031341259
