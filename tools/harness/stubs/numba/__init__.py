"""Stub of numba for importing the reference's Python in the build container (numba is absent).
njit/jit become identity decorators, so the @njit functions run as plain CPython."""
def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    def deco(f):
        return f
    return deco
njit = jit = vectorize = guvectorize = _identity
def prange(*a):
    return range(*a)
class _Types:
    def __getattr__(self, k):
        return None
types = _Types()
from . import typed  # noqa
int64 = float64 = int32 = float32 = None
