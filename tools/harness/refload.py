"""Import the reference's Python modules IN PLACE from /root/reference (build container only).

Nothing from /root/reference is copied: a temporary package directory holding SYMLINKS to the
reference files is put on sys.path together with the stub modules in ./stubs (numba, edlib, pysam,
Bio, cigar are absent from this image). Bytecode writing is disabled. See SURVEY.md Appendix B.

np.argsort is patched to kind='stable' while the reference runs (SURVEY §8(a) note T1): the
reference's default quicksort is not stable and tie order is implementation-defined; the build
defines stable order as normative.
"""
import os, sys, tempfile, importlib

sys.dont_write_bytecode = True
REF = '/root/reference'
_HERE = os.path.dirname(os.path.abspath(__file__))
_pkgroot = None

MODE_MODULE = {'H': 'mammap_clrnano', 'L': 'mammap_ccs', 'S': 'mammap_sensitive', 'R': 'mammap_noprefercloser', 'asm': 'mammap_asm'}


def _setup():
    global _pkgroot
    if _pkgroot is not None:
        return
    if not os.path.isdir(REF):
        raise RuntimeError('reference not mounted at %s (harness runs only in the build container)' % REF)
    _pkgroot = tempfile.mkdtemp(prefix='vacref_')
    pk = os.path.join(_pkgroot, 'vacmap')
    os.mkdir(pk)
    for fn in os.listdir(os.path.join(REF, 'src', 'vacmap')):
        if fn.endswith('.py'):
            os.symlink(os.path.join(REF, 'src', 'vacmap', fn), os.path.join(pk, fn))
    sys.path.insert(0, _pkgroot)
    sys.path.insert(0, os.path.join(_HERE, 'stubs'))
    import numpy as np
    _orig = np.argsort

    def stable_argsort(a, axis=-1, kind=None, order=None, **kw):
        return _orig(a, axis=axis, kind='stable', order=order, **kw)
    np.argsort = stable_argsort


def load(mode='H', vacmap_index_module=None):
    """returns the imported mode module. vacmap_index_module: object to register as `vacmap_index`."""
    _setup()
    if vacmap_index_module is not None:
        sys.modules['vacmap_index'] = vacmap_index_module
    elif 'vacmap_index' not in sys.modules:
        import types
        sys.modules['vacmap_index'] = types.ModuleType('vacmap_index')
    import logging
    logging.disable(logging.CRITICAL)
    import matplotlib
    matplotlib.use('Agg')
    m = importlib.import_module('vacmap.' + MODE_MODULE[mode])
    _numba_minmax(m)
    return m


def _numba_minmax(m):
    """numba types `min(50, skipcost)` (mammap_ccs.py:27475, :28587 ...) as float64 when any argument is a float; plain CPython returns the
    Python int / float it picked, which NumPy (NEP 50) then treats as a WEAK scalar: `50 + np.float32(x)` is computed in float32. Round 3's
    wider mode-L goldens showed the difference (one float32 ulp in the strand-switch penalty, local_skipcost = 59 > 50). The reference
    modules therefore get min / max that return np.float64 whenever a float took part, as the compiled code does."""
    import builtins
    import numpy as np

    def wrap(f):
        def g(*a, **kw):
            r = f(*a, **kw)
            if len(a) > 1 and not isinstance(r, np.generic) and isinstance(r, (int, float)) and not isinstance(r, bool) \
                    and any(isinstance(x, (float, np.floating)) for x in a):
                return np.float64(r)
            return r
        return g
    m.min = wrap(builtins.min); m.max = wrap(builtins.max)


def load_output_functions():
    _setup()
    return importlib.import_module('vacmap.output_functions')
