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

MODE_MODULE = {'H': 'mammap_clrnano', 'L': 'mammap_ccs', 'S': 'mammap_sensitive', 'R': 'mammap_noprefercloser'}


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
    return importlib.import_module('vacmap.' + MODE_MODULE[mode])


def load_output_functions():
    _setup()
    return importlib.import_module('vacmap.output_functions')
