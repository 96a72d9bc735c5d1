"""TEST-ONLY: open the product's kernels compiled against the CPU fiber emulator (tests/emu). Never used by vacmap_amd."""
import os, sys
_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, 'emu'))
_ctx = None


def context():
    global _ctx
    if _ctx is None:
        import build_emu
        from vacmap_amd.lib import VmxLib, Context
        so = build_emu.build()
        _ctx = Context(0, lib=VmxLib(so))
    return _ctx
