"""vacmap_amd — MI355X-native seed -> non-linear chain -> extend for long reads (drop-in for VACmap's hot path).

The compute lives in libvacmapx.so (hand-written HIP for gfx950, vacmap_amd/csrc) behind the C-ABI of
include/vacmapx.h; this package holds the ctypes binding (lib.py), the `vacmap_index`-shaped interface the
reference's Python calls (aligner.py) and the synthetic data generator (synth.py).
"""
__version__ = '0.1'
