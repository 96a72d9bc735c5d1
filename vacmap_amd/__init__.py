"""vacmap_amd — MI355X-native seed -> non-linear chain -> extend for long reads (drop-in for VACmap's hot path).

The compute lives in libvacmapx.so (hand-written HIP for gfx950, vacmap_amd/csrc) behind the C-ABI of
include/vacmapx.h; this package holds the ctypes binding (lib.py), the `vacmap_index`-shaped interface the
reference's Python calls (aligner.py) and the synthetic data generator (synth.py).
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4). A batch in flight drives one main
# and four side streams (the LDS buckets of the chain kernels run side by side), so with the default the batches of vacmap_amd.pipeline
# serialise behind one another's queues (round 1, 4 -> 8 queues: 73.5 -> 65.5 ms per step; round 2, 8 / 16 / 24 queues: 33.9 / 33.4 / 33.5 ms). It has to be in the environment
# before the runtime initialises, i.e. before the library (or torch) first touches the GPU; an explicit setting wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

__version__ = '0.2'
