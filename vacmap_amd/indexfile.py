"""Where the driver gets its index from (src/vacmap/vacmap:324-344: `<ref>.w<w>_k<k>.mmi` is built once with `minimap2 -d` and reused).

Order: `<ref>.w<w>_k<k>.vmx` (own format, loads in seconds) if it is not older than the FASTA; else a minimap2-built
`<ref>.w<w>_k<k>.mmi` (index format v3, read by vm_index_load_mmi); else the index is built on the GPU from the FASTA and, unless
`write` is off, saved as `.vmx` for the next run. A stale or corrupt cached file is never trusted: the loaders validate every field
(VM_ERR_IO) and the driver then rebuilds."""
import os
import sys

from .lib import Index, VmxError


def index_paths(ref, k, w):
    base = '%s.w%d_k%d' % (ref, w, k)
    return base + '.vmx', base + '.mmi'


def find_index(ctx, ref, k, w, write=True, log=sys.stderr):
    vmx, mmi = index_paths(ref, k, w)
    ref_m = os.path.getmtime(ref)
    for path, loader in ((vmx, Index.load), (mmi, Index.load_mmi)):
        if os.path.exists(path) and os.path.getmtime(path) >= ref_m:
            try:
                idx = loader(ctx, path)
                if idx.k == k and idx.w == w:
                    return idx
                log.write('vacmapx: %s holds k=%d w=%d, not k=%d w=%d: ignored\n' % (path, idx.k, idx.w, k, w))
                idx.close()
            except VmxError as e:
                log.write('vacmapx: cached index %s rejected (%s): rebuilding\n' % (path, e))
        elif os.path.exists(path):
            log.write('vacmapx: cached index %s is older than %s: rebuilding\n' % (path, ref))
    idx = Index.from_fasta(ctx, ref, k=k, w=w)
    if write:
        try:
            idx.save(vmx)
        except (VmxError, OSError) as e:
            log.write('vacmapx: could not save %s (%s)\n' % (vmx, e))
    return idx
