"""Build libvacmapx.so (hand-written HIP for gfx950) in-tree with hipcc. No CUDA paths, no fallbacks.

    python -m vacmap_amd.build            # compiles vacmap_amd/csrc/*.hip -> vacmap_amd/libvacmapx.so
"""
import glob, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libvacmapx.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-Wno-unused-result'] + os.environ.get('VMX_EXTRA_FLAGS', '').split()   # e.g. -DVMX_BAND_W=48 for tuning runs


# per-file flags after the common ones (the last -O wins). VMX_FILE_FLAGS="k_local_band.hip:-Os;k_dp.hip:-O2" overrides / adds for tuning runs.
FILE_FLAGS = {}
for _it in os.environ.get('VMX_FILE_FLAGS', '').split(';'):
    if ':' in _it:
        FILE_FLAGS[_it.split(':', 1)[0].strip()] = _it.split(':', 1)[1].split()


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = glob.glob(os.path.join(CSRC, '*')) + [os.path.join(HERE, '..', 'include', 'vacmapx.h')]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, '_build'), exist_ok=True)
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        obj = os.path.join(HERE, '_build', os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    fail = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            sys.stderr.write('hipcc failed on %s:\n%s\n' % (src, out))
            fail = True
        elif verbose and out.strip():
            print(out)
    if fail:
        raise RuntimeError('libvacmapx build failed')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-lz'])      # zlib: gzip FASTQ input (vmx_sam.hip)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
