"""TEST-ONLY: compile the product's kernels + host code against the fiber emulator (hip_emu.h) with g++,
into tests/emu/_build/libvacmapx_emu.so. Used by the `-m "not gpu"` tests to run the SAME kernel sources on the CPU.
The product package never loads this library."""
import glob, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'vacmap_amd', 'csrc')
BUILD = os.path.join(HERE, '_build')
OUT = os.path.join(BUILD, 'libvacmapx_emu.so')
FLAGS = ['-O1', '-g', '-std=c++17', '-fPIC', '-ffp-contract=off', '-DVMX_EMU', '-I', HERE, '-I', CSRC, '-pthread', '-Wno-unused-result',
         '-Wno-attributes'] + os.environ.get('VMX_EMU_EXTRA_FLAGS', '').split()      # e.g. -DVMX_RW_DPP_WINNER: a kernel variant under test


def build(force=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip'))) + [os.path.join(HERE, 'hip_emu.cpp')]
    deps = glob.glob(os.path.join(CSRC, '*')) + glob.glob(os.path.join(HERE, 'hip_emu.*')) + [os.path.join(ROOT, 'include', 'vacmapx.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(BUILD, os.path.basename(s) + '.o')
        objs.append(o)
        procs.append((s, subprocess.Popen(['g++'] + FLAGS + ['-x', 'c++', '-c', s, '-o', o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    bad = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            sys.stderr.write('emu build failed on %s\n%s\n' % (s, out)); bad = True
    if bad:
        raise RuntimeError('emu build failed')
    subprocess.check_call(['g++', '-shared', '-pthread', '-o', OUT] + objs + ['-lz'])
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
