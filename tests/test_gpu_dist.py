"""GPU test (pytest -m gpu) of the N-rank code on the ONE GPU a test box has: bench.py and the driver run with VMX_FORCE_DIST=1, i.e. at world 1
over the nccl backend (= RCCL on ROCm) — process-group start, broadcast_index with the receiving side forced (a replica built from the broadcast
metadata, the collectives on zero-copy views of raw hipMalloc blocks, the replica filled from them), the all-reduces, the barriers, the teardown.
The world-2 / world-4 forms of the same code run on the CPU emulator over gloo (tests/test_index_dist.py); the 8-GPU run is the driver's.
Reference counterpart: the forked workers that share one index copy-on-write, /root/reference/src/vacmap/vacmap:414-420."""
import json, os, subprocess, sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_distributed_path_world_1_nccl():
    """bench.py --gpus 1 with the N-rank code forced: nccl group, index through broadcast_index's replica, all-reduced totals; the records of the
    cross-checked reads still equal the oracle's (they were mapped with the REPLICA of the index)"""
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['VMX_FORCE_DIST'] = '1'                # no launcher variables: bench.py's OWN launcher (launch_ranks) starts the rank, as `bench.py --gpus N` does for N > 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--ref-mb', '20', '--steps', '3', '--reads-per-step', '256', '--streams', '2',
                          '--cpu-sample', '0', '--verify', '8', '--extra-configs', '', '--no-host-input'], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d.get('forced_dist_world_1') is True and d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['launched_by'] == 'bench.py'
    assert len(d['ms_per_step_per_rank']) == 1 and d['host_cores_busy_all_ranks'] >= 0
    assert d['index_broadcast_s'] is not None and d['index_broadcast_s'] > 0
    assert d['oracle_crosscheck'] == '8/8' and d['failed_reads'] == 0 and d['value'] > 0
    assert d['config']['reads_timed'] == 3 * 256                                    # the all-reduced read count


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus N` on a box with fewer than N GPUs exits non-zero with a message — never a silent world-1 run that prints n_gpus 1"""
    sys.path.insert(0, ROOT)
    from vacmap_amd.lib import load
    have = load().L.vm_device_count()
    assert have >= 1
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 1), '--steps', '2'], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and 'refusing to run fewer ranks than asked' in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_driver_distributed_start_world_1_nccl(tmp_path):
    """the driver's N-rank start-up at world 1: nccl + gloo groups, index through the replica; the SAM body equals the plain one-process run's"""
    sys.path.insert(0, ROOT)
    from vacmap_amd import synth
    contigs = synth.make_reference([400000, 150000], seed=91)
    with open(tmp_path / 'ref.fa', 'w') as f:
        for n, c in zip(['a', 'b'], contigs):
            f.write('>%s\n%s\n' % (n, c.tobytes().decode()))
    cat, off, _ = synth.sample_reads_concat(contigs, 24, mean_len=3000, err=0.08, seed=93, min_len=800, max_len=8000)
    with open(tmp_path / 'reads.fq', 'w') as f:
        for i in range(24):
            s_ = cat[off[i]:off[i + 1]].tobytes().decode()
            f.write('@r%d\n%s\n+\n%s\n' % (i, s_, 'I' * len(s_)))
    outs = {}
    for tag, extra_env in (('plain', {}), ('dist', {'VMX_FORCE_DIST': '1', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29543'})):
        o = str(tmp_path / (tag + '.sam'))
        env = dict(os.environ, **extra_env)
        pr = subprocess.run([sys.executable, '-m', 'vacmap_amd.driver', '-ref', str(tmp_path / 'ref.fa'), '-read', str(tmp_path / 'reads.fq'), '-mode', 'H', '-o', o, '-t', '4',
                             '--nowriteindex', '--force', '--batch-reads', '8'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert pr.returncode == 0, pr.stderr[-3000:]
        if tag == 'dist':
            assert 'VMX_FORCE_DIST: process group nccl' in pr.stderr, pr.stderr[-2000:]
        outs[tag] = [l for l in open(o) if not l.startswith('@')]
    assert len(outs['plain']) >= 24 and outs['dist'] == outs['plain']
