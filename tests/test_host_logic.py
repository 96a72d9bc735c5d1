"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/vacmapx.h declares (no compute without a
GPU), the product fails loudly without a device, the per-read serial pieces (vmx_select.h / vmx_local.h / vmx_extend.h) compile
for the host, and the multi-rank read sharding used by bench.py agrees across ranks (gloo, world_size 2)."""
import ctypes, os, re, subprocess, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vacmap_amd import build
    so = build.build()
    hdr = open(os.path.join(ROOT, 'include', 'vacmapx.h')).read()
    declared = sorted(set(re.findall(r'\b(vm_[a-z_0-9]+)\s*\(', hdr)))
    L = ctypes.CDLL(so)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared) >= 35


def test_no_cpu_fallback_without_gpu():
    """in the GPU-less container vm_ctx_create must fail with VM_ERR_NO_DEVICE; compute entries refuse a NULL context"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from vacmap_amd.lib import Context, VmxError, load
    with pytest.raises(VmxError) as e:
        Context(0)
    assert e.value.code == -2
    L = load().L
    d = ctypes.POINTER(ctypes.c_int64)()
    off = (ctypes.c_int64 * 2)(0, 1)
    assert L.vm_edit_distance_batch(None, 1, b'A', off, b'A', off, ctypes.byref(d)) == -3   # VM_ERR_NO_CTX


def test_product_never_touches_oracle():
    """no file of the product package or its C sources refers to oracle/ (the oracle is test infrastructure)"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'vacmap_amd')):
        if '_build' in base:
            continue
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(base, f), errors='replace').read()
                # comments may CITE oracle files as the spec twin; what must never appear is an include / load / call
                if 'liboracle' in txt or 'oracle_lib' in txt or re.search(r'#include\s*"[^"]*vmo', txt) or re.search(r'\bvmo_[a-z_]+\s*\(', txt):
                    bad.append(f)
    assert not bad, bad


def test_params_defaults_match_reference_modes():
    from vacmap_amd.lib import load
    L = load()
    h, l, s = L.params('H'), L.params('L'), L.params('S')
    assert (h.local_skipcost, h.global_skipcost, h.maxdivergence, h.nodiscard) == (40.0, 40.0, 0.2, 0)      # vacmap:261-272,286-296
    assert (l.local_skipcost, l.global_skipcost, l.maxdivergence, l.nodiscard) == (59.0, 40.0, 0.1, 0)
    assert (s.local_skipcost, s.global_skipcost, s.maxdivergence, s.nodiscard) == (30.0, 30.0, 0.5, 1)
    assert (h.check_num, h.global_maxdiff, h.local_maxdiff, h.local_kmersize) == (100, 50, 30, 9)


def test_synth_generators_are_seeded_and_consistent():
    from vacmap_amd import synth
    c = synth.make_reference([50000], seed=3)
    a1 = synth.sample_reads_concat(c, 16, mean_len=3000, seed=9)
    a2 = synth.sample_reads_concat(c, 16, mean_len=3000, seed=9)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])
    assert a1[1][0] == 0 and len(a1[1]) == 17 and a1[1][-1] == len(a1[0])
    d = synth.implant_svs(c[0][:1000], [('INV', 100, 50), ('DEL', 300, 20), ('DUP', 500, 30, 2)])
    assert len(d) == 1000 - 20 + 60


_WORKER = r'''
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from vacmap_amd import synth
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
contigs = synth.make_reference([200000], seed=1)        # every rank rebuilds the same seeded reference (index replicated)
h = torch.tensor([int(np.frombuffer(contigs[0].tobytes(), dtype=np.uint8).astype(np.int64).sum())])
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
assert all(int(x) == int(h) for x in hs), 'replicated reference differs between ranks'
seeds = [1000 + 7919 * (s * world + rank) for s in range(3)]   # bench.py's (step, rank) read streams
allseeds = [None] * world
dist.all_gather_object(allseeds, seeds)
flat = [s for ss in allseeds for s in ss]
assert len(set(flat)) == len(flat), 'two ranks would draw the same reads'
cat, off, _ = synth.sample_reads_concat(contigs, 8, mean_len=2000, seed=seeds[0])
v = torch.tensor([float(off[-1]), 8.0], dtype=torch.float64)
dist.all_reduce(v)                                          # the reductions bench.py performs (sum of bases / reads)
t = torch.tensor([0.5 + rank], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({'bases': float(v[0]), 'reads': float(v[1]), 'tmax': float(t[0])}))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_sharding_gloo(tmp_path):
    import json
    script = tmp_path / 'w.py'
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', '29617', str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['reads'] == 16.0 and d['tmax'] == 1.5 and d['bases'] > 16 * 1000


def _write_bam(path, records):
    """minimal BAM writer for the reader test: records = (name, seq, qual or None, flag); BGZF = a series of gzip members"""
    import gzip, struct
    code = {c: i for i, c in enumerate('=ACMGRSVTWYHKDBN')}
    body = b'BAM\x01' + struct.pack('<i', 0) + struct.pack('<i', 0)
    blocks = [body]
    for name, seq, qual, flag in records:
        nm = name.encode() + b'\0'
        packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= code[ch] << (4 if i % 2 == 0 else 0)
        q = bytes([0xff] * len(seq)) if qual is None else bytes(ord(c) - 33 for c in qual)
        rec = struct.pack('<iiBBHHHiiii', -1, -1, len(nm), 0, 4680, 0, flag, len(seq), -1, -1, 0) + nm + bytes(packed) + q
        blocks.append(struct.pack('<i', len(rec)) + rec)
    with open(path, 'wb') as f:
        for i in range(0, len(blocks), 3):                       # several gzip members, like BGZF blocks
            f.write(gzip.compress(b''.join(blocks[i:i + 3])))


def test_bam_reader_and_blob_chunks(tmp_path):
    """driver.read_bam (vacmap:455-471): names, sequences, qualities; a reverse-strand record comes back in the read's own orientation;
    missing qualities (0xff) and empty sequences are handled like the reference's pysam loop"""
    from vacmap_amd import driver
    recs = [('r1', 'ACGTNACGTA', 'IIIIIHHHHH', 0), ('r2', 'AACCGGTTA', 'ABCDEFGHI', 16), ('r3', 'GATTACA', None, 4), ('r4', '', None, 4), ('r5', 'ACGRYK', '!!!!!!', 0)]
    p = str(tmp_path / 'x.bam')
    _write_bam(p, recs)
    got = list(driver.read_bam(p))
    assert got == [('r1', 'ACGTNACGTA', 'IIIIIHHHHH', None), ('r2', 'TAACCGGTT', 'IHGFEDCBA', None), ('r3', 'GATTACA', None, None), ('r5', 'ACGRYK', '!!!!!!', None)]
    chunks = list(driver._bam_chunks(p, 3))
    assert [len(c['seqs_off']) - 1 for c in chunks] == [3, 1]
    c0 = chunks[0]
    assert c0['seqs'].tobytes() == b'ACGTNACGTATAACCGGTTGATTACA' and c0['quals_off'].tolist() == [0, 10, 19, 19] and c0['names'].tobytes() == b'r1r2r3'


def test_pipeline_gives_up_a_context_when_hbm_runs_low():
    """the pools regrow after trim_to_memory has run; between two batches a worker whose context sees less than the low-water
    mark free gives the context up (pools freed, thread ends) while more than `keep` are at work; the first context never goes"""
    from vacmap_amd import pipeline

    class FakeCtx:
        def __init__(self, free_gb): self.free, self.closed, self.inflight = free_gb, False, None
        def mem_info(self): return int(self.free * 1e9), int(309e9)
        def close(self): self.closed = True
        def set_inflight(self, n): self.inflight = n
    p = object.__new__(pipeline.Pipeline)
    p.ctxs = [FakeCtx(2.0) for _ in range(4)]; p.inflight = 4
    assert not p.retire_if_low(p.ctxs[0])                     # the index's context stays whatever the memory says
    last = p.ctxs[3]
    assert p.retire_if_low(last) and last.closed and len(p.ctxs) == 3 and p.inflight == 3 and all(c.inflight == 3 for c in p.ctxs)
    assert p.retire_if_low(p.ctxs[2]) and len(p.ctxs) == 2
    assert not p.retire_if_low(p.ctxs[1]) and len(p.ctxs) == 2     # `keep` contexts remain
    p.ctxs = [FakeCtx(50.0) for _ in range(4)]; p.inflight = 4
    assert not p.retire_if_low(p.ctxs[3]) and len(p.ctxs) == 4     # enough memory: nothing happens


def test_pipeline_warm_gives_up_contexts_without_memory():
    """the sizing run of a context that finds no HBM left (VM_ERR_OOM) drops that context and the ones after it; an OOM below `keep` still raises"""
    import pytest
    from vacmap_amd import pipeline
    from vacmap_amd.lib import VmxError

    class FakeCtx:
        def __init__(self, ok): self.ok, self.closed, self.inflight = ok, False, None
        def close(self): self.closed = True
        def set_inflight(self, n): self.inflight = n

    def run(cx):
        if not cx.ok:
            raise VmxError(-4, 'out of memory')
    p = object.__new__(pipeline.Pipeline)
    p.ctxs = [FakeCtx(True), FakeCtx(True), FakeCtx(True), FakeCtx(False), FakeCtx(True)]; p.inflight = 5
    gone = p.ctxs[3:]
    assert p.warm(run=run, keep=2) == 2 and len(p.ctxs) == 3 and p.inflight == 3 and all(c.closed for c in gone) and all(c.inflight == 3 for c in p.ctxs)
    p.ctxs = [FakeCtx(True), FakeCtx(False), FakeCtx(True)]; p.inflight = 3
    with pytest.raises(VmxError):
        p.warm(run=run, keep=2)


def test_pipeline_grows_batches_in_flight_to_memory(monkeypatch):
    """Pipeline.grow_to_memory: contexts are added while the HBM holds another context's pools (measured on the first one added) plus the head-room;
    a sizing run that finds no memory ends the growth without an error"""
    from vacmap_amd import pipeline
    state = {'free': 150e9, 'per': 35e9, 'made': 0}

    class FakeCtx:
        def __init__(self, device=0, lib=None): self.lib, self.closed, self.inflight = lib, False, None; state['made'] += 1
        def mem_info(self): return int(state['free']), int(309e9)
        def close(self): self.closed = True
        def set_inflight(self, n): self.inflight = n
    monkeypatch.setattr(pipeline, 'Context', FakeCtx)

    def run(cx):
        state['free'] -= state['per']
    p = object.__new__(pipeline.Pipeline)
    p.device = 0; p.ctxs = [FakeCtx() for _ in range(5)]; p.inflight = 5
    # 150 GB free, 35 GB per context, 14 GB head-room: 115 -> 80 -> 45 -> (45 < 35 + 14) stop: three added
    assert p.grow_to_memory(run=run, max_inflight=12) == 3 and p.inflight == 8 and all(c.inflight == 8 for c in p.ctxs)
    state['free'] = 200e9
    assert p.grow_to_memory(run=run, max_inflight=9) == 1 and p.inflight == 9          # the cap
    state['free'] = 50e9
    assert p.grow_to_memory(run=run, max_inflight=12) == 0                              # nothing known about the pools and little room: not tried


def test_pipeline_stream_ramps_contexts_in_the_background(monkeypatch):
    """Pipeline.run_stream(ramp=...): the stream starts on the one sized context; the others are created, sized (run) and put to work by a background
    thread while batches already run — by the memory rule (target 0) or up to a fixed number —, every job runs exactly once, a sizing run without
    memory ends the growth without failing the run, and nothing more is sized once the jobs are exhausted"""
    import threading, time
    from vacmap_amd import pipeline
    from vacmap_amd.lib import VmxError
    state = {'free': 200e9, 'per': 40e9, 'made': 0, 'oom_at': None}

    class FakeCtx:
        def __init__(self, device=0, lib=None): self.lib, self.closed, self.inflight, self.blocking = lib, False, None, False; state['made'] += 1
        def mem_info(self): return int(state['free']), int(309e9)
        def close(self): self.closed = True
        def set_inflight(self, n): self.inflight = n
    monkeypatch.setattr(pipeline, 'Context', FakeCtx)

    def sizing(cx):
        if state['oom_at'] is not None and state['made'] - 1 >= state['oom_at']:
            raise VmxError(-4, 'out of device memory')
        time.sleep(0.01); state['free'] -= state['per']

    def make():
        p = object.__new__(pipeline.Pipeline)
        p.device = 0; p.ctxs = [FakeCtx()]; p.inflight = 1
        return p
    seen = {}; lk = threading.Lock()

    def do_job(job, cx):
        time.sleep(0.004)
        with lk:
            seen.setdefault(job, []).append(cx)
    # memory rule: 200 GB free, 40 GB per context, 14 GB head-room: 160 -> 120 -> 80 -> 40 (40 < 40 + 14: stop) = four added
    p = make(); state['made'] = 1
    p.run_stream(iter(range(60)), do_job, ramp=dict(run=sizing, target=0, max_inflight=8, on_ctx=lambda cx: setattr(cx, 'blocking', True)))
    assert sorted(seen) == list(range(60)) and all(len(v) == 1 for v in seen.values())
    assert p.inflight == 5 and all(c.inflight == 5 for c in p.ctxs) and all(c.blocking for c in p.ctxs[1:])
    assert len(set(id(v[0]) for v in seen.values())) >= 3                                # the late contexts did take batches
    assert seen[0][0] is p.ctxs[0]                                                       # and the first batch did not wait for them
    # fixed target, and a sizing run that finds no memory: growth ends there, the run does not fail
    seen.clear(); state.update(free=300e9, made=1, oom_at=3)
    p = make()
    p.run_stream(iter(range(40)), do_job, ramp=dict(run=sizing, target=6, max_inflight=8))
    assert sorted(seen) == list(range(40)) and p.inflight == 2 and p.ramp_oom == 1
    # a short stream: nothing is sized after the jobs are gone
    seen.clear(); state.update(free=300e9, made=1, oom_at=None)
    p = make()
    p.run_stream(iter(range(2)), do_job, ramp=dict(run=sizing, target=8))
    assert sorted(seen) == [0, 1] and p.inflight <= 2


def test_pipeline_small_contexts_only_take_small_batches():
    """the size-aware schedule of Pipeline._run: a context added by add_small_contexts never runs a batch above its limit, every batch runs exactly once,
    and the entries that do not know batch sizes (run_stream, run_host_blobs) leave the small contexts out"""
    import threading
    from vacmap_amd import pipeline

    class FakeCtx:
        def __init__(self, name): self.name = name
        def mem_info(self): return int(100e9), int(309e9)
        def set_inflight(self, n): pass
        def close(self): pass
    p = object.__new__(pipeline.Pipeline)
    full = [FakeCtx('F%d' % i) for i in range(2)]; small = [FakeCtx('S%d' % i) for i in range(3)]
    p.ctxs = full + small; p.small_ctxs = list(small); p.small_limit = 50; p.inflight = 5; p.device = 0
    bases = [160, 120, 90, 70, 55, 50, 45, 40, 30, 20, 10, 5] * 2            # two windows, longest first inside each
    ran = {}; lock = threading.Lock()

    def do_job(i, cx):
        import time; time.sleep(0.002 * bases[i] / 40.0)
        with lock:
            assert i not in ran
            ran[i] = cx.name
        return i
    p._run(len(bases), do_job, None, job_bases=bases, horizon=12)
    assert sorted(ran) == list(range(len(bases)))
    assert all(bases[i] <= 50 for i, nm in ran.items() if nm.startswith('S')) and any(nm.startswith('S') for nm in ran.values())
    assert [c.name for c in p.full_ctxs()] == ['F0', 'F1']
    # without sizes the small contexts stay out
    ran.clear()
    p._run(6, do_job, None)
    assert sorted(ran) == list(range(6)) and all(nm.startswith('F') for nm in ran.values())


def _bench(args, env_extra=None, drop=('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')):
    env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=600, env=env)


def test_bench_launcher_dry_run_prints_the_rank_environments():
    """`bench.py --gpus N` starts its N ranks itself (launch_ranks; the reference's -t forks its own workers, /root/reference/src/vacmap/vacmap:414-420):
    the dry run shows one environment per rank, a shared rendezvous on 127.0.0.1, and the child command = this command line"""
    import json
    out = _bench(['--gpus', '8', '--steps', '20', '--warmup', '5', '--launch-dry-run'])
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    envs = d['environments']
    assert d['n_ranks'] == 8 and len(envs) == 8
    assert [e['RANK'] for e in envs] == [str(r) for r in range(8)] and [e['LOCAL_RANK'] for e in envs] == [str(r) for r in range(8)]
    assert {e['WORLD_SIZE'] for e in envs} == {'8'} and {e['MASTER_ADDR'] for e in envs} == {'127.0.0.1'} and len({e['MASTER_PORT'] for e in envs}) == 1
    assert d['command'][1].endswith('bench.py') and d['command'][2:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']


def test_bench_launcher_refuses_missing_devices_and_disagreeing_world():
    """fewer devices than --gpus: non-zero exit and a message, no result line (this container has no GPU at all); under a launcher, --gpus must equal WORLD_SIZE"""
    out = _bench(['--gpus', '2', '--steps', '2'])
    assert out.returncode != 0 and 'refusing to run fewer ranks than asked' in out.stderr and '{' not in out.stdout
    out = _bench(['--gpus', '2'], env_extra={'WORLD_SIZE': '4', 'RANK': '0'})
    assert out.returncode != 0 and 'does not agree with WORLD_SIZE' in out.stderr


def test_bench_launcher_propagates_a_failing_rank():
    """a rank that dies takes the launcher's exit code with it and the other ranks are stopped (here every rank fails at vm_ctx_create: no device)"""
    if os.path.exists('/dev/kfd'):
        pytest.skip('needs a box without a GPU: the ranks are meant to fail at context creation')
    out = _bench(['--gpus', '2', '--steps', '2', '--ref-mb', '1', '--reads-per-step', '8', '--extra-configs', ''], env_extra={'VMX_BENCH_ASSUME_DEVICES': '2'})
    assert out.returncode != 0 and 'exited with code' in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_cross_rank_name_dedup_keeps_first_in_input_order(tmp_path):
    """range mode: a name that occurs in two ranks' byte ranges keeps its FIRST occurrence in input order — (file, then byte range = rank) — as the reference's
    single pass over the files does (/root/reference/src/vacmap/vacmap:457-487), not the lowest rank's; and the part filter drops exactly the
    lines of those names, hashing the names block-wise"""
    import io
    from vacmap_amd import driver
    nm = lambda names: driver.name_hashes(np.frombuffer(b''.join(names), dtype=np.uint8), np.concatenate([[0], np.cumsum([len(x) for x in names])]))
    # rank 0 holds 'x' from file 1 (second file) and 'a' from file 0; rank 3 holds 'x' from file 0 (first file): the file-0 occurrence (rank 3) stays
    h0, f0 = nm([b'a', b'x']), np.asarray([0, 1], np.int32)
    h1, f1 = nm([b'b']), np.asarray([0], np.int32)
    h2, f2 = nm([b'c', b'a']), np.asarray([0, 1], np.int32)           # 'a' again in file 1 on rank 2: goes (rank 0 has it from file 0)
    h3, f3 = nm([b'x']), np.asarray([0], np.int32)
    drop = driver.cross_rank_duplicates([h0, h1, h2, h3], [f0, f1, f2, f3])
    assert set(drop) == {0, 2}
    assert list(drop[0]) == list(nm([b'x'])) and list(drop[2]) == list(nm([b'a']))
    # same file on both sides: the lower rank (= lower byte range) stays
    drop = driver.cross_rank_duplicates([nm([b'q']), nm([b'q'])], [np.asarray([0], np.int32), np.asarray([0], np.int32)])
    assert set(drop) == {1}
    # the filter: header lines pass, every line of a dropped name goes, block boundaries inside lines are handled
    lines = [b'@HD\tVN:1.6\n', b'@SQ\tSN:a\tLN:9\n'] + [b'r%d\t0\ta\t%d\t60\t5M\t*\t0\t0\tACGTA\tIIIII\n' % (i % 7, i) for i in range(50)] + [b'x\t4\t*\n']
    src = io.BytesIO(b''.join(lines)); dst = io.BytesIO()
    bad = np.sort(nm([b'r3', b'x']))
    gone, gl = driver._filter_part(src, dst, bad, block=97)
    want = [l for l in lines if not (l.startswith(b'r3\t') or l.startswith(b'x\t'))]
    assert dst.getvalue() == b''.join(want) and gl == len(lines) - len(want) and len(gone) == 2
    src = io.BytesIO(b''.join(lines)[:-1]); dst = io.BytesIO()      # no newline at the end of the part
    driver._filter_part(src, dst, np.sort(nm([b'r1'])), block=1 << 20)
    assert dst.getvalue() == b''.join(l for l in lines if not l.startswith(b'r1\t'))


def test_plan_batches_bases_limit_per_job():
    """pipeline.plan_batches(max_bases=...): a batch above the limit becomes equal-bases jobs of consecutive reads (the longest reads' part first), every read
    exactly once, the other batches untouched; off (0) by default"""
    import numpy as np
    from vacmap_amd import pipeline
    rng = np.random.default_rng(5)
    ln = np.clip(rng.gamma(2.0, 7500, size=4096 * 4 + 100), 1000, 100000).astype(np.int64)
    base = pipeline.plan_batches(ln, 4096, 16, max_bases=0)
    assert [len(b) for b in base] == [4096] * 4 + [100] and pipeline.DEFAULT_BATCH_MAX_BASES == 0
    assert [b.tolist() for b in pipeline.plan_batches(ln, 4096, 16)] == [b.tolist() for b in base]
    lim = 60_000_000
    cut = pipeline.plan_batches(ln, 4096, 16, max_bases=lim)
    assert sorted(np.concatenate(cut).tolist()) == list(range(len(ln))) and len(cut) > len(base)
    assert max(int(ln[b].sum()) for b in cut) <= lim * 1.02 and int(ln[base[0]].sum()) > lim
    first = [b for b in cut if set(b.tolist()) <= set(base[0].tolist())]
    assert len(first) >= 2 and np.array_equal(np.concatenate(first[::-1]), base[0])          # consecutive slices, listed from the long end
    shares = [int(ln[b].sum()) for b in first]
    assert max(shares) - min(shares) <= 2 * int(ln.max())
    small = [b for b in base if int(ln[b].sum()) <= lim]
    assert all(any(np.array_equal(b, c) for c in cut) for b in small)
    one = pipeline.split_by_bases(base[0][:1], ln, 10)
    assert len(one) == 1 and np.array_equal(one[0], base[0][:1])              # a single read is never cut


def test_vacsim_event_positions_point_into_the_donor():
    """vacsim.event_positions: every position it returns is the donor coordinate of an implanted event's left end (mapped back through the
    contig's own forward pieces it is an event's `start`), and `sample_reads_concat(around=...)` takes them"""
    import numpy as np
    from vacmap_amd import synth, vacsim
    contigs = synth.make_reference([400_000, 300_000], seed=5)
    text = 'Specified{INV:300:600,DUP:300:600:1:2,TRA:400:800:1;number=4}\nSpecified{DEL:100:200,INS:100:1000;number=6}\nSpecified{INV:300:900;number=5}\n'
    donor, pieces, events = vacsim.implant(contigs, text, seed=7)
    ev_c, ev_p = vacsim.event_positions(pieces, events)
    assert len(ev_c) == len(ev_p) and 10 <= len(ev_c) <= len(events)
    starts = {}
    for ev in events:
        starts.setdefault(ev['contig'], set()).add(int(ev['start']))
    for c, p in zip(ev_c.tolist(), ev_p.tolist()):
        assert 0 <= p <= len(donor[c])
        back = [ss + (p - ds) for ds, de, sc, ss, se, st in pieces[c] if sc == c and st > 0 and ds <= p <= ds + (se - ss)]      # the reference coordinate(s) a forward piece of the
        assert any(x in starts[c] for x in back), (c, p, back)                                                                # contig itself maps this donor position to: an event's start
    cat, off, truth = synth.sample_reads_concat(donor, 20, seed=3, around=(ev_c, ev_p), mean_len=8000, err=0.0, min_len=4000)
    assert len(off) == 21 and off[-1] == len(cat)
