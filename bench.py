#!/usr/bin/env python3
"""bench.py — aligned Gbp/s of the seed -> non-linear chain -> extend path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads-per-step R] [--ref-mb M] [--cpu-sample S]
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (one rank per GPU)

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts the N ranks ITSELF (launch_ranks below: one child of this same command
line per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's JSON line relayed, any rank's non-zero exit code
propagated) — the reference's `-t` forks its own workers too (/root/reference/src/vacmap/vacmap:414-420). Fewer than N devices: a loud error,
never a silent world-1 run. Under torchrun (WORLD_SIZE set) `--gpus` must agree with WORLD_SIZE.

Default workload = the configuration the metric is quoted on: synthetic ONT-shape reads (Gamma lengths, mean 15 kb, 10 % error 4:3:3
sub:del:ins) against the hg38-size synthetic reference (24 contigs with hg38's chromosome proportions, 3.1 Gb, seed 3; SURVEY §8(d)
config 3/4 reference), -mode H -k 15 -w 10 -c 100. `--ref-mb 100` selects BASELINE configs[1] instead (one 100 Mb contig, seed 1).
One "step" = one pass of the whole hot path (vm_align_resident: seed, global chain, local re-seed + chain, extend) over one batch of
reads that is already resident in HBM; the K timed steps process K different batches through the PRODUCT's scheduler
(vacmap_amd.pipeline: length-binned batches inside a bounded window, several batches in flight). value = aligned bases of all ranks /
max-over-ranks time. Multi-GPU: rank 0 builds the index on its GPU and broadcasts it into every other GPU's HBM over RCCL
(vacmap_amd.dist.broadcast_index, timed separately); reads are sharded across ranks; no data-path collective (weak scaling).
"""
import argparse, glob, json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')     # see vacmap_amd/__init__.py: must be set before the HIP runtime starts
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

# single-GPU workloads of BASELINE.json `configs`: reference, read shape, mode, k
CONFIGS = {
    'ont_hg38': dict(ref_mb=0, shape='ont', mean_len=15000, err=0.10, min_len=1000, mode='H', k=15, tag='ont15k_hg38size_H_k15',
                     what='the metric\'s configuration (configs[3] on one GPU): synthetic ONT reads (Gamma mean %d bp, %.1f%% err) vs hg38-size synthetic ref (24 contigs, hg38 proportions, 3.1 Gb, seed 3), -mode H -k 15 -w 10 -c 100'),
    'ont_100mb': dict(ref_mb=100, shape='ont', mean_len=15000, err=0.10, min_len=1000, mode='H', k=15, tag='ont15k_ref100mb_H_k15',
                      what='configs[1]: synthetic ONT reads (Gamma mean %d bp, %.1f%% err) vs 100 Mb synthetic ref (1 contig), -mode H -k 15 -w 10 -c 100'),
    'hifi_hg38': dict(ref_mb=0, shape='hifi', mean_len=18000, err=0.005, min_len=5000, mode='L', k=19, tag='hifi18k_hg38size_L_k19',
                      what='configs[2]: synthetic HiFi reads (Normal mean %d bp, sd 2000, %.1f%% err) vs hg38-size synthetic ref (24 contigs, 3.1 Gb, seed 3), -mode L -k 19 -w 10 -c 100'),
    'vacsim_r': dict(ref_mb=0, shape='hifi', mean_len=18000, err=0.005, min_len=5000, mode='R', k=15, tag='vacsim_hifi18k_hg38size_R_k15', vacsim=True,
                     what='configs[4] on one GPU: donor genome made from the hg38-size synthetic ref by the vacsim-grammar implanter (vacmap_amd/vacsim.py: nested INV, DUP:..:rev:times, '
                          'TRA across contigs, DEL / INS chains, Random{} event sets), HiFi-shape reads (Normal mean %d bp, sd 2000, %.1f%% err) sampled ACROSS the implanted '
                          'events (every read covers at least one), aligned to the unaltered reference, -mode R -k 15 -w 10 -c 100'),
}
# the grammar text of the vacsim_r workload (the line set of tests/test_gpu_kernels.py::test_config5_vacsim_grammar_mode_r, numbers scaled)
VACSIM_TEXT = '''Specified{INV:300:600,DUP:300:600:1:2,TRA:400:800:1;number=%(n)d}
Specified{DEL:100:200,INS:100:1000,INV:100:200,DUP:100:200:0:4,TRA:200:400:1;number=%(n)d}
Specified{INV:400:800,NML:100:200,TRA:400:800:0;number=%(n)d}
Specified{INV:300:900;number=%(n)d}
Random{eventset=["DEL:100:200","INS:100:1000","INV:300:600","DUP:300:600","TRA:400:800"];eventcount=[1,5];number=%(n)d}
Random{eventset=["DEL:100:200,INV:300:600","INS:100:1000,NML:100:200","NML:100:200,INV:300:600","DUP:300:600","TRA:400:800"];eventcount=[4,12];number=%(n)d}
'''
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = 'aligned Gbp/s (whole node) + reads/s, 15 kb ONT-shape reads vs hg38-size ref, 1/2/4/8 MI355X'      # BASELINE.json


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:        # a container's CPU quota (cgroup v2 cpu.max) is what a CPU leg can actually use; more threads only get throttled
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            cores = max(1, min(cores, int(round(int(q) / int(per)))))
    except Exception:
        pass
    return cores


def latest_pmc(workload_id):
    """rocprofv3 PMC summary (tools/pmc_traffic.py) of this same command and workload, newest first; bench.py cannot collect
    counters on itself. None when no summary matches the workload (a summary of another workload would be a stale constant)."""
    for p in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')), reverse=True):
        try:
            pj = json.load(open(p))
        except Exception:
            continue
        if pj.get('workload_id') == workload_id:
            return pj, os.path.relpath(p, ROOT)
    return None, None


def run_side_configs(args):
    """the other single-GPU BASELINE configs, each in a process of its own (own reference, index and work pools), with nothing else on the GPU"""
    import subprocess
    extra = {'configs': []}
    for name in [c for c in args.extra_configs.split(',') if c and c != args.config]:
        cmd = [sys.executable, os.path.abspath(__file__), '--config', name, '--steps', str(args.extra_steps), '--cpu-sample', '0', '--verify', '16', '--extra-configs', '',
               '--streams', str(args.streams), '--reads-per-step', str(args.reads_per_step), '--no-host-input']
        try:
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            d = json.loads(pr.stdout.decode().strip().splitlines()[-1])
            extra['configs'].append({kk: d[kk] for kk in ('value', 'unit', 'ms_per_step', 'steps', 'reads_per_s', 'failed_reads', 'unmapped_reads', 'oracle_crosscheck', 'per_read', 'stage_ms_per_step',
                                                          'hbm_used_gb', 'local_general_reads')} | {'config': name, 'workload': d['config']['workload'], 'dominant_kernel': d['roofline']['kernel'],
                                                          'kernel_ms_per_step': {kn: e['ms_per_step'] for kn, e in d['roofline']['kernels'].items()},
                                                          # the side config's own counters (profiles/r*_pmc_hbm_traffic.json with its workload_id), None until they exist
                                                          'traffic_over_algorithmic': {kn: e['traffic_over_algorithmic'] for kn, e in d['roofline']['kernels'].items()},
                                                          'valu_frac_of_calibrated_peak': {kn: (e['valu'] or {}).get('frac_of_calibrated_peak') for kn, e in d['roofline']['kernels'].items()},
                                                          'ms_per_step_over_valu_floor': (d['roofline']['pipeline_valu'] or {}).get('ms_per_step_over_floor'),
                                                          'traffic_source': d['roofline']['traffic_source'], 'vacsim': d.get('vacsim')})
        except Exception as e:                                                     # a failed side run is reported, never hidden
            extra['configs'].append({'config': name, 'error': repr(e)[:300]})
    return extra


def visible_devices():
    """number of GPUs a rank of this command would see (vm_device_count through the HIP library), asked in a CHILD process: the launcher itself never
    starts the HIP runtime (a parent holding queues on the device is time-sliced against its children). 0 when the library or the device is missing."""
    import subprocess
    code = 'import sys; sys.path.insert(0, %r)\nfrom vacmap_amd.lib import load\nprint("VMX_DEVICES", load().L.vm_device_count())' % ROOT
    try:
        pr = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        for l in pr.stdout.decode().splitlines():
            if l.startswith('VMX_DEVICES'):
                return max(0, int(l.split()[1]))
    except Exception:
        pass
    return 0


def rank_environments(n, port=None):
    """the N environments of the ranks this launcher starts (what torch.distributed.run would set, for one node)"""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
    return [{'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n), 'LOCAL_WORLD_SIZE': str(n), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
             'VMX_BENCH_LAUNCHED': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')} for r in range(n)]


def launch_ranks(n, argv, dry_run=False):
    """start the N ranks of `bench.py --gpus N` (one process per GPU), relay rank 0's JSON line, return the first non-zero exit code of any rank.
    A rank that fails takes the others down with it (their exact PIDs), so a half-started group cannot hang in a collective."""
    import subprocess, threading
    envs = rank_environments(n, port=int(os.environ['VMX_BENCH_PORT']) if os.environ.get('VMX_BENCH_PORT') else None)
    cmd = [sys.executable, os.path.abspath(__file__)] + [a_ for a_ in argv if a_ != '--launch-dry-run']
    if dry_run:
        print(json.dumps({'launcher': 'bench.py', 'n_ranks': n, 'command': cmd, 'environments': envs}))
        return 0
    have = int(os.environ['VMX_BENCH_ASSUME_DEVICES']) if os.environ.get('VMX_BENCH_ASSUME_DEVICES') else visible_devices()      # (the variable: the launcher's own unit test)
    if have < n:
        sys.stderr.write('bench: --gpus %d asked for, but %d GPU(s) are visible to this process (vm_device_count): refusing to run fewer ranks than asked\n' % (n, have))
        return 2
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=dict(os.environ, **e)) for e in envs]
    lines = [[] for _ in procs]

    def pump(r):                        # rank 0's stdout is the result; the other ranks' goes to stderr with a prefix
        for raw in procs[r].stdout:
            l = raw.decode(errors='replace').rstrip('\n')
            lines[r].append(l)
            if r != 0 or not l.startswith('{'):
                sys.stderr.write('[rank %d] %s\n' % (r, l))
    th = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(n)]
    for t_ in th:
        t_.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 128 - c
                sys.stderr.write('bench: rank %d exited with code %d; stopping the other ranks\n' % (r, c))
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    for t_ in th:
        t_.join(timeout=5)
    js = [l for l in lines[0] if l.startswith('{')]
    if rc == 0 and not js:
        sys.stderr.write('bench: rank 0 printed no result line\n'); rc = 1
    if rc == 0:
        print(js[-1])
    return rc


def main():
    from vacmap_amd.driver import _keep_heap_pages
    _keep_heap_pages()                          # the driver's allocator setting (freed result buffers are reused, not unmapped): VMX_DRIVER_MALLOPT=0 disables
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=25, help='timed batches (default 25 x 4096 = the 100k reads of configs[1]/[2])')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads-per-step', type=int, default=4096)
    ap.add_argument('--ref-mb', type=float, default=0.0, help='0 (default): hg38-size 3.1 Gb / 24 contigs; M > 0: one contig of M Mb (100 = BASELINE configs[1])')
    ap.add_argument('--mean-len', type=int, default=0, help='0: the config\'s (15000 ONT, 18000 HiFi)')
    ap.add_argument('--max-len', type=int, default=100000, help='longest read drawn (ONT shape: the Gamma tail is clipped here; 200000 with --mean-len 30000 = the ultra-long robustness run)')
    ap.add_argument('--err', type=float, default=None, help='default: the config\'s (0.10 ONT, 0.005 HiFi)')
    ap.add_argument('--cpu-sample', type=int, default=96, help='minimum reads for the CPU baseline leg (rank 0, N=1 only); 0 disables')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='target wall time of the CPU baseline leg (the sample is sized by a pilot)')
    ap.add_argument('--streams', type=int, default=0, help='batches in flight per GPU (vacmap_amd.pipeline); 0 (default): five, then as many more (up to eight) as the HBM has room '
                    'for after the sizing run — the scheduler\'s own rule (Pipeline.grow_to_memory)')
    ap.add_argument('--window-batches', type=int, default=16, help='length binning window of the scheduler, in batches')
    ap.add_argument('--arrival-order', action='store_true', help='no length binning: batches in arrival order (measured once for comparison)')
    ap.add_argument('--no-host-input', dest='host_input', action='store_false', help='skip the second timed pass, in which the same batches are handed over as HOST buffers (page-locked, as the '
                    'product\'s driver holds them) and uploaded inside vm_align_batch: the PCIe-inclusive rate, reported as `host_input` next to `value`, never as it')
    ap.add_argument('--verify', type=int, default=64, help='reads of the first batch cross-checked against the oracle (0 disables)')
    ap.add_argument('--config', choices=sorted(CONFIGS), default='ont_hg38', help='BASELINE.json workload: ont_hg38 = the metric\'s configuration (default), '
                    'ont_100mb = configs[1], hifi_hg38 = configs[2] (HiFi 18 kb, 0.5 %% error, -mode L -k 19), vacsim_r = configs[4] (vacsim-grammar donor, HiFi-shape reads across the SVs, -mode R)')
    ap.add_argument('--vacsim-svs', type=int, default=1000, help='vacsim_r: complex SVs per grammar line (six lines)')
    ap.add_argument('--extra-configs', default='ont_100mb,hifi_hg38,vacsim_r', help='other single-GPU BASELINE configs timed in their own short runs of this script (N = 1 only) and '
                    'reported under extra.configs next to the headline; "" disables')
    ap.add_argument('--extra-steps', type=int, default=24)
    ap.add_argument('--launch-dry-run', action='store_true', help='with --gpus N: print the N rank environments and the command the launcher would start, and exit')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if 'WORLD_SIZE' in os.environ:
        if int(os.environ['WORLD_SIZE']) != args.gpus:       # under a launcher (torchrun): one rank per GPU asked for, nothing else
            sys.stderr.write('bench: --gpus %d does not agree with WORLD_SIZE=%s of the launcher\n' % (args.gpus, os.environ['WORLD_SIZE']))
            sys.exit(2)
    elif args.gpus > 1 or args.launch_dry_run or os.environ.get('VMX_FORCE_DIST') == '1':
        # no launcher around this process: it becomes the launcher (VMX_FORCE_DIST=1: the N-rank code at world 1 goes through it as well)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.launch_dry_run))
    cfg = dict(CONFIGS[args.config])
    if args.ref_mb > 0:                               # (kept: --ref-mb M = the ONT workload against one contig of M Mb)
        cfg = dict(CONFIGS['ont_100mb']); cfg['ref_mb'] = args.ref_mb
    mean_len = args.mean_len if args.mean_len else cfg['mean_len']
    err = args.err if args.err is not None else cfg['err']

    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and args.extra_configs and os.environ.get('VMX_BENCH_CHILD') != '1' and os.environ.get('VMX_FORCE_DIST') != '1':
        # One GPU, side configs asked for (the default command): this process only orchestrates and never touches the GPU. The headline config runs FIRST in a process of
        # its own (this command line with --extra-configs ""), then every side config in its own; each finds the device as a fresh process does and has it to itself.
        # (Until late in round 5 the side configs ran in front of the headline run inside this process; run after it, with this process still holding its HIP queues,
        # they lost 6 - 23 %: ont_100mb 3.90 / 3.93 against 4.23, vacsim_r 2.40 / 2.34 against 3.03 Gbp/s — two processes' queues on one device are time-sliced.)
        import subprocess
        argv = [a_ for a_ in sys.argv[1:]]
        cmd = [sys.executable, os.path.abspath(__file__)] + argv + ['--extra-configs', '']
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, env=dict(os.environ, VMX_BENCH_CHILD='1'), timeout=3000)
        lines = [l for l in pr.stdout.decode().splitlines() if l.startswith('{')]
        if pr.returncode != 0 or not lines:
            sys.stderr.write('bench: the headline run failed (rc %d)\n' % pr.returncode)
            sys.exit(pr.returncode or 1)
        out = json.loads(lines[-1])
        out['extra'] = run_side_configs(args)
        print(json.dumps(out))
        return
    from vacmap_amd import synth, pipeline
    t0 = time.time()
    cores = host_cores()
    # One process per GPU. VMX_FORCE_DIST=1 runs the N-rank code at world 1 as well (process group over nccl = RCCL, index broadcast through a replica
    # built from the metadata, the all-reduces, the barriers): the one-GPU test of what the driver's 8-GPU run executes (tests/test_gpu_dist.py)
    force_dist = os.environ.get('VMX_FORCE_DIST') == '1'
    dist = None
    if world > 1 or force_dist:
        import torch, torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29531')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank), rank=rank, world_size=world)
    k = cfg['k']
    if cfg['ref_mb'] > 0:
        names = ['chr1']
        contigs = synth.make_reference([int(cfg['ref_mb'] * 1e6)], seed=1)             # configs[1]: 1 contig x 100 Mb, seed 1
        workload_id = cfg['tag'] if cfg['ref_mb'] == 100 else 'ont15k_ref%dmb_H_k15' % int(cfg['ref_mb'])
        workload = (cfg['what'] % (mean_len, err * 100)).replace('100 Mb', '%.0f Mb' % cfg['ref_mb'])
    else:
        names = list(synth.HG38_NAMES)
        if world > 1:
            # rank 0 generates the 3.1 Gb once, with every core, into /dev/shm; the other ranks map it (shared pages): eight ranks regenerating it on cores / 8
            # threads each was the longest part of an 8-rank start
            contigs = synth.shared_reference(synth.hg38_like_lengths(), 3, rank, dist.barrier, threads=cores, tag=os.environ.get('MASTER_PORT', '0'))
        else:
            contigs = synth.make_reference_fast(synth.hg38_like_lengths(), seed=3, threads=cores)
        workload_id = cfg['tag']
        workload = cfg['what'] % (mean_len, err * 100)
    if args.mean_len or args.max_len != 100000 or args.err is not None:
        workload_id += '_mean%d_max%d_err%g' % (mean_len, args.max_len, err)          # (not the BASELINE shape: no PMC summary of another shape may stand in)
    t_ref = time.time() - t0

    nsteps = args.steps
    # the rank's reads: every (step, rank) draws its own; the product's scheduler forms the batches (length binning inside a bounded
    # window, longest reads first). The K timed steps process exactly the drawn reads, each once.
    pool_cat, pool_off = [], [0]
    source, around, vacsim_info = contigs, None, None
    if cfg.get('vacsim'):
        # configs[4]: the reads come from a DONOR genome (the reference with complex SVs implanted by the vacsim grammar) and are aligned to the reference
        from vacmap_amd import vacsim
        tq = time.time()
        donor, pieces, events = vacsim.implant(contigs, VACSIM_TEXT % {'n': args.vacsim_svs}, seed=172)
        around = vacsim.event_positions(pieces, events)
        source, ev_c = donor, around[0]
        types = {}
        for ev in events:
            types[ev['type']] = types.get(ev['type'], 0) + 1
        vacsim_info = {'complex_svs': 6 * args.vacsim_svs, 'events': len(events), 'events_by_type': types, 'events_reads_are_drawn_around': len(ev_c), 'implant_s': time.time() - tq}
    for s in range(nsteps):
        seed = 1000 + 7919 * (s * world + rank)
        cat, off, truth = synth.sample_reads_concat(source, args.reads_per_step, mean_len=mean_len, err=err, seed=seed, shape=cfg['shape'], min_len=cfg['min_len'], max_len=args.max_len, around=around)
        pool_cat.append(cat); pool_off.extend((off[1:] + pool_off[-1]).tolist())
    pool_cat = np.concatenate(pool_cat); pool_off = np.asarray(pool_off, dtype=np.int64)
    lens = np.diff(pool_off)
    plan = pipeline.plan_batches(lens, args.reads_per_step, args.window_batches, sort=not args.arrival_order)
    assert len(plan) >= nsteps and sum(len(p_) for p_ in plan) == nsteps * args.reads_per_step      # (a step = one batch of reads_per_step reads; a batch above the
                                                                                                        # scheduler's bases limit runs as several jobs: pipeline.plan_batches)
    njobs = len(plan)
    t_reads = time.time() - t0 - t_ref

    import torch
    from vacmap_amd.lib import Context, Index, load
    ctx = Context(local_rank)                 # raises without the HIP library / a GPU: no fallback
    lib = load()
    prm = lib.params(cfg['mode'])

    # index: built ON THE GPU by rank 0 only; the other ranks receive it over RCCL into their own HBM
    t1 = time.time()
    index = Index.from_seqs(ctx, names, contigs, k=k, w=10) if rank == 0 else None
    t_index = time.time() - t1
    t_bcast = None
    if dist is not None:
        from vacmap_amd.dist import broadcast_index
        index, t_bcast = broadcast_index(ctx, index, src=0, device=torch.device('cuda', local_rank), self_replica=(world == 1))
    n_minimizers = index.n_minimizers()

    resident = pipeline.upload_batches(ctx, pool_cat, pool_off, plan)        # inputs resident in HBM before timing
    pipe = pipeline.Pipeline(index, prm, device=local_rank, inflight=max(1, min(args.streams or int(os.environ.get('VMX_FULL_CTX', '5')), nsteps)), first_ctx=ctx)
    if (world > 1 or os.environ.get('VMX_BLOCKING_SYNC') == '1') and os.environ.get('VMX_SPIN_SYNC') != '1':
        # N ranks on one host: the contexts' threads wait on a blocking event (the driver's setting; VMX_BLOCKING_SYNC=1 asks for it at N = 1) instead of
        # hipStreamSynchronize. Measured late in round 5 (`host_cores_busy_timed_pass`): the process still keeps one core busy per context either way — the waits that
        # matter are inside the runtime's pageable copies, which spin; hipSetDeviceFlags(hipDeviceScheduleBlockingSync) does put them to sleep (5.7 -> 1.0 cores busy at the
        # same 15.7 ms per batch on the ONT workload) but HUNG the runs with eight contexts in flight (HiFi, vacsim_r, the GPU tests): not used
        for cx in pipe.ctxs:
            cx.set_blocking_sync(True)
    t_setup = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: every context runs the batch with the longest reads (sizes its grow-only pools: no hipMalloc inside the timed region; a
    # long-running mapper reaches that state after its first large batch). The first pass is cross-checked against the oracle.
    longest = int(np.argmax([lens[p].sum() for p in plan]))
    # (with a bases limit per job the job with most bases and the job with most READS are different ones, and each is the largest user of some pools)
    full_jobs = [j for j in range(njobs) if len(plan[j]) == max(len(p_) for p_ in plan)]
    warm_set = [longest] + [j for j in [max(full_jobs, key=lambda j_: resident[j_].bases)] if j != longest]

    def size_pools(cx):
        for j in warm_set:
            resident[j].align(index, prm, want_records=False, ctx=cx)
    verified = None; oi = None; t_oracle_index = None
    warm_runs = 0; warm_oom = 0
    for s in range(args.warmup):
        if s == 0 and args.verify > 0 and rank == 0:
            st, recs, _ = resident[longest].align(index, prm, want_records=True, ctx=ctx); warm_runs += 1
            import oracle_lib as O
            tq = time.time()
            oi = O.Index.from_seqs(names, contigs, k=k, w=10)
            t_oracle_index = time.time() - tq
            op = O.params(cfg['mode'])
            idx = plan[longest]
            nv = min(args.verify, len(idx)); ok = 0
            for j in np.linspace(0, len(idx) - 1, nv).astype(int):
                i = idx[j]
                ost, orecs = O.align_read(oi, pool_cat[pool_off[i]:pool_off[i + 1]].tobytes(), op)
                mine = [t[1:] for t in recs if t[0] == j]
                ok += int((st[j] == 0) == (ost == 0) and mine == [t[1:] for t in orecs])
            verified = '%d/%d' % (ok, nv)
        n_before = pipe.inflight
        warm_oom += pipe.warm(run=size_pools); warm_runs += min(n_before, pipe.inflight + 1)
    ctx_dropped = warm_oom + pipe.trim_to_memory()
    ctx_added = 0; ctx_small = 0
    if args.streams == 0 and not ctx_dropped and args.warmup > 0:
        if os.environ.get('VMX_NO_FULL_GROWTH') != '1':
            ctx_added = pipe.grow_to_memory(run=size_pools, max_inflight=min(int(os.environ.get('VMX_MAX_FULL_CTX', '8')), nsteps)); warm_runs += ctx_added
        if os.environ.get('VMX_SMALL_CTX', '1') != '0' and nsteps >= 8:
            # where no (further) full context fits — its pools are sized by the window's longest batch — contexts for the shorter batches only, sized on the median
            # batch (ONT-hg38: 5 full + 1 small = 283 GB, 15.1-15.7 ms per step against 15.8-15.9; fewer full ones lose: 4 + 3: 15.5-16.1, 4 + 4: 15.4, 3 + 6: 16.1)
            by_bases = sorted(range(njobs), key=lambda j: resident[j].bases)
            med = by_bases[int(len(by_bases) * float(os.environ.get('VMX_SMALL_PCT', '0.5')))]
            ctx_small = pipe.add_small_contexts(resident[med], resident[med].bases, max_inflight=min(int(os.environ.get('VMX_MAX_CTX', '9')), nsteps))
            warm_runs += ctx_small * resident[med].bases / float(max(resident[longest].bases, 1))          # (in units of the longest batch: the PMC summaries scale by warm-up bases)
        if (world > 1 or os.environ.get('VMX_BLOCKING_SYNC') == '1') and os.environ.get('VMX_SPIN_SYNC') != '1':
            for cx in pipe.ctxs:
                cx.set_blocking_sync(True)          # (the product's rule: a context is given up when the sized pools leave < 10 GB of HBM free; not the case at the default sizes)

    agg = {}

    def on_result(i, res):
        stats = res[2]
        if os.environ.get('VMX_DBG_SYNCS'):                   # tuning aid: which batches wait more often than the rest
            sys.stderr.write('[syncs] batch %d reads %d bases %d waits %d records %d cigar bytes %d\n' % (i, stats['n_reads'], stats['read_bases'], stats['n_host_syncs'], stats['n_records'], stats['cigar_bytes']))
        for kk, v in stats.items():
            if kk != 'ms_stage':
                agg[kk] = agg.get(kk, 0) + v
        agg['ms_stage'] = [a + b for a, b in zip(agg.get('ms_stage', [0.0] * 16), stats['ms_stage'])]

    barrier()
    cpu0 = os.times()
    t1 = time.time()
    pipe.run_resident(resident, want_records=False, on_result=on_result)      # the product's schedule (vacmap_amd/pipeline.py)
    barrier()
    dt = time.time() - t1
    cpu1 = os.times()
    host_cores_busy = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(dt, 1e-9)      # CPU seconds of this process per second of the timed pass
    if getattr(pipe, 'timeline', None):                     # VMX_DBG_TIMELINE=1: when every batch of the timed pass started and ended, on which context
        for row in sorted(pipe.timeline, key=lambda r_: r_[3]):
            sys.stderr.write('[timeline] batch %2d ctx %d%s  %7.1f -> %7.1f ms  (%5.1f ms)  %6.1f Mbases\n' % (row[0], row[1], ' small' if row[2] else '      ', row[3] * 1e3, row[4] * 1e3, (row[4] - row[3]) * 1e3, row[5] / 1e6))

    host_rate = None
    if args.host_input and rank == 0 and world == 1:
        # the same batches as HOST buffers: page-locked ones from the driver's pool (vacmap_amd.lib.PinnedPool: the driver gathers every batch's reads into
        # one), so that the upload inside vm_align_batch is a DMA that runs under the other contexts' kernels
        from vacmap_amd.lib import PinnedPool
        pinned = PinnedPool(lib, local_rank)
        blobs = []
        for idx in plan:
            ln = lens[idx]; off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
            cat = pinned.get(int(off[-1]))[:int(off[-1])]
            for j, i in enumerate(idx):
                cat[off[j]:off[j + 1]] = pool_cat[pool_off[i]:pool_off[i + 1]]
            blobs.append((cat, off))
        hagg = {'aligned': 0}

        def on_host(i, st):
            hagg['aligned'] += st['aligned_bases']
        prefetch = os.environ.get('VMX_BENCH_HOST_PREFETCH', '1') != '0'
        if prefetch:
            pipe.upload_slots(max(blobs, key=lambda b: int(b[1][-1])))          # the uploader's context and its reusable device slots, sized like the work pools before anything is timed
        barrier(); th = time.time()
        pipe.run_host_blobs(blobs, on_result=on_host, prefetch=prefetch)
        barrier(); dth = time.time() - th
        host_rate = {'aligned_Gbp_per_s': hagg['aligned'] / dth / 1e9, 'ms_per_step': dth / nsteps * 1e3, 'over_value': (hagg['aligned'] / dth) / (agg['aligned_bases'] / dt),
                     'note': 'second timed pass: same batches, same schedule, reads handed over in page-locked HOST memory (1 B/base); ' + (
                             'an uploader thread with a context of its own streams them into HBM ahead of the aligning contexts (vm_reads_reupload, at most streams + 2 batches ahead), ' if prefetch else
                             'uploaded inside vm_align_batch in front of the batch\'s own kernels, ') + 'results downloaded inside the call in both passes'}
        for cat, off in blobs:
            pinned.release(cat)
        pinned.close()
        if os.environ.get('VMX_BENCH_REPEAT_RESIDENT') == '1':       # tuning aid: the resident pass once more, AFTER the host pass (is a later pass slower whatever it does?)
            barrier(); tr = time.time()
            pipe.run_resident(resident, want_records=False, on_result=None)
            barrier(); host_rate['resident_again_ms_per_step'] = (time.time() - tr) / len(resident) * 1e3

    # (a device tensor only where a collective needs one: after a run that filled the HBM with work pools torch may not find room for its first block)
    vals = torch.tensor([dt, float(agg['aligned_bases']), float(agg['n_reads']), float(agg['read_bases']), float(agg['n_failed'])], dtype=torch.float64, device='cuda' if dist is not None else 'cpu')
    if dist is not None:
        tmax = vals[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = vals[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dt_all = float(tmax[0]); aligned, nreads, rbases, nfail = [float(x) for x in sums]
        mine = torch.tensor([dt, host_cores_busy], dtype=torch.float64, device='cuda')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                        # every rank's own timed pass and host load, for the line
        per_rank_dt = [float(e[0]) for e in every]; per_rank_cores = [float(e[1]) for e in every]
        world_seen = dist.get_world_size()
    else:
        dt_all = dt; aligned, nreads, rbases, nfail = [float(x) for x in vals[1:]]
        per_rank_dt = [dt]; per_rank_cores = [host_cores_busy]; world_seen = 1

    if rank == 0:
        K = args.steps
        # roofline of the DOMINANT kernel of this run: the three heaviest kernels of the path are bracketed by HIP events on the stream
        # they run on (vm_batch_stats: gap-fill fill, local re-seeding, hit clustering); the one with the largest time per step in THIS
        # run is reported. achieved = SURVEY 8(d)'s path-level algorithmic bytes per step / that kernel's time per step:
        #   B(read) = L + 16 M + 8 n + (L + 14000)/4 + 40 R + C   with measured M (minimizers), n (anchors), R (records), C (CIGAR bytes)
        _free, _tot = ctx.mem_info(); hbm_used_gb = (_tot - _free) / 1e9      # index + reads + every context's work pools
        algo_bytes = (agg['read_bases'] + 16 * agg['n_minimizers'] + 8 * agg['n_anchors'] + (agg['read_bases'] + 14000 * agg['n_reads']) / 4.0 +
                      40 * agg['n_records'] + agg['cigar_bytes'])
        # ('k_local_seed' = the local stage's main launch: k_local_seed_band since round 4)
        kms = {'k_gapfill_fill_ns': agg['ms_gapfill_fill'] / K, 'k_local_seed': agg.get('ms_local_seed', 0.0) / K, 'k_cluster_big': agg.get('ms_cluster', 0.0) / K}
        # what each of them must move at the least (its own algorithmic bytes per step): fill = the DP strings + one traceback byte per band
        # cell; local re-seeding = the read (1 B/base) + the 2-bit reference window (SURVEY 8(d)'s (L + 14000)/4); clustering = 8 B per hit in, 32 B per anchor out
        kalgo = {'k_gapfill_fill_ns': (agg['dp_cells'] + agg['dp_string_bytes']) / K,
                 'k_local_seed': (agg['read_bases'] + (agg['read_bases'] + 14000 * agg['n_reads']) / 4.0) / K,
                 'k_cluster_big': (8 * agg['n_hits'] + 32 * agg['n_anchors']) / K}
        dom = max(kms, key=lambda k_: kms[k_])
        dom_ms = kms[dom]
        achieved = (algo_bytes / K) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        pj, src = latest_pmc(workload_id)
        per_kernel = {}
        for kn in kms:
            e = {'ms_per_step': kms[kn], 'kernel_algorithmic_bytes_per_step': kalgo[kn], 'traffic': None, 'traffic_over_algorithmic': None, 'valu': None}
            pks = (pj or {}).get('kernels') or {}
            pk = pks.get('k_local_seed_band' if kn == 'k_local_seed' and 'k_local_seed_band' in pks else (kn if kn != 'k_cluster_big' or 'k_cluster_big' in pks else 'k_cluster'))
            if pk:
                e['traffic'] = pk.get('hbm_bytes_per_step'); e['valu'] = pk.get('valu')
                if kn == 'k_cluster_big':                   # the HIP events bracket every clustering launch: the filtered form, its LONG form, the general path
                    e['traffic'] = sum((pks.get(x) or {}).get('hbm_bytes_per_step') or 0.0 for x in ('k_cluster_big', 'k_cluster_long', 'k_cluster_gen', 'k_cluster')) or e['traffic']
                if e['traffic'] and kalgo[kn] > 0:
                    e['traffic_over_algorithmic'] = e['traffic'] / kalgo[kn]
            e['kernel_GBps'] = (e['traffic'] or kalgo[kn]) / (kms[kn] * 1e-3) / 1e9 if kms[kn] > 0 else 0.0
            per_kernel[kn] = e
        pipeline_valu = None
        if pj is not None and pj.get('valu_wave_insts_per_step'):
            floor_ms = pj['valu_wave_insts_per_step'] / pj['valu_peak_wave_insts_per_s'] * 1e3
            pipeline_valu = {'wave_insts_per_step': pj['valu_wave_insts_per_step'], 'floor_ms_per_step': floor_ms, 'ms_per_step_over_floor': (dt_all * 1e3 / K) / floor_ms,
                             'note': 'all kernels of a step: SQ_INSTS_VALU / calibrated issue peak (profiles/r04_q_valu_calibration.md; tools/pmc_traffic.py holds the constant) against the measured step'}
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                    'traffic': per_kernel[dom]['traffic'], 'traffic_source': src, 'traffic_over_kernel_algorithmic': per_kernel[dom]['traffic_over_algorithmic'],
                    'avg_kernel_ms_per_step': dom_ms, 'algorithmic_bytes_per_step': algo_bytes / K,
                    'kernels': per_kernel, 'valu': per_kernel['k_gapfill_fill_ns']['valu'], 'pipeline_valu': pipeline_valu,
                    'dp_cells_per_s': (agg['dp_cells'] / K) / (kms['k_gapfill_fill_ns'] * 1e-3) if kms['k_gapfill_fill_ns'] > 0 else 0.0,
                    'note': 'kernel = the one with the largest HIP-event time per step in this run (batches share the GPU, so the time includes what the other '
                            'batches\' kernels cost it); achieved = path-level algorithmic bytes of SURVEY 8(d) / that time. The path is not HBM-bound (SURVEY 8(d)): '
                            '`kernels` gives every instrumented kernel its own algorithmic bytes, its PMC traffic (FETCH_SIZE + WRITE_SIZE) and their ratio, '
                            '`valu` the VALU issue fraction of the gap fill, `pipeline_valu` the step against the VALU floor of all its kernels; DESIGN.md §4'}
        cpu = None
        if args.cpu_sample > 0 and world == 1:
            import oracle_lib as O
            if oi is None:
                tq = time.time()
                oi = O.Index.from_seqs(names, contigs, k=k, w=10)
                t_oracle_index = time.time() - tq
            op = O.params(cfg['mode'])
            order = np.argsort(lens, kind='stable')

            def cpu_leg(ns):       # ns reads evenly spaced over the length-sorted timed pool (same length mix as the timed workload)
                pick = order[np.linspace(0, len(order) - 1, ns).astype(np.int64)]
                rds = [pool_cat[pool_off[i]:pool_off[i + 1]].tobytes() for i in pick]
                tc = time.time()
                cst, crecs = O.align_batch(oi, rds, op, nthreads=min(cores, ns))
                tcpu = time.time() - tc
                return sum(t[4] - t[3] for t in crecs), sum(len(r) for r in rds), tcpu
            # a pilot sizes the sample to about --cpu-seconds of wall time on this host
            pilot = min(max(args.cpu_sample, 4 * cores), len(order))
            cal, cb, tcpu = cpu_leg(pilot)
            ns = int(min(len(order), max(pilot, pilot * args.cpu_seconds / max(tcpu, 1e-3))))
            if ns > pilot:
                cal, cb, tcpu = cpu_leg(ns)
            else:
                ns = pilot
            cpu = {'value': cal / tcpu / 1e9, 'unit': 'Gbp/s', 'cores': min(cores, ns), 'kind': 'port',
                   'sample': '%d reads evenly spaced over the length-sorted timed pool (%d bases), oracle/liboracle.so vmo_align_batch with %d std::threads, '
                             'index build excluded (%.0f s)' % (ns, cb, min(cores, ns), t_oracle_index or 0.0),
                   'seconds': tcpu, 'reads_per_s': ns / tcpu}
        out = {
            'metric': METRIC, 'value': aligned / dt_all / 1e9, 'unit': 'Gbp/s',
            'n_gpus': world_seen, 'rccl_ranks': world_seen if dist is not None else 0, 'steps': K, 'warmup': args.warmup, 'ms_per_step': dt_all * 1e3 / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'i16/i32 DP + f64 chains', 'data': 'synthetic',
            'config': {'workload': workload, 'workload_id': workload_id, 'reads_per_step_per_gpu': args.reads_per_step, 'reads_timed': int(nreads),
                       'schedule': 'vacmap_amd.pipeline: %s, %d batches in flight per GPU' % (
                           'arrival-order batches' if args.arrival_order else 'length-binned batches inside windows of %d batches' % args.window_batches, pipe.inflight),
                       'parallelism': 'reads sharded over %d GPU(s), index built by rank 0 and broadcast over RCCL' % world if world > 1 else 'one GPU'},
            'timed_bases': int(rbases), 'warmup_batches': round(warm_runs, 2), 'warmup_bases': int(warm_runs * lens[plan[longest]].sum()),
            'reads_per_s': nreads / dt_all, 'input_Gbp_per_s': rbases / dt_all / 1e9, 'failed_reads': int(nfail), 'unmapped_reads': int(agg['n_unmapped']),
            'device_ms_per_step': agg['ms_total'] / K, 'stage_ms_per_step': [x / K for x in agg['ms_stage'][:8]],
            'stage_names': ['seed', 'global_chain', 'local', 'divergence_filter', 'edge_extension', 'gapfill+records', 'nofilter_redo', 'download'],
            'gapfill_trace_ms_per_step': agg['ms_gapfill_trace'] / K, 'host_syncs_per_step': agg.get('n_host_syncs', 0) / K,
            # per batch, seen from its host thread: wall time of the library call, of it spent inside waits for the stream; the rest is host work with the context's stream empty
            'host_cores_busy_timed_pass': round(host_cores_busy, 2), 'host_cores_busy_all_ranks': round(sum(per_rank_cores), 2), 'host_cores': cores,
            'ms_per_step_per_rank': [round(x * 1e3 / K, 3) for x in per_rank_dt], 'launched_by': 'bench.py' if os.environ.get('VMX_BENCH_LAUNCHED') == '1' else ('launcher' if 'WORLD_SIZE' in os.environ else 'direct'),
            'host_call_ms_per_batch': agg['ms_stage'][15] / K, 'host_wait_ms_per_batch': agg['ms_stage'][14] / K, 'host_active_ms_per_batch': (agg['ms_stage'][15] - agg['ms_stage'][14]) / K,
            'per_read': {'minimizers': agg['n_minimizers'] / max(agg['n_reads'], 1), 'hits': agg['n_hits'] / max(agg['n_reads'], 1), 'anchors': agg['n_anchors'] / max(agg['n_reads'], 1),
                         'local_anchors': agg['n_local_anchors'] / max(agg['n_reads'], 1), 'dp_problems': agg['n_dp_problems'] / max(agg['n_reads'], 1),
                         'dp_cells': agg['dp_cells'] / max(agg['n_reads'], 1), 'records': agg['n_records'] / max(agg['n_reads'], 1)},
            'ed_problems_per_step': agg['n_ed_problems'] / K, 'ed_tier1_per_step': agg.get('n_ed_tier1', 0) / K, 'ed_tier2_per_step': agg.get('n_ed_tier2', 0) / K, 'ed_unbanded_per_step': agg.get('n_ed_full', 0) / K,
            'side_batches_per_step': agg.get('n_ext_retries', 0) / K, 'batch_retries_per_step': agg.get('n_batch_retries', 0) / K,      # reads run again alone (rare parts of the path) / batches run again (an assumed pool size did not hold)
            'dp_redo_per_step': agg.get('n_dp_redo', 0) / K, 'dp_redo_tb_bytes_per_step': agg.get('dp_redo_tb_bytes', 0) / K,
            'oracle_crosscheck': verified, 'setup_s': t_setup, 'reference_gen_s': t_ref, 'read_gen_s': t_reads, 'index_build_s': t_index, 'index_broadcast_s': t_bcast,
            'index_minimizers': int(n_minimizers), 'index_mid_occ': int(index.mid_occ), 'oracle_index_build_s': t_oracle_index, 'hbm_used_gb': hbm_used_gb,
            'local_general_reads': int(agg.get('n_local_general', 0)), 'contexts_given_up_for_memory': int(ctx_dropped), 'contexts_added_for_memory': int(ctx_added), 'small_contexts_added': int(ctx_small), 'longest_read': int(lens.max()),
            'jobs_per_step': njobs / float(K), 'largest_job_bases': int(max(r_.bases for r_ in resident)), 'contexts_in_flight': int(len(pipe.ctxs)),
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        if vacsim_info is not None:
            out['vacsim'] = vacsim_info
        if dist is not None and world == 1:
            out['forced_dist_world_1'] = True
        if host_rate is not None:
            out['host_input'] = host_rate
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
