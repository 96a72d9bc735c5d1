#!/usr/bin/env python3
"""bench.py — aligned Gbp/s of the seed -> non-linear chain -> extend path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads-per-step R] [--ref-mb M] [--cpu-sample S]
    torchrun / python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...   (one rank per GPU)

Workload (BASELINE.json configs[1], scaled by --steps): synthetic ONT-shape reads (Gamma lengths, mean 15 kb, 10 % error
4:3:3 sub:del:ins) against a synthetic 100 Mb reference, -mode H -k 15 -w 10 -c 100. One "step" = one pass of the whole
hot path (vm_align_resident: seed, global chain, local re-seed + chain, extend) over one batch of reads that is already
resident in HBM; K timed steps use K different batches. value = aligned bases of all ranks / max-over-ranks time.
Multi-GPU: the reference index is replicated per GPU (each rank builds the same seeded reference), reads are sharded
across ranks, no data-path collective (weak scaling); torch.distributed (RCCL) provides the barrier and the reductions.
"""
import argparse, json, os, sys, time
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4). Each batch in flight drives one
# main stream and four side streams (the LDS buckets of the chain kernels run side by side), so with the default the batches in
# flight serialise behind one another's queues; 8 queues measured best (3 batches in flight: 73.5 -> 65.5 ms per step). Must be set
# before the runtime initialises, i.e. before torch / the library touch the GPU. See INTEGRATION.md.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12, help='timed batches (default 12: half of the 24 batches of the 100k-read workload; enough for the 3 batches in flight to reach steady state)')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads-per-step', type=int, default=4096)
    ap.add_argument('--ref-mb', type=float, default=100.0)
    ap.add_argument('--mean-len', type=int, default=15000)
    ap.add_argument('--err', type=float, default=0.10)
    ap.add_argument('--cpu-sample', type=int, default=96, help='minimum reads for the CPU baseline leg (rank 0, N=1 only); 0 disables')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='target wall time of the CPU baseline leg (the sample is sized by a pilot)')
    ap.add_argument('--streams', type=int, default=3, help='batches in flight per GPU: one context (HIP stream set + work pools) and one host thread each')
    ap.add_argument('--verify', type=int, default=8, help='reads of the first batch cross-checked against the oracle (0 disables)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1')); local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    from vacmap_amd import synth
    from vacmap_amd.lib import Context, Index, ResidentReads, load
    ctx = Context(local_rank)                 # raises without the HIP library / a GPU: no fallback
    lib = load()
    prm = lib.params('H')

    t0 = time.time()
    ref_len = int(args.ref_mb * 1e6)
    contigs = synth.make_reference([ref_len], seed=1)             # config 2: 1 contig x 100 Mb, seed 1
    index = Index.from_seqs(ctx, ['chr1'], [contigs[0].tobytes()], k=15, w=10)
    t_index = time.time() - t0

    nsteps = args.steps
    # draw the rank's reads, then form LENGTH-SORTED batches (the driver batches reads of similar length together so that the
    # one-wavefront-per-read kernels of a batch finish together; the reference does not preserve input order either,
    # mammap_clrnano.py:24147-24150). The K timed steps process exactly the drawn reads, each once. The W warm-up steps re-run
    # batches of the same pool, the longest-read batch first: it sizes every grow-only device pool of the context, so that no
    # hipMalloc happens inside the timed region (a long-running mapper reaches that state after its first large batch).
    pool_cat, pool_off = [], [0]
    for s in range(nsteps):
        seed = 1000 + 7919 * (s * world + rank)                    # every (step, rank) draws its own reads
        cat, off, truth = synth.sample_reads_concat(contigs, args.reads_per_step, mean_len=args.mean_len, err=args.err, seed=seed)
        pool_cat.append(cat); pool_off.extend((off[1:] + pool_off[-1]).tolist())
    pool_cat = np.concatenate(pool_cat); pool_off = np.asarray(pool_off, dtype=np.int64)
    lens = np.diff(pool_off)
    order = np.argsort(lens, kind='stable')
    slices = [order[i * args.reads_per_step:(i + 1) * args.reads_per_step] for i in range(nsteps)]
    warm = [(nsteps - 1 - i) % nsteps for i in range(args.warmup)]  # batch indices re-run as warm-up: longest first
    batches = []
    for b in range(nsteps):
        idx = slices[b]
        ln = lens[idx]
        off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        cat = np.empty(int(off[-1]), np.uint8)
        for j, i in enumerate(idx):
            cat[off[j]:off[j + 1]] = pool_cat[pool_off[i]:pool_off[i + 1]]
        batches.append((cat, off))
    resident = [ResidentReads(ctx, concat=c, offsets=o) for c, o in batches]   # inputs resident in HBM before timing
    t_setup = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # optional cross-check of the first batch against the oracle (checker only; outside the timed region)
    verified = None
    oi = None
    # S contexts per GPU (vm_ctx = HIP streams + work pools; "one per host thread per GPU", include/vacmapx.h): batches of different
    # contexts overlap on the device, so the latency-bound chain / re-seeding kernels of one batch run under the VALU-bound DP of another
    nstream = max(1, min(args.streams, nsteps))
    ctxs = [ctx] + [Context(local_rank) for _ in range(nstream - 1)]
    for cx in ctxs:
        cx.set_inflight(nstream)            # contexts that share the GPU launch their latency-bound kernels narrower (vm_ctx_set_inflight)
    for s in range(args.warmup):
        for ci, cx in enumerate(ctxs):      # every context is warmed on the same batches (sizes its pools)
            st, recs, stats = resident[warm[s]].align(index, prm, want_records=(ci == 0 and s == 0 and args.verify > 0 and rank == 0), ctx=cx)
            if ci == 0 and s == 0 and args.verify > 0 and rank == 0:
                import oracle_lib as O
                oi = O.Index.from_seqs(['chr1'], [contigs[0].tobytes()], k=15, w=10)
                op = O.params('H')
                cat, off = batches[warm[0]]
                ok = 0
                for i in np.linspace(0, args.reads_per_step - 1, min(args.verify, args.reads_per_step)).astype(int):
                    rd = cat[off[i]:off[i + 1]].tobytes()
                    ost, orecs = O.align_read(oi, rd, op)
                    mine = [t[1:] for t in recs if t[0] == i]
                    ok += int((st[i] == 0) == (ost == 0) and mine == [t[1:] for t in orecs])
                verified = '%d/%d' % (ok, min(args.verify, args.reads_per_step))

    import threading
    agg = {}
    lock = threading.Lock()
    nxt = [0]
    errs = []

    def worker(cx):
        try:
            while True:
                with lock:
                    s = nsteps - 1 - nxt[0]; nxt[0] += 1      # longest-read batch first: the streams finish on the short ones
                if s < 0:
                    return
                st, _, stats = resident[s].align(index, prm, want_records=False, ctx=cx)     # ctypes releases the GIL for the call
                with lock:
                    for k, v in stats.items():
                        if k != 'ms_stage':
                            agg[k] = agg.get(k, 0) + v
                    agg['ms_stage'] = [a + b for a, b in zip(agg.get('ms_stage', [0.0] * 16), stats['ms_stage'])]
        except Exception as e:      # noqa: a failed batch must fail the bench, not hang it
            errs.append(e)

    barrier()
    t1 = time.time()
    if nstream == 1:
        worker(ctxs[0])
    else:
        th = [threading.Thread(target=worker, args=(cx,)) for cx in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
    barrier()
    dt = time.time() - t1
    if errs:
        raise errs[0]

    vals = torch.tensor([dt, float(agg['aligned_bases']), float(agg['n_reads']), float(agg['read_bases']), float(agg['n_failed'])], dtype=torch.float64, device='cuda')
    if dist is not None:
        tmax = vals[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = vals[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dt_all = float(tmax[0]); aligned, nreads, rbases, nfail = [float(x) for x in sums]
    else:
        dt_all = dt; aligned, nreads, rbases, nfail = [float(x) for x in vals[1:]]

    if rank == 0:
        K = args.steps
        # roofline of the dominant kernel (k_gapfill_fill): algorithmic bytes per launch per SURVEY §8(d):
        #   B(read) = L + 16 M + 8 n + (L + 14000)/4 + 40 R + C   with measured M (minimizers), n (anchors), R (records), C (CIGAR bytes)
        nl = max(int(agg['n_gapfill_launches']), 1)
        _free, _tot = torch.cuda.mem_get_info(local_rank); hbm_used_gb = (_tot - _free) / 1e9      # index + reads + every context's work pools
        algo_bytes = (agg['read_bases'] + 16 * agg['n_minimizers'] + 8 * agg['n_anchors'] + (agg['read_bases'] + 14000 * agg['n_reads']) / 4.0 +
                      40 * agg['n_records'] + agg['cigar_bytes'])
        # one step launches the kernel twice (normal pass + nofilter redo); the redo launch covers a handful of reads, so the
        # per-launch figures below are per STEP (both launches together)
        fill_ms = agg['ms_gapfill_fill'] / K
        achieved = (algo_bytes / K) / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0
        # HBM traffic of the kernel per launch: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of this same command, summarised
        # in profiles/ (bench.py cannot collect counters on itself); null when the summary is absent
        traffic, traffic_src, valu_util = None, None, None
        try:
            pj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_m_pmc_hbm_traffic.json')))
            traffic = (pj['kernels'].get('k_gapfill_fill_ns') or pj['kernels']['k_gapfill_fill'])['hbm_bytes_per_step']        # per step, like achieved (a step = a few chunk launches)
            traffic_src = 'profiles/r01_m_pmc_hbm_traffic.json'
            valu_util = (pj['kernels'].get('k_gapfill_fill_ns') or pj['kernels']['k_gapfill_fill']).get('valu_utilisation')
        except Exception:
            pass
        # what the kernel itself must move: one traceback byte per DP cell (the reference's k_cigar materialises the same matrix) + its strings
        kbytes = (agg['dp_cells'] + agg['dp_string_bytes']) / K
        roofline = {'bound': 'hbm', 'kernel': 'k_gapfill_fill_ns', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                    'traffic': traffic, 'traffic_source': traffic_src, 'valu_utilisation_pmc': valu_util, 'avg_kernel_ms_per_step': fill_ms, 'algorithmic_bytes_per_step': algo_bytes / K,
                    'kernel_bytes_per_step': kbytes, 'kernel_GBps': kbytes / (fill_ms * 1e-3) / 1e9 if fill_ms > 0 else 0.0,
                    'dp_cells_per_s': (agg['dp_cells'] / K) / (fill_ms * 1e-3) if fill_ms > 0 else 0.0,
                    'note': 'achieved uses the path-level algorithmic bytes of SURVEY 8(d); the kernel is an integer DP bound by VALU issue (packed int16, ~75 VALU ops per '
                            '128-cell step, banded stripes with an optimality proof; 96-98 % VALU utilisation while the main launch runs), its own stream is one traceback byte per cell incl. stripe padding (kernel_bytes_per_step); see DESIGN.md'}
        cpu = None
        if args.cpu_sample > 0 and world == 1:
            import oracle_lib as O
            if oi is None:
                oi = O.Index.from_seqs(['chr1'], [contigs[0].tobytes()], k=15, w=10)
            op = O.params('H')
            cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
            try:        # a container's CPU quota (cgroup v2 cpu.max) is what the leg can actually use; more threads only get throttled
                q, per = open('/sys/fs/cgroup/cpu.max').read().split()
                if q != 'max':
                    cores = max(1, min(cores, int(round(int(q) / int(per)))))
            except Exception:
                pass

            def cpu_leg(ns):       # ns reads evenly spaced over the length-sorted timed pool (same length mix as the timed workload)
                pick = order[np.linspace(0, nsteps * args.reads_per_step - 1, ns).astype(np.int64)]
                rds = [pool_cat[pool_off[i]:pool_off[i + 1]].tobytes() for i in pick]
                tc = time.time()
                cst, crecs = O.align_batch(oi, rds, op, nthreads=min(cores, ns))
                tcpu = time.time() - tc
                return sum(t[4] - t[3] for t in crecs), sum(len(r) for r in rds), tcpu
            # a pilot sizes the sample to about --cpu-seconds of wall time on this host
            pilot = min(max(args.cpu_sample, 4 * cores), nsteps * args.reads_per_step)
            cal, cb, tcpu = cpu_leg(pilot)
            ns = int(min(nsteps * args.reads_per_step, max(pilot, pilot * args.cpu_seconds / max(tcpu, 1e-3))))
            if ns > pilot:
                cal, cb, tcpu = cpu_leg(ns)
            else:
                ns = pilot
            cpu = {'value': cal / tcpu / 1e9, 'unit': 'Gbp/s', 'cores': min(cores, ns), 'kind': 'port',
                   'sample': '%d reads evenly spaced over the length-sorted timed pool (%d bases), oracle/liboracle.so vmo_align_batch with %d std::threads, '
                             'index build excluded' % (ns, cb, min(cores, ns)),
                   'seconds': tcpu, 'reads_per_s': ns / tcpu}
        out = {
            'metric': 'aligned Gbp/s (whole node) + reads/s, 15 kb ONT-shape reads vs synthetic ref', 'value': aligned / dt_all / 1e9, 'unit': 'Gbp/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': dt_all * 1e3 / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'int32+f64', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: synthetic ONT reads (Gamma mean %d bp, %.0f%% err) vs %.0f Mb synthetic ref, -mode H -k 15 -w 10 -c 100' % (
                args.mean_len, args.err * 100, args.ref_mb), 'reads_per_step_per_gpu': args.reads_per_step, 'reads_timed': int(nreads),
                'parallelism': 'reads sharded over %d GPU(s), index replicated, %d batches in flight per GPU' % (world, nstream)},
            'reads_per_s': nreads / dt_all, 'input_Gbp_per_s': rbases / dt_all / 1e9, 'failed_reads': int(nfail), 'unmapped_reads': int(agg['n_unmapped']),
            'device_ms_per_step': agg['ms_total'] / K, 'stage_ms_per_step': [x / K for x in agg['ms_stage'][:8]],
            'stage_names': ['seed', 'global_chain', 'local', 'divergence_filter', 'edge_extension', 'gapfill+records', 'nofilter_redo', 'download'],
            'gapfill_trace_ms_per_step': agg['ms_gapfill_trace'] / K,
            'per_read': {'minimizers': agg['n_minimizers'] / max(agg['n_reads'], 1), 'anchors': agg['n_anchors'] / max(agg['n_reads'], 1),
                         'local_anchors': agg['n_local_anchors'] / max(agg['n_reads'], 1), 'dp_problems': agg['n_dp_problems'] / max(agg['n_reads'], 1),
                         'dp_cells': agg['dp_cells'] / max(agg['n_reads'], 1), 'records': agg['n_records'] / max(agg['n_reads'], 1)},
            'ed_problems_per_step': agg['n_ed_problems'] / K, 'ed_tier1_per_step': agg.get('n_ed_tier1', 0) / K, 'ed_tier2_per_step': agg.get('n_ed_tier2', 0) / K, 'ed_unbanded_per_step': agg.get('n_ed_full', 0) / K,
            'oracle_crosscheck': verified, 'setup_s': t_setup, 'index_build_s': t_index, 'hbm_used_gb': hbm_used_gb,
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
