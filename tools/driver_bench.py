#!/usr/bin/env python3
"""End-to-end throughput of the command-line driver (vacmap_amd/driver.py: FASTQ in, SAM out) next to the bench's number for the same
reads — VERDICT r1 item 4: the schedule bench.py times must be the one a user runs.

    python tools/driver_bench.py [--ref-mb 100] [--reads 20480] [--t 16] [--out gpurun_out/driver_bench.json]

Writes a synthetic reference FASTA (BASELINE configs[1] shape: one contig, seed 1) and ONT-shape reads as FASTQ under /tmp, runs
`driver.main` on them (index built on the GPU, `.vmx` cache off), and reports wall time split into index build and the read loop
(parse + align + SAM emission + write), reads/s and aligned Gbp/s of the read loop. Then the same reads go through
`Pipeline.run_resident` (what bench.py times) for the comparison figure."""
import argparse, json, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref-mb', type=float, default=100.0); ap.add_argument('--reads', type=int, default=20480)
    ap.add_argument('--t', type=int, default=16); ap.add_argument('--out', default=None); ap.add_argument('--tmp', default='/tmp/vmx_driver_bench')
    ap.add_argument('--replicate', type=int, default=1, help='write the generated reads this many times under different names (a long input from a short '
                    'generation: --reads 163840 --replicate 10 = 1.6 M reads, ~48 GB of FASTQ; use --tmp /dev/shm/... for that); the quarter run is skipped')
    ap.add_argument('--inflight', type=int, default=0); ap.add_argument('--sam-dir', default=None, help='directory of the SAM output (default: --tmp)')
    ap.add_argument('--driver-args', default='', help='extra arguments for the driver command line, e.g. "--window-batches 16"')
    args = ap.parse_args()
    from vacmap_amd import synth, driver, pipeline
    os.makedirs(args.tmp, exist_ok=True)
    if args.sam_dir:
        os.makedirs(args.sam_dir, exist_ok=True)
    ref = synth.make_reference([int(args.ref_mb * 1e6)], seed=1)[0]
    fa = os.path.join(args.tmp, 'ref.fa'); fq = os.path.join(args.tmp, 'reads.fq'); sam_path = os.path.join(args.sam_dir or args.tmp, 'out.sam')
    with open(fa, 'wb') as f:
        f.write(b'>chr1\n'); f.write(ref.tobytes()); f.write(b'\n')
    cat, off = [], [0]
    for s in range(0, args.reads, 4096):
        c, o, _ = synth.sample_reads_concat([ref], min(4096, args.reads - s), mean_len=15000, err=0.10, seed=1000 + 7919 * (s // 4096))
        cat.append(c); off.extend((o[1:] + off[-1]).tolist())
    cat = np.concatenate(cat); off = np.asarray(off, dtype=np.int64)
    n = len(off) - 1
    with open(fq, 'wb', buffering=1 << 24) as f:
        for rp in range(args.replicate):
            for i in range(n):
                L = int(off[i + 1] - off[i])
                f.write(b'@r%d_%d\n' % (rp, i)); f.write(cat[off[i]:off[i + 1]].tobytes()); f.write(b'\n+\n'); f.write(b'I' * L); f.write(b'\n')
    fq_bytes = os.path.getsize(fq)
    n_total = n * args.replicate
    # the first quarter of the reads as a file of its own: the driver's start-up (three contexts' first pool allocation, ~50 GB of
    # hipMalloc each) is paid once per run whatever its length, so the steady-state rate is the MARGINAL one between the two runs
    nq = max(1, n // 4)
    fq4 = os.path.join(args.tmp, 'reads_quarter.fq')
    with open(fq4, 'wb') as f:
        for i in range(nq):
            L = int(off[i + 1] - off[i])
            f.write(b'@r%d\n' % i); f.write(cat[off[i]:off[i + 1]].tobytes()); f.write(b'\n+\n'); f.write(b'I' * L); f.write(b'\n')
    # each run in a FRESH process, as the command line is used
    import subprocess
    progress = []

    def run_driver(reads):
        env = dict(os.environ, VMX_DRIVER_TIMING='1', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
        t0_ = time.time()
        pr = subprocess.run([sys.executable, '-m', 'vacmap_amd.driver', '-ref', fa, '-read', reads, '-mode', 'H', '-o', sam_path, '-t', str(args.t), '--nowriteindex', '--force',
                             '--inflight', str(args.inflight)] + args.driver_args.split(), env=env, stderr=subprocess.PIPE, text=True)
        dt_ = time.time() - t0_
        tm_ = {}
        progress.clear()
        for ln in pr.stderr.splitlines():
            if ln.startswith('vacmapx timing (s):'):
                tm_ = {kv.split('=')[0]: float(kv.split('=')[1]) for kv in ln.split(':', 1)[1].split()}
            if ' / sec in the last ' in ln:                      # the reference's progress line, every 100 000 reads (vacmap:498-514): rate over the last stretch, average, count
                f_ = ln.split()
                progress.append({'reads': int(f_[-3]), 'per_s_last_100k': int(f_[0]), 'per_s_avg': int(f_[8])})
        sys.stderr.write(pr.stderr[-2000:])
        return pr.returncode, dt_, tm_
    # (the device scrubs memory a process has freed before it hands it out again, ~33 ms per GB: a run that starts right after another one's
    # 190 GB were freed waits for that, which a command-line run on an idle GPU never does; hence the pauses)
    t_quarter = None
    if args.replicate == 1:
        _, t_quarter, _ = run_driver(fq4)
        time.sleep(12)
    rc, t_driver, tm_full = run_driver(fq)
    progress_full = list(progress)
    time.sleep(12)
    t_index = tm_full.get('setup', 0.0)                  # the driver's own set-up (context, FASTA parse, index build on the GPU)
    # the bench's figure for the same reads comes AFTER the driver runs, in this process
    from vacmap_amd.lib import Context, Index, load
    ctx = Context(0)
    idx = Index.from_fasta(ctx, fa, k=15, w=10)
    prm = load().params('H')
    # the bench's figure for these reads (inputs resident, no SAM): the product's scheduler on length-binned batches
    plan = pipeline.plan_batches(np.diff(off), 4096, 16)
    res = pipeline.upload_batches(ctx, cat, off, plan)
    pipe = pipeline.Pipeline(idx, prm, inflight=args.inflight or 5, first_ctx=ctx)
    pipe.warm(res[0])
    if args.inflight == 0:
        pipe.grow_to_memory(res[0], max_inflight=8)
    agg = {'aligned': 0}

    def on(i, r):
        agg['aligned'] += r[2]['aligned_bases']
    t0 = time.time(); pipe.run_resident(res, on_result=on); t_res = time.time() - t0
    del res
    pipe.close(); idx.close(); ctx.close()
    # the driver, end to end: a quarter of the reads, then all of them
    lines = 0
    with open(sam_path, 'rb') as f_:
        while True:
            blk = f_.read(1 << 26)
            if not blk:
                break
            lines += blk.count(b'\n')
    res_rate = n / t_res
    # steady state from the driver's own progress lines: everything after the first 300 000 reads (start-up: first window, the pools' sizing run) up to the last line
    steady = None
    pf = [p_ for p_ in progress_full if p_['reads'] >= 300000]
    if len(pf) >= 2:
        # time of a progress line = reads / average rate
        t_a, t_b = pf[0]['reads'] / max(pf[0]['per_s_avg'], 1), pf[-1]['reads'] / max(pf[-1]['per_s_avg'], 1)
        steady = (pf[-1]['reads'] - pf[0]['reads']) / max(t_b - t_a, 1e-9)
    out = {'reads': n_total, 'unique_reads': n, 'replicate': args.replicate, 'read_bases': int(off[-1]) * args.replicate, 'fastq_bytes': fq_bytes, 'sam_lines_incl_header': lines, 'driver_rc': rc,
           'host_threads_t': args.t, 'inflight': args.inflight,
           'driver_wall_s': t_driver, 'index_build_s': t_index, 'driver_read_loop_s': t_driver - t_index,
           'driver_reads_per_s': n_total / max(t_driver - t_index, 1e-9), 'driver_input_Gbp_per_s': float(off[-1]) * args.replicate / max(t_driver - t_index, 1e-9) / 1e9,
           'resident_pipeline_s': t_res, 'resident_reads_per_s': res_rate, 'resident_aligned_Gbp_per_s': agg['aligned'] / t_res / 1e9,
           'driver_over_resident': (n_total / max(t_driver - t_index, 1e-9)) / res_rate,
           'driver_quarter_wall_s': t_quarter,
           'driver_marginal_reads_per_s': (n - nq) / max(t_driver - t_quarter, 1e-9) if t_quarter else None,
           'driver_marginal_over_resident': ((n - nq) / max(t_driver - t_quarter, 1e-9)) / res_rate if t_quarter else None,
           # the read loop alone (first window read -> last line written), from the driver's own clock
           'driver_loop_s': tm_full.get('loop'), 'driver_loop_reads_per_s': n_total / max(tm_full.get('loop', 0.0), 1e-9),
           'driver_loop_over_resident': (n_total / max(tm_full.get('loop', 0.0), 1e-9)) / res_rate, 'driver_phase_seconds': tm_full,
           # the reference's own yardstick (vacmap:498-514): reads per second over every stretch of 100 000 reads
           'progress_lines': progress_full, 'steady_state_reads_per_s_after_300k': steady, 'steady_state_over_resident': (steady / res_rate) if steady else None}
    print(json.dumps(out))
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
