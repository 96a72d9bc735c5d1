"""CPU tests (no GPU) of the index build / save / load / replication code and of the N > 1 path. The PRODUCT's own sources run here
compiled against the test-only fiber emulator (tests/emu); multi-process cases use torch.distributed with the gloo backend, world 2."""
import json, os, subprocess, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ctx():
    import emu_lib
    return emu_lib.context()


def _ref(seed=5, lens=(30000, 12000, 40)):
    from vacmap_amd import synth
    c = synth.make_reference(list(lens), seed=seed)
    c[1][3000:3040] = ord('N')                      # an ambiguous stretch: no minimizer may touch it
    return ['c%d' % i for i in range(len(lens))], c


def test_index_build_matches_oracle_multi_contig(ctx, oracle):
    """device build (sketch tiles -> radix sort -> run-length -> table) vs the oracle's index: both columns, cap, contig table"""
    from vacmap_amd.lib import Index
    names, c = _ref()
    for k, w in ((15, 10), (19, 10), (9, 4), (5, 20)):
        gi = Index.from_seqs(ctx, names, c, k=k, w=w)
        oi = oracle.Index.from_seqs(names, c, k=k, w=w)
        gh, gp = gi.minimizers(); oh, op = oi.minimizers()
        assert np.array_equal(gh, oh) and np.array_equal(gp, op), (k, w)
        assert gi.mid_occ == oi.mid_occ and gi.n_distinct() == oi.n_distinct() and gi.offsets == oi.offsets
        assert gi.seq(1, 2990, 3050) == oi.seq(1, 2990, 3050)
        gi.close()


def test_index_save_load_roundtrip_and_rejects_corruption(ctx, oracle, tmp_path):
    from vacmap_amd.lib import Index, VmxError
    from vacmap_amd import synth
    names, c = _ref(seed=6)
    gi = Index.from_seqs(ctx, names, c, k=15, w=10)
    p = str(tmp_path / 'ref.fa.w10_k15.vmx')
    gi.save(p)
    li = Index.load(ctx, p)
    assert np.array_equal(li.minimizers()[0], gi.minimizers()[0]) and np.array_equal(li.minimizers()[1], gi.minimizers()[1])
    assert (li.k, li.w, li.names, li.lens, li.mid_occ) == (gi.k, gi.w, gi.names, gi.lens, gi.mid_occ)
    rd = synth.mutate(c[0][2000:9000], 0.08, np.random.default_rng(1)).tobytes()
    assert np.array_equal(ctx.map_batch(li, [rd])[0], ctx.map_batch(gi, [rd])[0])
    raw = bytearray(open(p, 'rb').read())
    hdr = 8 + 48

    def bad(mut, what):
        b = bytearray(raw); mut(b)
        q = str(tmp_path / 'bad.vmx'); open(q, 'wb').write(bytes(b))
        with pytest.raises(VmxError) as e:
            Index.load(ctx, q)
        assert e.value.code == -5, what            # VM_ERR_IO, never a crash

    def put(off, v):
        return lambda b: b.__setitem__(slice(off, off + 8), int(v).to_bytes(8, 'little', signed=True))
    bad(put(8, 99), 'k out of range'); bad(put(16, 0), 'w = 0'); bad(put(24, -3), 'negative contig count'); bad(put(32, 1 << 40), 'huge minimizer count')
    bad(put(40, 1 << 50), 'huge total length'); bad(lambda b: b.__delitem__(slice(len(b) - 16, len(b))), 'truncated')
    bad(lambda b: b.extend(b'\0' * 8), 'trailing bytes'); bad(lambda b: b.__setitem__(slice(0, 8), b'VMXIDX01'), 'old format')
    npos = gi.n_minimizers(); pos0 = len(raw) - 8 * npos
    bad(put(pos0 + 8 * 5, int.from_bytes(raw[pos0 + 8 * 5:pos0 + 8 * 5 + 8], 'little') ^ 1), 'strand bit flipped')
    bad(put(pos0 + 8 * 7, (60000 << 1)), 'position crossing a contig end / in the N run')

    def swap(b):
        b[pos0:pos0 + 8], b[pos0 + 800:pos0 + 808] = b[pos0 + 800:pos0 + 808], b[pos0:pos0 + 8]
    bad(swap, 'order broken')
    assert hdr < pos0
    # a replica made from the metadata has no host copy of the bases: seq() and save() decode them from the device pieces
    from vacmap_amd.dist import index_blobs
    rep = Index.from_meta(ctx, gi.meta())
    for d, s in zip(index_blobs(rep, 'cpu'), index_blobs(gi, 'cpu')):
        d.copy_(s)
    assert rep.seq(1, 2990, 3050) == gi.seq(1, 2990, 3050).replace('N', 'N') and rep.n_minimizers() == gi.n_minimizers()
    assert np.array_equal(ctx.map_batch(rep, [rd])[0], ctx.map_batch(gi, [rd])[0])
    p2 = str(tmp_path / 'rep.vmx'); rep.save(p2)
    assert np.array_equal(Index.load(ctx, p2).minimizers()[1], gi.minimizers()[1])
    with pytest.raises(VmxError):
        Index.from_meta(ctx, gi.meta()[:40])
    with pytest.raises(VmxError):
        Index.from_meta(ctx, b'\xff' * 80)


def test_plan_batches_windows():
    from vacmap_amd.pipeline import plan_batches
    rng = np.random.default_rng(3)
    lens = rng.integers(100, 50000, size=1000)
    plan = plan_batches(lens, batch_reads=64, window_batches=4)
    assert sorted(np.concatenate(plan).tolist()) == list(range(1000))            # every read exactly once
    assert len(plan) == 16 and all(len(b) == 64 for b in plan[:15])
    for w in range(0, 16, 4):                                                      # a window never mixes with the next one
        ids = np.concatenate(plan[w:w + 4])
        assert ids.min() >= w * 64 and ids.max() < min(1000, (w + 4) * 64)
        tot = [lens[b].sum() for b in plan[w:w + 4]]
        assert tot == sorted(tot, reverse=True)                                    # longest reads first inside a window
        for b in plan[w:w + 4]:
            assert (np.diff(lens[b]) >= 0).all()                                   # a batch holds reads of ascending length
    flat = plan_batches(lens, 64, 4, sort=False)
    assert np.array_equal(np.concatenate(flat), np.arange(1000))


def test_pipeline_matches_single_context(ctx, oracle):
    """the scheduler (3 contexts in flight, length-binned windows) returns exactly what one batch through one context returns"""
    from vacmap_amd import synth, pipeline
    from vacmap_amd.lib import Index, align_batch
    contigs = synth.make_reference([60000], seed=21)
    gi = Index.from_seqs(ctx, ['chr1'], contigs, k=15, w=10)
    cat, off, _ = synth.sample_reads_concat(contigs, 10, mean_len=1500, err=0.08, seed=22, min_len=400, max_len=4000)
    reads = [cat[off[i]:off[i + 1]].tobytes() for i in range(10)]
    prm = ctx.lib.params('H')
    st0, rec0, _ = align_batch(ctx, gi, prm, reads)
    plan = pipeline.plan_batches(np.diff(off), batch_reads=3, window_batches=2)
    pipe = pipeline.Pipeline(gi, prm, inflight=3, first_ctx=ctx)
    got = {}

    def on_result(i, res):
        st, recs, _ = res
        for j, r in enumerate(plan[i]):
            got[int(r)] = (int(st[j]), [t[1:] for t in recs if t[0] == j])
    res = pipeline.upload_batches(ctx, cat, off, plan)
    pipe.run_resident(res, want_records=True, on_result=on_result)
    for r in range(10):
        assert got[r] == (int(st0[r]), [t[1:] for t in rec0 if t[0] == r])
    got.clear()
    pipe.run_host([[reads[int(r)] for r in b] for b in plan], on_result=on_result)
    assert all(got[r] == (int(st0[r]), [t[1:] for t in rec0 if t[0] == r]) for r in range(10))
    pipe.close()


_WORKER = r'''
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch, torch.distributed as dist
dist.init_process_group(backend='gloo')
rank, world = dist.get_rank(), dist.get_world_size()
import emu_lib                                   # TEST-ONLY: the product's sources on the CPU fiber emulator
ctx = emu_lib.context()
import vacmap_amd.lib as VL
VL._default = ctx.lib                            # the driver below binds the same library
from vacmap_amd import synth
from vacmap_amd.dist import broadcast_index
from vacmap_amd.lib import Index
contigs = synth.make_reference([40000, 9000], seed=31)
built = {'n': 0}
_orig = Index.from_seqs.__func__
def counting(cls, *a, **k):
    built['n'] += 1
    return _orig(cls, *a, **k)
Index.from_seqs = classmethod(counting)
index = Index.from_seqs(ctx, ['a', 'b'], contigs, k=15, w=10) if rank == 0 else None
index, secs = broadcast_index(ctx, index, src=0, device=torch.device('cpu'), chunk_bytes=1 << 16)
assert built['n'] == (1 if rank == 0 else 0)      # no rank other than 0 builds
rd = synth.mutate(contigs[0][5000:11000], 0.08, np.random.default_rng(7)).tobytes()
rows = ctx.map_batch(index, [rd])[0]
allrows = [None] * world
dist.all_gather_object(allrows, rows.tolist())
assert allrows[0] == allrows[1] and len(allrows[0]) > 50
assert index.seq(0, 100, 160) == contigs[0][100:160].tobytes().decode()
# sharded driver run: every rank aligns its share of the batches, rank 0 gathers the SAM lines
tmp = sys.argv[1]
if rank == 0:
    with open(os.path.join(tmp, 'ref.fa'), 'w') as f:
        for n, c in zip(['a', 'b'], contigs):
            f.write('>%%s\n%%s\n' %% (n, c.tobytes().decode()))
    cat, off, _ = synth.sample_reads_concat(contigs, 7, mean_len=1500, err=0.06, seed=33, min_len=500, max_len=3000)
    with open(os.path.join(tmp, 'reads.fa'), 'w') as f:
        for i in range(7):
            f.write('>r%%d\n%%s\n' %% (i, cat[off[i]:off[i + 1]].tobytes().decode()))
dist.barrier()
from vacmap_amd import driver
rc = driver.main(['-ref', os.path.join(tmp, 'ref.fa'), '-read', os.path.join(tmp, 'reads.fa'), '-mode', 'H', '-o', os.path.join(tmp, 'out.sam'), '-t', '8',     # 2 concurrent emit jobs: the replica's host copy of the bases is decoded once, under call_once
                  '--nowriteindex', '--batch-reads', '2', '--window-batches', '2', '--force'], comm=dist)
assert rc == 0
# -mode asm, sharded: contig c -> rank c mod 2, rank 0 writes the lines of every contig in input order
asm_q = [synth.mutate(contigs[0][a:b], 0.01, np.random.default_rng(50 + a)).tobytes().decode() for a, b in ((1000, 4000), (9000, 15000), (20000, 22500), (25000, 31000), (33000, 36000))]
if rank == 0:
    with open(os.path.join(tmp, 'asm.fa'), 'w') as f:
        for i, q in enumerate(asm_q):
            f.write('>c%%d\n%%s\n' %% (i, q))
dist.barrier()
rc = driver.main(['-ref', os.path.join(tmp, 'ref.fa'), '-read', os.path.join(tmp, 'asm.fa'), '-mode', 'asm', '-workdir', os.path.join(tmp, 'wd%%d' %% rank),
                  '-o', os.path.join(tmp, 'asm.sam'), '--nowriteindex', '--force'], comm=dist)
assert rc == 0
if rank == 0:
    from vacmap_amd import sam as S
    from vacmap_amd.lib import align_batch
    prm = ctx.lib.params('asm'); prm.eqx = 1
    st, recs, _ = align_batch(ctx, index, prm, asm_q)
    want = []
    for x, q in enumerate(asm_q):
        mine = [('c%%d' %% x, index.names[t[1]]) + tuple(t[2:]) for t in recs if t[0] == x]
        assert st[x] == 0 and mine
        want += S.sam_lines(mine, q, None, lambda cn, a, b: index.seq(index.names.index(cn), a, b), rg_id='1', asm=True)
    got_asm = [l.rstrip('\n') for l in open(os.path.join(tmp, 'asm.sam')) if not l.startswith('@')]
    assert got_asm == want, (len(got_asm), len(want))
    body = [l.split('\t')[0] for l in open(os.path.join(tmp, 'out.sam')) if not l.startswith('@')]
    print(json.dumps({'names': body, 'secs': secs, 'asm_lines': len(got_asm)}))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_index_broadcast_and_sharded_driver_gloo(tmp_path):
    """world 2, gloo: rank 0 builds, rank 1 receives metadata + the four pieces and maps identically; then the driver runs sharded
    (batch i -> rank i mod 2) and rank 0 writes every read's lines in input order"""
    script = tmp_path / 'w.py'
    script.write_text(_WORKER % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', '29631', str(script), str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    names = d['names']
    assert sorted(set(names), key=lambda s: int(s[1:])) == ['r%d' % i for i in range(7)]
    assert names == sorted(names, key=lambda s: int(s[1:]))          # input order
    assert d['asm_lines'] >= 5                                        # -mode asm sharded by contig: lines equal the one-rank records' (checked in the worker)


# ---------------------------------------------------------------- minimap2 index files
def _py_write_mmi(path, names, contigs, k, w, hashes, positions, offsets, b=6):
    """an INDEPENDENT writer of minimap2's on-disk index (format v3, index.c mm_idx_dump) from sorted (hash, gpos<<1|strand) columns:
    test-side restatement of the format, used to feed vm_index_load_mmi something the product did not write itself"""
    import struct
    out = [b'MMI\x02', struct.pack('<5I', w, k, b, len(names), 0)]
    for n, c in zip(names, contigs):
        out += [struct.pack('<B', len(n)), n.encode(), struct.pack('<I', len(c))]
    offs = np.asarray(offsets, dtype=np.int64)
    buckets = [[] for _ in range(1 << b)]
    for h, p in zip(hashes.tolist(), positions.tolist()):
        g = p >> 1
        rid = int(np.searchsorted(offs, g, side='right') - 1)
        y = (rid << 32) | ((g - int(offs[rid]) + k - 1) << 1) | (p & 1)
        buckets[h & ((1 << b) - 1)].append((h >> b, y))
    for bk in buckets:
        bk.sort()
        groups = {}
        for key, y in bk:
            groups.setdefault(key, []).append(y)
        p, ent = [], []
        for key in sorted(groups, reverse=True):           # any entry order is legal: the reader inserts them into a hash table
            ys = groups[key]
            if len(ys) == 1:
                ent.append(((key << 1) | 1, ys[0]))
            else:
                ent.append((key << 1, (len(p) << 32) | len(ys))); p += ys
        out.append(struct.pack('<i', len(p))); out.append(np.asarray(p, dtype=np.uint64).tobytes())
        out.append(struct.pack('<I', len(ent))); out.append(np.asarray(ent, dtype=np.uint64).tobytes())
    tot = sum(len(c) for c in contigs)
    S = np.zeros((tot + 7) // 8, np.uint32)
    code = np.full(256, 4, np.uint32); code[[65, 67, 71, 84]] = [0, 1, 2, 3]
    cat = code[np.concatenate([np.asarray(c, dtype=np.uint8) for c in contigs])]
    for i in range(8):
        part = cat[i::8]
        S[:len(part)] |= part << np.uint32(4 * i)
    out.append(S.tobytes())
    open(path, 'wb').write(b''.join(out))


def _py_read_mmi(path):
    """independent reader of the same format -> (w, k, names, lens, sorted (hash, gpos<<1|strand) pairs, bases)"""
    import struct
    d = open(path, 'rb').read()
    assert d[:4] == b'MMI\x02'
    w, k, b, nseq, flag = struct.unpack_from('<5I', d, 4); o = 24
    names, lens = [], []
    for _ in range(nseq):
        l = d[o]; o += 1; names.append(d[o:o + l].decode()); o += l; lens.append(struct.unpack_from('<I', d, o)[0]); o += 4
    offs = np.concatenate([[0], np.cumsum(lens)])
    pairs = []
    for bi in range(1 << b):
        n, = struct.unpack_from('<i', d, o); o += 4
        p = np.frombuffer(d, np.uint64, n, o); o += 8 * n
        size, = struct.unpack_from('<I', d, o); o += 4
        ent = np.frombuffer(d, np.uint64, 2 * size, o).reshape(-1, 2); o += 16 * size
        for key, val in ent.tolist():
            h = ((key >> 1) << b) | bi
            ys = [val] if key & 1 else p[val >> 32:(val >> 32) + (val & 0xffffffff)].tolist()
            for y in ys:
                rid, last, z = y >> 32, (y & 0xffffffff) >> 1, y & 1
                pairs.append((h, ((int(offs[rid]) + last - (k - 1)) << 1) | z))
    tot = int(offs[-1])
    S = np.frombuffer(d, np.uint32, (tot + 7) // 8, o); o += 4 * ((tot + 7) // 8)
    assert o == len(d)
    codes = np.stack([(S >> np.uint32(4 * i)) & 15 for i in range(8)], axis=1).reshape(-1)[:tot]
    bases = np.frombuffer(b'ACGTN', np.uint8)[np.minimum(codes, 4)]
    return w, k, names, lens, sorted(pairs), bases


def test_mmi_reader_writer(ctx, oracle, tmp_path):
    from vacmap_amd.lib import Index, VmxError
    from vacmap_amd import synth
    names, c = _ref(seed=8, lens=(26000, 9000, 700))
    oi = oracle.Index.from_seqs(names, c, k=15, w=10)
    oh, op = oi.minimizers()
    # (1) a file written by the independent test-side writer loads into the same index (hashes re-derived from the sequence agree)
    p = str(tmp_path / 'ref.fa.w10_k15.mmi')
    _py_write_mmi(p, names, c, 15, 10, oh, op, oi.offsets)
    gi = Index.load_mmi(ctx, p)
    gh, gp = gi.minimizers()
    assert np.array_equal(gh, oh) and np.array_equal(gp, op) and gi.mid_occ == oi.mid_occ and (gi.k, gi.w, gi.names, gi.lens) == (15, 10, names, [len(x) for x in c])
    assert gi.seq(1, 2990, 3050) == oi.seq(1, 2990, 3050)                    # the N run survives the 4-bit store
    rd = synth.mutate(c[0][3000:9000], 0.08, np.random.default_rng(2)).tobytes()
    assert np.array_equal(ctx.map_batch(gi, [rd])[0], oi.map(rd, 100, -1))
    # (2) a file written by the product is read back by the independent reader with the same content, and by the product itself
    q = str(tmp_path / 'out.mmi')
    built = Index.from_seqs(ctx, names, c, k=15, w=10)
    built.save_mmi(q)
    w, k, n2, l2, pairs, bases = _py_read_mmi(q)
    assert (w, k, n2, l2) == (10, 15, names, [len(x) for x in c])
    assert pairs == sorted(zip(oh.tolist(), op.tolist()))
    assert bases.tobytes() == np.concatenate(c).tobytes()
    again = Index.load_mmi(ctx, q)
    assert np.array_equal(again.minimizers()[1], op)
    # (3) a subset of minimizers is a legal index (minimap2's own selection differs at sequence ends): it loads and maps with ITS set
    keep = np.ones(len(oh), bool); keep[::7] = False
    _py_write_mmi(p, names, c, 15, 10, oh[keep], op[keep], oi.offsets)
    sub = Index.load_mmi(ctx, p)
    assert sub.n_minimizers() == int(keep.sum())
    # (4) rejected: a hash that is not hash64 of its k-mer, HPC / sequence-less flags, truncation, trailing part
    bad = oh.copy(); bad[5] ^= 3
    _py_write_mmi(p, names, c, 15, 10, bad, op, oi.offsets)
    with pytest.raises(VmxError):
        Index.load_mmi(ctx, p)
    raw = bytearray(open(q, 'rb').read())
    for mut in (lambda x: x.__setitem__(slice(20, 24), (1).to_bytes(4, 'little')), lambda x: x.__setitem__(slice(20, 24), (2).to_bytes(4, 'little')),
                lambda x: x.__delitem__(slice(len(x) - 40, len(x))), lambda x: x.extend(b'MMI\x02' + b'\0' * 20), lambda x: x.__setitem__(slice(8, 12), (40).to_bytes(4, 'little'))):
        y = bytearray(raw); mut(y); open(p, 'wb').write(bytes(y))
        with pytest.raises(VmxError):
            Index.load_mmi(ctx, p)


def test_driver_native_path_matches_python_emitter(ctx, oracle, tmp_path, monkeypatch):
    """the command-line driver on the CPU emulator build: gzip FASTQ + BAM input, a repeated read name, --eqx --MD --copycomments; every
    line equals what vacmap_amd/sam.py (the Python statement of the reference's emitter) makes of the same records, in input order"""
    import gzip
    from vacmap_amd import synth, driver, sam
    import vacmap_amd.lib as VL
    from test_host_logic import _write_bam
    monkeypatch.setattr(VL, '_default', ctx.lib)
    contigs = synth.make_reference([50000, 20000], seed=41)
    names = ['cA', 'cB']
    fa = tmp_path / 'ref.fa'
    fa.write_text(''.join('>%s\n%s\n' % (n, c.tobytes().decode()) for n, c in zip(names, contigs)))
    cat, off, _ = synth.sample_reads_concat(contigs, 6, mean_len=1500, err=0.05, seed=42, min_len=600, max_len=2500)
    reads = [cat[off[i]:off[i + 1]].tobytes().decode() for i in range(6)]
    quals = [''.join(chr(33 + (3 * j) % 40) for j in range(len(r))) for r in reads]
    fq = tmp_path / 'r.fq.gz'
    with gzip.open(fq, 'wt') as f:
        for i in range(4):
            f.write('@q%d XC:Z:c%d\n%s\n+\n%s\n' % (i, i, reads[i].lower() if i == 1 else reads[i], quals[i]))
        f.write('@q0 XC:Z:dup\n%s\n+\n%s\n' % (reads[4], quals[4]))             # a repeated name: dropped (vacmap:457)
    bam = tmp_path / 'more.bam'
    _write_bam(str(bam), [('b4', reads[4], quals[4], 0), ('b5', synth.tostr(synth.revcomp(np.frombuffer(reads[5].encode(), np.uint8))), quals[5][::-1], 16)])
    out = tmp_path / 'o.sam'
    assert driver.main(['-ref', str(fa), '-read', str(fq), str(bam), '-mode', 'H', '-o', str(out), '-t', '2', '--nowriteindex', '--eqx', '--MD', '--copycomments',
                        '--batch-reads', '2', '--window-batches', '2', '--inflight', '2']) == 0
    lines = open(out).read().split('\n')
    hdr = [x for x in lines if x.startswith('@')]; body = [x for x in lines if x and not x.startswith('@')]
    assert hdr[1:3] == ['@SQ\tSN:cA\tLN:50000', '@SQ\tSN:cB\tLN:20000'] and hdr[3] == '@RG\tID:1\tSM:sample'
    from vacmap_amd.lib import Index, align_batch
    gi = Index.from_seqs(ctx, names, contigs, k=15, w=10)
    prm = ctx.lib.params('H'); prm.eqx = 1
    order = [('q0', reads[0], quals[0], 'XC:Z:c0'), ('q1', reads[1], quals[1], 'XC:Z:c1'), ('q2', reads[2], quals[2], 'XC:Z:c2'), ('q3', reads[3], quals[3], 'XC:Z:c3'),
             ('b4', reads[4], quals[4], None), ('b5', reads[5], quals[5], None)]
    status, recs, _ = align_batch(ctx, gi, prm, [r[1] for r in order])
    expect = []
    cs = {n: c.tobytes().decode() for n, c in zip(names, contigs)}
    for i, (nm, sq, ql, com) in enumerate(order):
        rr = [(nm, names[t[1]], t[2], t[3], t[4], t[5], t[6], t[7], t[8]) for t in recs if t[0] == i]
        if status[i] == 0 and rr:
            expect += sam.sam_lines(rr, sq, ql, lambda n, a, b: cs[n][a:b], md=True, shortcs=True, markunbalancetra=True, rg_id='1', comments=com)
    assert body == expect and len(body) >= 6


# ---------------------------------------------------------------- N-rank driver on the host: byte-range sharding, per-rank parts
_RANGE_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import torch.distributed as dist
dist.init_process_group(backend='gloo')
import emu_lib                                   # TEST-ONLY: the product's sources on the CPU fiber emulator
ctx = emu_lib.context()
import vacmap_amd.lib as VL
VL._default = ctx.lib
from vacmap_amd import driver
tmp, mode, reads = sys.argv[1], sys.argv[2], sys.argv[3]
extra = ['--parts'] if mode == 'parts' else (['--shard', 'batch'] if mode == 'batch' else [])
rc = driver.main(['-ref', os.path.join(tmp, 'ref.fa'), '-read', os.path.join(tmp, reads), '-mode', 'H', '-o', os.path.join(tmp, 'out_%%s_%%d.sam' %% (mode, dist.get_world_size())),
                  '-t', '4', '--nowriteindex', '--batch-reads', '2', '--window-batches', '2', '--force', '--parse-threads', '2'] + extra, comm=dist)
assert rc == 0
dist.barrier(); dist.destroy_process_group()
'''


def test_fastx_byte_ranges_partition_the_records(ctx, tmp_path):
    """vm_fastx_open_range: whatever the cut points, consecutive ranges yield every record exactly once, in order — FASTQ whose quality lines
    begin with '@', '+' and '>', multi-line FASTA, more ranges than records"""
    import random
    from vacmap_amd.lib import Fastx
    rng = random.Random(5)
    for fq, lead in ((True, ''), (False, ''), (True, '\n\n'), (False, ' \n')):      # (a file that begins with blank lines: the format is its first non-blank byte — ADVICE r4)
        R = []
        for i in range(200):
            L = rng.randint(1, 300)
            R.append(('r%d' % i, ''.join(rng.choice('ACGTNacgt') for _ in range(L)), ''.join(rng.choice('@+!IJ5>') for _ in range(L))))
        path = str(tmp_path / ('t.fq' if fq else 't.fa'))
        with open(path, 'w') as f:
            f.write(lead)
            for nm, s_, q in R:
                if fq:
                    f.write('@%s c%s\n%s\n+\n%s\n' % (nm, nm, s_, q))
                else:
                    f.write('>%s\n' % nm + ''.join(s_[a:a + 60] + '\n' for a in range(0, len(s_), 60)))
        size = os.path.getsize(path)
        want = [(nm, s_.upper(), q if fq else '') for nm, s_, q in R]
        for N in (1, 2, 3, 7, 64, 201):
            cuts = [size * r // N for r in range(N + 1)] if N != 7 else sorted(set([0, size] + [rng.randint(0, size) for _ in range(6)]))
            got = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                rd = Fastx(path, lib=ctx.lib, byte_range=(a, b))
                for ch in iter(lambda: rd.read(17), None):
                    nb, no, sb, so, qb, qo = ch['names'].tobytes(), ch['names_off'], ch['seqs'].tobytes(), ch['seqs_off'], ch['quals'].tobytes(), ch['quals_off']
                    got += [(nb[no[i]:no[i + 1]].decode(), sb[so[i]:so[i + 1]].decode(), qb[qo[i]:qo[i + 1]].decode()) for i in range(len(no) - 1)]
                rd.close()
            assert got == want, (fq, lead, N)


def test_n_rank_driver_range_sharding_parts_gloo(ctx, tmp_path, monkeypatch):
    """world 2 and 4, gloo: every rank parses its own byte range of the FASTQ input (two parser threads over slices) and writes its own part;
    the joined file (and the parts left by --parts, and the --shard batch run) hold the one-rank run's lines — the same multiset, and the
    same bytes once sorted"""
    import vacmap_amd.lib as VL
    from vacmap_amd import synth, driver
    monkeypatch.setattr(VL, '_default', ctx.lib)
    contigs = synth.make_reference([30000, 8000], seed=41)
    with open(tmp_path / 'ref.fa', 'w') as f:
        for n, c in zip(['a', 'b'], contigs):
            f.write('>%s\n%s\n' % (n, c.tobytes().decode()))
    cat, off, _ = synth.sample_reads_concat(contigs, 11, mean_len=1200, err=0.06, seed=43, min_len=400, max_len=2500)
    with open(tmp_path / 'reads.fq', 'w') as f:
        for i in range(11):
            s_ = cat[off[i]:off[i + 1]].tobytes().decode()
            f.write('@r%d\n%s\n+\n%s\n' % (i, s_, '@' * len(s_)))          # (quality lines of '@': the resynchronisation must not take them for headers)
    # the same reads with the last two carrying the names of the first two: the reference keeps the first occurrence of a name, wherever in the
    # input the later ones are (vacmap:457-487) — here in another rank's byte range
    with open(tmp_path / 'reads_dup.fq', 'w') as f:
        for i in range(11):
            s_ = cat[off[i]:off[i + 1]].tobytes().decode()
            f.write('@r%d\n%s\n+\n%s\n' % (i if i < 9 else 10 - i, s_, '@' * len(s_)))
    monkeypatch.setenv('VMX_SLICE_MB', '0.004')                            # ~4 KB slices: several per rank
    wants = {}
    for rf in ('reads.fq', 'reads_dup.fq'):
        one = str(tmp_path / ('one_' + rf + '.sam'))
        assert driver.main(['-ref', str(tmp_path / 'ref.fa'), '-read', str(tmp_path / rf), '-mode', 'H', '-o', one, '-t', '4', '--nowriteindex',
                            '--batch-reads', '2', '--window-batches', '2', '--force', '--parse-threads', '2']) == 0
        wants[rf] = sorted(l for l in open(one) if not l.startswith('@'))
        header = [l for l in open(one) if l.startswith('@') and not l.startswith('@PG')]
    assert len(wants['reads.fq']) >= 11 and len(wants['reads_dup.fq']) < len(wants['reads.fq'])
    script = tmp_path / 'w.py'
    script.write_text(_RANGE_WORKER % (ROOT, ROOT))
    port = 29651
    for world, mode, rf in ((4, 'join', 'reads_dup.fq'), (2, 'parts', 'reads.fq'), (2, 'batch', 'reads_dup.fq')):      # (world 2 'join' ran here too until the suite passed ten minutes)
        port += 1
        want = wants[rf]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VMX_SLICE_MB='0.004')
        out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
                              '--master-port', str(port), str(script), str(tmp_path), mode, rf], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        base = str(tmp_path / ('out_%s_%d.sam' % (mode, world)))
        files = [base + '.part%03d' % r for r in range(world)] if mode == 'parts' else [base]
        assert all(os.path.exists(f) for f in files) and (mode == 'parts' or not os.path.exists(base + '.part000'))
        lines = [l for f in files for l in open(f)]
        assert [l for l in lines if l.startswith('@') and not l.startswith('@PG')] == header, (world, mode)       # the header once
        assert sorted(l for l in lines if not l.startswith('@')) == want, (world, mode, rf)


def test_driver_mode_asm_ignores_c_and_maxdivergence_reads_bam(ctx, tmp_path, monkeypatch):
    """-mode asm through the command line: the fork hard-codes check_num = -1 and maxdivergence = 1.0 (mammap_asm.py:23206, :23483) whatever -c /
    -maxdivergence say (the driver used to pass -c 100 on, keeping only the top 100 clusters), and contigs may come from a .bam"""
    from vacmap_amd import synth, driver
    import vacmap_amd.lib as VL
    import vacmap_amd.driver as D
    from test_host_logic import _write_bam
    monkeypatch.setattr(VL, '_default', ctx.lib)
    contigs = synth.make_reference([30000], seed=77)
    fa = tmp_path / 'ref.fa'
    fa.write_text('>c\n%s\n' % contigs[0].tobytes().decode())
    q = [synth.mutate(contigs[0][a:b], 0.01, np.random.default_rng(a)).tobytes().decode() for a, b in ((2000, 6000), (10000, 13000))]
    bam = tmp_path / 'asm.bam'
    _write_bam(str(bam), [('k0', q[0], 'I' * len(q[0]), 0), ('k1', q[1], 'I' * len(q[1]), 0)])
    seen = []
    real = VL.align_batch_raw

    def spy(cx, index, prm, sb, so):
        seen.append((int(prm.check_num), float(prm.maxdivergence), int(prm.eqx)))
        return real(cx, index, prm, sb, so)
    monkeypatch.setattr(VL, 'align_batch_raw', spy)
    out = tmp_path / 'asm.sam'
    assert driver.main(['-ref', str(fa), '-read', str(bam), '-mode', 'asm', '-workdir', str(tmp_path / 'wd'), '-o', str(out), '-c', '100', '-maxdivergence', '0.2',
                        '--nowriteindex', '--force']) == 0
    assert seen and all(s == (-1, 1.0, 1) for s in seen), seen
    body = [l.split('\t')[0] for l in open(out) if not l.startswith('@')]
    assert sorted(set(body)) == ['k0', 'k1']


def test_host_blobs_prefetch_uploader_matches_inline_upload(ctx, oracle):
    """Pipeline.run_host_blobs(prefetch=True): an uploader thread streams the batches into reusable device slots (vm_reads_reupload) ahead of the
    aligning contexts; per-batch results equal the inline-upload path's (same aligned bases, records and reads per batch), slots are reused"""
    from vacmap_amd import synth, pipeline
    from vacmap_amd.lib import Index
    contigs = synth.make_reference([60000], seed=7)
    gi = Index.from_seqs(ctx, ['c'], [synth.tostr(contigs[0])], k=15, w=10)
    prm = ctx.lib.params('H')
    blobs = []
    for b in range(5):
        cat, off, _ = synth.sample_reads_concat(contigs, 3 + b % 2, mean_len=1500, err=0.08, seed=100 + b, min_len=600, max_len=3000)
        blobs.append((cat, off))
    pipe = pipeline.Pipeline(gi, prm, inflight=2, first_ctx=ctx)
    got = {}
    for pf in (False, True):
        res = {}
        pipe.run_host_blobs(blobs, on_result=lambda i, st: res.__setitem__(i, (st['n_reads'], st['aligned_bases'], st['n_records'])), prefetch=pf)
        got[pf] = res
    assert got[True] == got[False] and len(got[True]) == 5 and all(v[1] > 0 for v in got[True].values())
    assert len(pipe._up[1]) == 4                       # inflight + 2 slots served five batches
    pipe.close()


_SHM_WORKER = r'''
import os, sys, zlib
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
dist.init_process_group(backend='gloo')
from vacmap_amd import synth
lens = [70001, 1234, 50000]
cs = synth.shared_reference(lens, 9, dist.get_rank(), dist.barrier, threads=2, tag=os.environ['MASTER_PORT'], shm_dir=sys.argv[1])
want = synth.make_reference_fast(lens, seed=9, threads=1)
assert [len(c) for c in cs] == lens and all(np.array_equal(a, b) for a, b in zip(cs, want))
cat, off, tr = synth.sample_reads_concat(cs, 5, mean_len=800, err=0.05, seed=3 + dist.get_rank(), min_len=200)      # read-only mapped contigs feed the read sampler
assert len(off) == 6 and off[-1] == len(cat)
dist.barrier(); dist.destroy_process_group()
'''


def test_shared_reference_two_ranks(tmp_path):
    """bench.py at N ranks: rank 0 generates the synthetic reference once into a shared-memory file, the other ranks map it (identical contigs in both,
    usable by the read sampler); the file is gone afterwards"""
    script = tmp_path / 'w.py'
    script.write_text(_SHM_WORKER % ROOT)
    shm = tmp_path / 'shm'; shm.mkdir()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29671')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', '29671',
                          str(script), str(shm)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert os.listdir(shm) == []
