"""CPU tests (no GPU): the PRODUCT's HIP kernels, compiled unchanged against the fiber emulator (tests/emu), vs the oracle.
These check kernel logic in the GPU-less container; the `-m gpu` twins in test_gpu_*.py run the real library."""
import numpy as np
import pytest
import kernel_cases as KC


@pytest.fixture(scope='module')
def ctx():
    import emu_lib
    return emu_lib.context()


def test_emu_tables(ctx, oracle):
    KC.check_tables(ctx, oracle)


def test_emu_edit_distance(ctx, oracle):
    KC.check_edit_distance(ctx, oracle, n=12, maxlen=300, seed=1)
    KC.check_edit_distance(ctx, oracle, n=2, maxlen=4500, seed=2, minlen=4200)   # > 64 blocks: multi-pass carry


def test_emu_edit_distance_bound(ctx, oracle):
    KC.check_edit_distance_bound(ctx, oracle, seed=5, lens=(1, 63, 64, 65, 700, 1700), big=True)
    KC.check_edit_distance_bound(ctx, oracle, seed=6, lens=(1, 63, 64, 65, 700, 1300), big=True, tier=1)      # four problems per wave


def test_emu_extend(ctx, oracle):
    KC.check_extend(ctx, oracle, n=16, seed=3)


def test_emu_gapfill(ctx, oracle):
    KC.check_gapfill(ctx, oracle, n=10, maxlen=150, seed=4)
    KC.check_gapfill(ctx, oracle, n=6, maxlen=330, seed=5)       # several 128-row stripes of the packed layout; beyond 420 cells of perimeter: int32 layout


def test_emu_reference_call_shapes(ctx, oracle, golden):
    KC.check_reference_call_shapes(ctx, oracle, golden, small=True)


def test_emu_gapfill_tie_order(ctx, oracle):
    KC.check_gapfill_ties(ctx, oracle)


def test_emu_gapfill_banded(ctx, oracle, monkeypatch):
    """the batched path's gap-fill schedule (anti-diagonal band + proof + redo queue + layout flag) with the emulator build's small
    constants (small class up to tl + ql = 160, packed int16 up to 420). The band-width rule is pushed through all four widths with
    VMX_AD_PCT (at its default every problem this small gets the narrowest band)."""
    seen = set()
    for pct, seed in ((100, 44), (250, 45), (330, 46), (400, 47)):
        monkeypatch.setenv('VMX_AD_PCT', str(pct))
        # (the wide bands hold every path of problems this small: g > min(tl, ql), nothing is left to redo)
        st = KC.check_gapfill_banded(ctx, oracle, x4_max=160, dp16_max=420, base_len=70, seed=seed, min_counts=(10, 5 if pct == 100 else 0, 5), pct=pct, redo_pk_min=120)
        assert st['proven'] > 0 and (pct != 100 or st['redo_packed'] > 0)
        seen.update(st['ns_kept'])
    assert seen == {1, 2, 3, 4}, seen


def test_emu_chain_global(ctx, oracle, golden):
    KC.check_chain_global_golden(ctx, oracle, golden, cases=['B', 'D'])


def test_emu_chain_rows_small_window(ctx, oracle, golden, monkeypatch):
    """k_chain_global_rows / k_chain_local_rows with a window of 3 entries instead of 16 (emulator-only hook VMX_RW_WIN): the scan that goes on
    through the index in HBM and the insertion below the window, rare at 16, are taken at almost every anchor — S / P / S_arg, the local
    chains and the records must not change. VMX_CHAIN_ROWS=0 runs the one-wavefront-per-read kernels on the same cases."""
    monkeypatch.setenv('VMX_RW_WIN', '3')
    KC.check_chain_global_golden(ctx, oracle, golden, cases=['B', 'D'])
    KC.check_local_golden(ctx, oracle, golden, cases=['D'])
    KC.check_local_golden(ctx, oracle, golden, cases=['P'])
    KC.check_align_golden(ctx, oracle, golden, cases=['D'])
    KC.check_align_golden(ctx, oracle, golden, cases=['H'], reads=[3])        # mode R: the penalty columns ride in the window too
    monkeypatch.delenv('VMX_RW_WIN')


def test_emu_seed(ctx, oracle, golden):
    KC.check_seed_golden(ctx, oracle, golden, cases=['B', 'D'])


def test_emu_seed_many_hits(ctx, oracle):
    KC.check_seed_many_hits(ctx, oracle, copies=44, unit=2500, read_len=4500, seed=61, min_hits=9000)      # > one 8192-key tile of the emulator build


def test_emu_seed_sparse_noise(ctx, oracle, monkeypatch):
    monkeypatch.setenv('VMX_CLUSTER_SMALL_MAX', '256')          # route these reads to k_cluster_big, whose filtered form is under test
    import ctypes
    f = ctx.lib.L.vmx_emu_cf_count; f.argtypes = [ctypes.c_int]
    t0, d0 = f(0), f(1)
    KC.check_seed_sparse_noise(ctx, oracle, ref_mb=10, read_len=4000, seed=71, min_hits=900)
    assert f(0) - t0 >= 8 and f(1) - d0 <= 12, (f(0) - t0, f(1) - d0)    # the filtered form answered (it declines check_num > 1024 and -1; k_cluster_gen's long form then declines them again)


def test_emu_seed_sparse_noise_long_form(ctx, oracle, monkeypatch):
    """the LONG filtered form inside k_cluster_gen (reads of more than 16383 hits on the GPU: candidates sorted through the tile, read from HBM)"""
    monkeypatch.setenv('VMX_CLUSTER_SMALL_MAX', '256'); monkeypatch.setenv('VMX_CLUSTER_HUGE_MIN', '256')
    import ctypes
    f = ctx.lib.L.vmx_emu_cf_count; f.argtypes = [ctypes.c_int]
    t0, d0 = f(0), f(1)
    KC.check_seed_sparse_noise(ctx, oracle, ref_mb=10, read_len=4000, seed=73, min_hits=900)
    assert f(0) - t0 >= 8 and f(1) - d0 <= 6, (f(0) - t0, f(1) - d0)


def test_emu_local_general_kernel(ctx, oracle, golden, monkeypatch):
    """k_local_seed, the general form that takes the reads k_local_seed_band hands back, on its own (VMX_LSEED_BAND=0)"""
    monkeypatch.setenv('VMX_LSEED_BAND', '0')
    KC.check_local_golden(ctx, oracle, golden, cases=['B', 'D', 'G'])


def test_emu_local_many_chains(ctx, oracle, monkeypatch):
    KC.check_local_many_chains(ctx, oracle, copies=70, unit=500)
    # more guide chains than the general kernel's emission key used to hold (511 until round 4), on both kernels
    KC.check_local_many_chains(ctx, oracle, copies=530, unit=400, seed=93, modes=('S',), min_copies=500)
    monkeypatch.setenv('VMX_LSEED_BAND', '0')
    KC.check_local_many_chains(ctx, oracle, copies=530, unit=400, seed=93, modes=('S',), min_copies=500)


def test_emu_local(ctx, oracle, golden):
    KC.check_local_golden(ctx, oracle, golden, cases=['B', 'D'])
    KC.check_local_golden(ctx, oracle, golden, cases=['P'])                  # mode R (_scar chains), 20 reads


def test_emu_align_end_to_end(ctx, oracle, golden):
    KC.check_align_golden(ctx, oracle, golden, cases=['B'], reads=[0, 1, 2])
    KC.check_align_golden(ctx, oracle, golden, cases=['D'])                   # chimeras, repeat, unmappable, short, 12-base and N-bearing reads
    KC.check_align_golden(ctx, oracle, golden, cases=['I'], reads=[0, 3])     # fix_simple_inv shift, drop_misplaced removal
    KC.check_align_golden(ctx, oracle, golden, cases=['N'])                   # fix_simple_inv's left-flank branch
    KC.check_align_golden(ctx, oracle, golden, cases=['J'], reads=[0, 5, 18]) # mode S (incl. the chimera)
    KC.check_align_golden(ctx, oracle, golden, cases=['K'], reads=[5, 7])     # mode L at k 19 (reads with a strand switch)
    KC.check_align_golden(ctx, oracle, golden, cases=['H'], reads=[3])        # nested SVs from the vacsim-grammar donor, mode R


def test_emu_extend_pools_grow_and_retry(ctx, oracle, golden, monkeypatch):
    """every pool of the extend stage made too small through the test hook (VMX_TEST_EXT_POOL=<mask>:<div>; 1 segment anchors, 2 segments, 4 record blob,
    8 problems per round, 16 problem strings): the read / batch reports it, the batch is run again with the pools x4 until they hold it, and the records
    are those of the reference (the extend stage's capacity ends used to stop at VM_READ_CAPACITY / VM_ERR_OOM)"""
    for mask, div in ((1, 64), (2, 512), (4, 64), (8, 64), (16, 512), (31, 512)):
        monkeypatch.setenv('VMX_TEST_EXT_POOL', '%d:%d' % (mask, div))
        KC.check_align_golden(ctx, oracle, golden, cases=['D'], reads=[1], min_ext_retries=1)      # (one read here: every retry is a whole pass of the emulator; all reads on the GPU)
    monkeypatch.setenv('VMX_TEST_EXT_POOL', '31:2000000000')      # nothing left of any pool: the retries run out (x1024) and the reads are REPORTED, not truncated
    from vacmap_amd.lib import align_batch
    meta, arrays = golden
    gi, _ = KC._case_index(ctx, oracle, meta, arrays, 'D')
    seqs = [arrays['D_r%d_seq' % ri].tobytes().decode() for ri in (1,)]
    try:
        status, recs, stats = align_batch(ctx, gi, ctx.lib.params(meta['D']['mode']), seqs)
        assert int(status[0]) == -20 and not recs and stats['n_ext_retries'] >= 5
    except Exception as e:                                           # (or the batch as a whole: the per-batch pools report through the error)
        assert 'pool' in str(e) or 'memory' in str(e), e
    monkeypatch.delenv('VMX_TEST_EXT_POOL')
    KC.check_align_golden(ctx, oracle, golden, cases=['D'], reads=[1], min_ext_retries=0)


def test_emu_local_stage_without_a_host_wait(oracle, golden):
    """the local stage's fast form (no host wait after a context's first batch) on a context of its own: the second run of the same reads waits less often and
    delivers the same records; with the emulator's small tiles the banded kernel hands reads back, which then go through the side batch"""
    import emu_lib
    from vacmap_amd.lib import Context
    cx = Context(0, lib=emu_lib.context().lib)
    st1 = KC.check_align_golden(cx, oracle, golden, cases=['D'], reads=[0, 1])
    st2 = KC.check_align_golden(cx, oracle, golden, cases=['D'], reads=[0, 1])
    assert st2['n_local_anchors'] == st1['n_local_anchors'] > 0
    KC.check_align_golden(cx, oracle, golden, cases=['H'], reads=[3])
    cx.close()


def test_emu_side_batches_of_the_rare_parts(ctx, oracle, golden, monkeypatch):
    """round 6: reads that need a part of the path a batch does not run (later tiers of the divergence filter, pass 1 = the nofilter re-run of mammap_clrnano.py:24079-24080)
    are run again alone with every part of it; the test hook sends every second read through that side batch — records unchanged"""
    monkeypatch.setenv('VMX_TEST_SIDE_EVERY', '2')
    st = KC.check_align_golden(ctx, oracle, golden, cases=['D'], reads=[0, 1])
    assert st['n_ext_retries'] >= 1
    st = KC.check_align_golden(ctx, oracle, golden, cases=['I'], reads=[0])
    assert st['n_ext_retries'] >= 1
    monkeypatch.delenv('VMX_TEST_SIDE_EVERY')


def test_emu_stage_trace(ctx, oracle, golden):
    """E1 / E3 / E4 stage by stage against the reference's captured segment lists (golden V4)"""
    n = KC.check_stage_trace_golden(ctx, oracle, golden, cases=['I'], reads=[0, 3])
    assert n[0] == 2 and n[3] == 2 and n[5] == 2
    KC.check_stage_trace_golden(ctx, oracle, golden, cases=['N'])               # fix_simple_inv's left-flank branch
    KC.check_stage_trace_golden(ctx, oracle, golden, cases=['B'], reads=[0, 1])


def test_emu_chain_global_fast(ctx, oracle):
    KC.check_chain_global_fast_synth(ctx, oracle, seed=31, n_reads=2, L=140, per_pos=6)


def test_emu_mode_r(ctx, oracle):
    """mode R (fixed-penalty chains with refund, all-chain re-seeding, _scar): GC-fast R on synthetic anchors, whole path on a few reads"""
    KC.check_chain_global_fast_synth(ctx, oracle, seed=33, n_reads=1, L=120, per_pos=6, mode='R')
    assert KC.check_align_random(ctx, oracle, mode='R', n=3, seed=51, reflen=60000, mean_len=2500) >= 3


def test_emu_oom_degrades_to_sub_batches(ctx, oracle, golden, monkeypatch):
    """a batch the device has no memory for is cut in two and tried again (down to single reads) instead of failing with VM_ERR_OOM:
    same records, same order, same per-read status (test hook VMX_TEST_OOM_ABOVE_BASES stands in for a failed hipMalloc)"""
    from vacmap_amd.lib import align_batch
    meta, arrays = golden
    gi, oi = KC._case_index(ctx, oracle, meta, arrays, 'D')
    seqs = [arrays['D_r%d_seq' % ri].tobytes().decode() for ri in (2, 4, 5, 3)]       # incl. the 12-base read that stays unmapped
    st0, rec0, _ = align_batch(ctx, gi, ctx.lib.params('H'), seqs)
    monkeypatch.setenv('VMX_TEST_OOM_ABOVE_BASES', str(max(len(s) for s in seqs) + 10))
    st1, rec1, stats = align_batch(ctx, gi, ctx.lib.params('H'), seqs)
    assert list(st0) == list(st1) and rec0 == rec1 and stats['n_reads'] == len(seqs)
    monkeypatch.setenv('VMX_TEST_OOM_ABOVE_BASES', '5')                                # not even one read fits: the error surfaces
    with pytest.raises(Exception):
        align_batch(ctx, gi, ctx.lib.params('H'), seqs)


def test_emu_mode_asm(ctx, oracle):
    """-mode asm, contigs below 500 kb (the fork's per-read function): records = the reference's goldens = the oracle's"""
    assert KC.check_asm_golden(ctx, oracle, cases=['AS1'], contigs=[3, 8, 9]) == 3       # 30 kb contig with an SV, unmappable, 900-base contig
    assert KC.check_asm_golden(ctx, oracle, cases=['AS5'], contigs=[1]) == 1             # decode_hit's edlib tie-break between two near-identical copies


def test_emu_asm_linked(ctx, oracle):
    """-mode asm's batch-linked chain DPs (k_chain_linked / k_link_carry): the reference's own carried states, then a batch with noise anchors"""
    n_full, n_carry, _ = KC.check_asm_linked_golden(ctx, oracle, cases=['AS3'], max_calls=2)
    assert n_full[0] == 2 and n_full[2] == 2 and n_carry >= 2
    KC.check_asm_linked_noise(ctx, oracle, seed=5, noise_per_anchor=2, which=0)
    KC.check_asm_linked_fast_golden(ctx, oracle)                 # the fork's GC-fast, plain and linked, against the reference's direct calls


def test_emu_mode_asm_long_contig(ctx, oracle, monkeypatch):
    """the long-contig loop of -mode asm (vm_align_asm) with shrunk sizes: linked first round, re-seeded second round, ass_extend_func; then with the
    second round's anchor slots and hit pools made too small (VMX_TEST_ASM_RESEED_DIV): the launch is repeated with larger pools, the records stay
    the reference's (it used to end in VM_READ_CAPACITY for the contig)"""
    assert KC.check_asm_long_golden(ctx, oracle, 'AS3', contigs=[2]) == 1
    monkeypatch.setenv('VMX_TEST_ASM_RESEED_DIV', '256')
    assert KC.check_asm_long_golden(ctx, oracle, 'AS3', contigs=[2]) == 1


def test_emu_mode_asm_long_contig_bail_out(ctx, oracle, monkeypatch):
    """GC-exact's bail-out into the linked GC-fast inside the long-contig loop (mammap_asm.py:23246-23247), forced by the max_factor test hooks"""
    KC.check_asm_long_forced_fast(ctx, oracle, monkeypatch)
