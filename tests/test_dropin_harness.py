"""CPU test, BUILD CONTAINER ONLY (skipped where /root/reference is absent, i.e. on the GPU box): the reference's own Python running on
top of the `vacmap_index`-shaped shim (vacmap_amd/aligner.py -> vm_map / vm_k_cigar / vm_edit_distance; emulator build of the
product's kernels) reproduces the reference's testdata answer (README.md:124: three alignments + 0-6811, - 6830-23034, + 23015-29829
on chr1) and the golden V6 records. tools/harness/dropin_run.py; nothing of the reference is copied."""
import io, os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir('/root/reference/src/vacmap'), reason='needs the reference checkout (build container only)')
def test_reference_python_runs_on_the_shim(oracle):
    sys.path.insert(0, os.path.join(ROOT, 'tools', 'harness'))
    import dropin_run
    log = io.StringIO()
    total, same, calls = dropin_run.run(('A',), log=log)
    assert total == same == 1, log.getvalue()
    assert "('chr1', '+', 0, 6811), ('chr1', '-', 6830, 23034), ('chr1', '+', 23015, 29829)" in log.getvalue()
    # every native primitive the reference calls went through the C-ABI entries that carry its call shape
    assert calls['map'] >= 1 and calls['k_cigar_global'] >= 100 and calls['k_cigar_zdrop'] >= 1 and calls['edlib'] >= 1
