"""GPU parity tests (run with -m gpu on an MI355X): the real libvacmapx.so through its C-ABI vs the oracle / goldens."""
import numpy as np
import pytest
import kernel_cases as KC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from vacmap_amd.lib import Context
    return Context(0)     # raises loudly when the HIP library or the device is missing


def test_tables_on_device(ctx, oracle):
    KC.check_tables(ctx, oracle)


def test_edit_distance(ctx, oracle):
    KC.check_edit_distance(ctx, oracle, n=64, maxlen=900, seed=11)
    KC.check_edit_distance(ctx, oracle, n=6, maxlen=9000, seed=12, minlen=4100)
    KC.check_edit_distance(ctx, oracle, n=2, maxlen=17000, seed=13, minlen=15000)


def test_extend(ctx, oracle):
    KC.check_extend(ctx, oracle, n=200, seed=14)
    KC.check_extend(ctx, oracle, n=8, seed=15, maxlen=6000)


def test_gapfill(ctx, oracle):
    KC.check_gapfill(ctx, oracle, n=200, maxlen=500, seed=16)
    KC.check_gapfill(ctx, oracle, n=6, maxlen=3000, seed=17)


def test_chain_global_golden(ctx, oracle, golden):
    KC.check_chain_global_golden(ctx, oracle, golden)


def test_seed_golden(ctx, oracle, golden):
    KC.check_seed_golden(ctx, oracle, golden)


def test_local_golden(ctx, oracle, golden):
    KC.check_local_golden(ctx, oracle, golden)


def test_align_golden(ctx, oracle, golden):
    KC.check_align_golden(ctx, oracle, golden)
