"""-m gpu: a fixed-seed slice of the randomised GPU-vs-oracle sweep (tests/fuzz_parity.py) inside the collected suite, so that
the sweep's coverage — random shapes of every C-ABI entry point, random FRI / PCS configurations through prove -> verify — is
re-run by the driver and not only by the builder.  Longer sweeps: python tests/fuzz_parity.py --seconds N [--big]."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_slice_small_shapes():
    import fuzz_parity
    # a mismatch raises inside run(); the count only has to show that every generator ran (a slow box runs fewer cases
    # under the wall-clock limit: that is not a parity failure)
    assert fuzz_parity.run(seconds=60, seed=20261004, max_cases=4000, verbose=False) >= fuzz_parity.N_GENERATORS


def test_fuzz_slice_tiled_sizes():
    import fuzz_parity
    assert fuzz_parity.run(seconds=120, seed=7, big=True, max_cases=40, verbose=False) >= fuzz_parity.N_GENERATORS_BIG
