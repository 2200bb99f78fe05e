"""-m gpu: BASELINE configs 3 and 4 AT FULL SIZE against the C oracle, every output word (config 5 at full size:
tests/test_gpu_config5.py).  The oracle restates the reference's per-row formulation (per-row to_point + Fermat inverse in
the fold, batchInverse per row in the quotients), so these take a few seconds of host time each.

  config 3: accumulateQuotients, 4 columns x 2^22, one sample batch at SECURE_FIELD_CIRCLE_GEN, random_coeff qm31(1,2,3,4)
            + QM31 batch inverse of a 2^22 SoA column                     (backend/cpu/quotients.ts:52-116, fields/fields.ts:66-207)
  config 4: fold_circle_into_line of a 2^24 secure column into a zero line, alpha = qm31(19283,1,2,3)
            + Blake2s Merkle commit of its 4 coordinate columns           (fri.ts:162-192, vcs/prover.ts:13-30)
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

from tstwo_amd import _lib as L  # noqa: E402
from bench import splitmix_column  # noqa: E402
from gpu_util import dev, dev_empty, host, p4, ptrs, vp  # noqa: E402

SECURE_GEN = ((1, 0, 478637715, 513582971), (992285211, 649143431, 740191619, 1186584352))      # circle.ts:143-146
OL = orc.lib()


@pytest.fixture(scope="module", autouse=True)
def _init():
    L.init(0)
    yield
    L.sync()


def test_config3_quotients_and_qm31_inverse_every_row_vs_oracle():
    n, N = 22, 1 << 22
    half = OL.orc_half_odds_initial(n - 1)
    cols = [splitmix_column(4 + c, N) for c in range(4)]
    vals = [(7 + c, 8, 9, 10) for c in range(4)]                         # any sampled values: the kernel is tested, not the protocol
    d = [dev(c) for c in cols]
    out = [dev_empty(N) for _ in range(4)]
    L.call("tstwo_quotients_accumulate_samples", half, n, ptrs(d), 4, 1, L.u32x([0, 4]), L.u32x([0, 1, 2, 3]),
           L.u32x(list(SECURE_GEN[0]) + list(SECURE_GEN[1])), L.u32x([x for v in vals for x in v]), L.u32x([1, 2, 3, 4]), p4(out))
    exp = orc.accumulate_quotients(half, n, cols, (1, 2, 3, 4), [(SECURE_GEN[0], SECURE_GEN[1], [(c, vals[c]) for c in range(4)])])
    for k in range(4):
        got = host(out[k], N)
        assert (got == exp[k]).all(), f"quotient coordinate {k}: {int((got != exp[k]).sum())} rows differ"
    sec = [splitmix_column(8 + c, N) for c in range(4)]
    for c in sec:
        c[c == 0] = 1
    ds = [dev(c) for c in sec]
    L.call("tstwo_qm31_batch_inverse", p4(ds), p4(out), N)
    exp = orc.qm31_batch_inverse(sec)
    for k in range(4):
        assert (host(out[k], N) == exp[k]).all(), f"inverse coordinate {k}"


def test_config4_fold_and_merkle_every_row_vs_oracle():
    n, N = 24, 1 << 24
    half = OL.orc_half_odds_initial(n - 1)
    src = [splitmix_column(9 + c, N) for c in range(4)]
    alpha = (19283, 1, 2, 3)
    tw, itw = dev_empty(N // 2), dev_empty(N // 2)
    L.call("tstwo_twiddles_build", half, n - 1, vp(tw), vp(itw))
    ds = [dev(c) for c in src]
    dst = [dev_empty(N // 2) for _ in range(4)]
    for b in dst:
        b.zero()
    L.call("tstwo_fri_fold_circle_into_line", p4(dst), N // 2, p4(ds), n, vp(itw), n - 1, L.u32x(alpha))
    exp = orc.fold_circle_into_line([np.zeros(N // 2, dtype=np.uint32)] * 4, src, n, half, alpha)
    for k in range(4):
        got = host(dst[k], N // 2)
        assert (got == exp[k]).all(), f"fold coordinate {k}: {int((got != exp[k]).sum())} rows differ"
    layers = L.DeviceBuffer(32 * ((2 << n) - 1))
    root = (C.c_uint8 * 32)()
    L.call("tstwo_merkle_commit", ptrs(ds), L.u32x([n] * 4), 4, C.c_void_p(layers.ptr), root)
    assert bytes(root) == orc.mt_merkle_root(src, n, 16)
    # the inner FRI layer's tree (log 23) over the folded line as well
    L.call("tstwo_merkle_commit", ptrs(dst), L.u32x([n - 1] * 4), 4, C.c_void_p(layers.ptr), root)
    assert bytes(root) == orc.mt_merkle_root(exp, n - 1, 16)
