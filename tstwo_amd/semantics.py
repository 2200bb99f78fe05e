"""Which of the reference's two definitions the host mirror follows where its TypeScript port and the Rust stwo text
it carries disagree (DESIGN.md §2 "Reference quirks" 2 and 7):

  * QM31.complexConjugate / the quotient line coefficients and denominators — Rust: conj(a + bu) = a - bu, Pr = c0, Pi = c1;
    the TS port conjugates each CM31 (fields/qm31.ts:433-435) and reads Pr / Pi from c0.real / c0.imag
    (backend/cpu/quotients.ts:168-174);
  * Blake2sChannel.draw_felt — Rust draws 8 fresh base felts per call and drops 4; the TS port keeps the 4 unused ones in a
    queue that survives later mix_*() calls (channel/blake2.ts:177-184);
  * SecureCirclePoly.evalAtPoint — Rust combines the four coordinate evaluations; the TS port returns coordinate 0 only
    (poly/circle/secure_poly.ts:14-18).

ONE process-wide setting decides, and every function that used to default its `ts_compat` argument to False now defaults
to it:

    tstwo_amd.set_semantics("rust")   # default: Rust stwo (the only semantics under which the quotients are low degree and
                                       # every challenge depends on everything mixed before it)
    tstwo_amd.set_semantics("ts")     # transcripts / quotients interchangeable with the TypeScript reference's CpuBackend

The default is therefore NOT transcript-compatible with the TypeScript reference prover or verifier; a deployment that
swaps CpuBackend for HipBackend inside the TS prover and must interoperate with TS-produced proofs selects "ts"
(INTEGRATION.md).  An explicit ts_compat=True/False argument still overrides the setting for one call.  The device kernels
are the same in both modes; only host-side constants and the channel differ.  Parity of the Rust mode beyond the field
vectors is pinned by three Rust transcript digests only (tests/test_cpu_host.py): see DESIGN.md "parity unpinned".
"""
_MODE = "rust"


def set_semantics(mode: str) -> None:
    global _MODE
    if mode not in ("rust", "ts"):
        raise ValueError('semantics must be "rust" or "ts"')
    _MODE = mode


def get_semantics() -> str:
    return _MODE


def ts_compat(flag=None) -> bool:
    """Resolves an optional per-call override against the process-wide setting."""
    return (_MODE == "ts") if flag is None else bool(flag)
