"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/tstwo_hip.h declares, the
product path fails loudly without a GPU (no CPU fallback), the host-side field/circle mirror agrees with the
oracle, and the multi-GPU layer (column sharding + all-gather of Merkle roots) works at world_size 2 on gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import P, ROOT, load_vectors, rand_column
from oracle import oracle as orc

import tstwo_amd as T
from tstwo_amd import _lib as L

OL = orc.lib()


def test_capi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tstwo_hip.h")).read()
    declared = set(re.findall(r"\b(tstwo_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    lib = L.lib()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/tstwo_hip.h but not exported"
    assert declared == set(L.EXPORTS)
    assert lib.tstwo_version().startswith(b"tstwo_hip")
    assert lib.tstwo_merkle_layers_bytes(3) == 32 * 15


def _header_prototypes():
    """{symbol: (return kind, [arg kinds])} parsed from include/tstwo_hip.h.  Kinds: 'ptr', 'u32', 'i32', 'u64'."""
    src = open(os.path.join(ROOT, "include", "tstwo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(?:^|;|\})\s*((?:const\s+)?[A-Za-z_0-9]+\s*\**)\s*(tstwo_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?=;)", src, flags=re.S | re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()

        def kind(t):
            t = " ".join(t.split())
            if "*" in t or "[" in t:
                return "ptr"
            base = re.sub(r"\bconst\b", "", t).split()[0]
            return {"uint32_t": "u32", "int": "i32", "size_t": "u64", "uint64_t": "u64"}[base]
        kinds = [] if args in ("", "void") else [kind(a) for a in args.split(",")]
        protos[name] = (kind(ret + " x") if "*" not in ret else "ptr", kinds)
    return protos


def test_ts_and_ctypes_bindings_match_the_header_prototypes():
    """Three descriptions of the C ABI must agree argument by argument: the header's prototypes, the ctypes binding every
    -m gpu test calls through (tstwo_amd/_lib.py), and ts/backend/hip/ffi.ts (the bun:ffi stub of INTEGRATION.md, which
    cannot run in this image).  Scalars must have the header's width and signedness; a pointer is `u64` in ffi.ts exactly
    where ctypes passes a raw device address (c_void_p) and `P` exactly where ctypes passes host memory."""
    import ctypes as C
    ts = open(os.path.join(ROOT, "ts", "backend", "hip", "ffi.ts")).read()
    protos = _header_prototypes()
    assert set(protos) == set(L.EXPORTS), set(protos) ^ set(L.EXPORTS)

    def ctypes_kind(t):
        if t is C.c_void_p:
            return "dev"
        if t in (C.c_uint32,):
            return "u32"
        if t in (C.c_int,):
            return "i32"
        if t in (C.c_size_t, C.c_uint64):
            return "u64"
        return "host"           # POINTER(...), arrays of pointers, c_char_p: host memory
    for sym in L.EXPORTS:
        m = re.search(rf"\b{sym}: \{{ args: \[([^\]]*)\], returns: (\w+)", ts)
        assert m, f"{sym} missing from ffi.ts"
        ts_args = [a.strip() for a in m.group(1).split(",") if a.strip()]
        ret_kind, h_args = protos[sym]
        assert len(ts_args) == len(h_args), (sym, ts_args, h_args)
        assert m.group(2) == {"i32": "i32", "u64": "u64", "ptr": "cstring"}[ret_kind], (sym, m.group(2), ret_kind)
        for i, (t, h) in enumerate(zip(ts_args, h_args)):
            ok = (h == "ptr" and t in ("P", "u64")) or (h != "ptr" and t == h)
            assert ok, f"{sym} argument {i}: ffi.ts says {t}, the header says {h}"
        if sym in L._SIGS:
            c_args = [ctypes_kind(t) for t in L._SIGS[sym]]
            for i in L.HOST_VOID_ARGS.get(sym, ()):
                assert c_args[i] == "dev"
                c_args[i] = "host"
            assert len(c_args) == len(h_args), (sym, c_args, h_args)
            for i, (c, h, t) in enumerate(zip(c_args, h_args, ts_args)):
                assert (c in ("dev", "host")) == (h == "ptr"), f"{sym} argument {i}: ctypes {c} vs header {h}"
                if c in ("u32", "i32", "u64"):
                    assert c == h, f"{sym} argument {i}: ctypes {c} vs header {h}"
                want = {"dev": "u64", "host": "P"}.get(c, c)
                assert t == want, f"{sym} argument {i}: ffi.ts {t}, ctypes binding implies {want}"


def test_no_cpu_fallback_without_gpu():
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(L.TstwoError, match="no HIP device|no CPU fallback"):
        L.init(0)
    with pytest.raises(L.TstwoError):
        T.HipColumn([1, 2, 3])


def test_missing_rccl_is_an_error_not_a_crash():
    """TSTWO_RCCL_LIB pointing nowhere: tstwo_comm_unique_id must return TSTWO_ERR_COMM with a message (round 2's code built
    the message from two dlerror() calls, the second of which returns NULL -> std::string(NULL) -> SIGSEGV)."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from tstwo_amd import _lib as L\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = L.lib().tstwo_comm_unique_id(buf)\n"
            "print(rc, L.lib().tstwo_last_error().decode())\n" % ROOT)
    env = dict(os.environ, TSTWO_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    rc, msg = r.stdout.strip().split(" ", 1)
    assert int(rc) == 9 and "RCCL is not available" in msg and "/nonexistent/librccl.so.1" in msg


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tstwo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_host_fields_against_vectors():
    for v in load_vectors("m31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op in ("add", "sub", "mul"):
            assert getattr(T.M31(i["a"]), op)(T.M31(i["b"])).value == out
        elif op == "neg":
            assert T.M31(i["a"]).neg().value == out
        elif op == "from_i32":
            assert T.M31.from_(i["value"]).value == out
        elif op == "reduce":
            assert T.M31.reduce(int(i["value"])).value == out
        elif op == "partial_reduce":
            assert T.M31.partialReduce(i["value"]).value == out
        elif op == "inverse":
            assert T.M31(i["value"]).inverse().value == out
    for v in load_vectors("qm31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op in ("add", "sub", "mul"):
            assert list(getattr(T.QM31.from_u32_unchecked(*i["a"]), op)(T.QM31.from_u32_unchecked(*i["b"])).tup()) == out
        elif op == "inverse":
            assert list(T.QM31.from_u32_unchecked(*i["value"]).inverse().tup()) == out
        elif op == "mul_cm31":
            assert list(T.QM31.from_u32_unchecked(*i["qm31"]).mul_cm31(T.CM31.from_u32_unchecked(*i["cm31"])).tup()) == out
    for v in load_vectors("cm31"):
        op, i, out = v["operation"], v["inputs"], v["output"]
        if op == "mul":
            r = T.CM31.from_u32_unchecked(i["a_real"], i["a_imag"]).mul(T.CM31.from_u32_unchecked(i["b_real"], i["b_imag"]))
            assert r.tup() == (out["real"], out["imag"])
        elif op == "inverse":
            assert T.CM31.from_u32_unchecked(i["real"], i["imag"]).inverse().tup() == (out["real"], out["imag"])
    with pytest.raises(ZeroDivisionError, match="0 has no inverse"):
        T.M31(0).inverse()
    with pytest.raises(ValueError):
        T.M31.from_u32_unchecked(P)


def test_host_circle_against_oracle():
    for k in (0, 1, 5, 21):
        c = T.Coset.half_odds(k)
        assert c.initial_index.value == OL.orc_half_odds_initial(k)
        for i in (0, 1, 3):
            if i < c.size():
                p, o = c.at(i), OL.orc_coset_at(c.initial_index.value, k, i)
                assert (p.x.value, p.y.value) == (o.x, o.y)
    d = T.CanonicCoset(5).circleDomain()
    assert d.isCanonic() and d.log_size() == 5
    for i in range(32):
        p, o = d.at(i), OL.orc_circle_domain_at(d.halfCoset.initial_index.value, 4, i)
        assert (p.x.value, p.y.value) == (o.x, o.y)
    root = T.Coset.half_odds(10)
    assert T.Coset.half_odds(7).is_doubling_of(root) and not T.Coset.half_odds(11).is_doubling_of(root)
    assert root.double().equals(T.Coset(root.initial_index.mul(2), 9))
    assert T.bit_reverse_index(6, 3) == 3
    T.LineDomain(T.Coset.half_odds(3))
    with pytest.raises(ValueError, match="not unique"):
        T.LineDomain(T.Coset.subgroup(3).shift(T.CirclePointIndex.subgroup_gen(3)))
    g = T.SECURE_FIELD_CIRCLE_GEN
    assert g.x.square().add(g.y.square()) == T.QM31.one()          # on the circle


def test_quotient_constants_match_oracle():
    px, py = T.SECURE_FIELD_CIRCLE_GEN.x, T.SECURE_FIELD_CIRCLE_GEN.y
    v, alpha = T.QM31.from_u32_unchecked(7, 8, 9, 10), T.QM31.from_u32_unchecked(1, 2, 3, 4)
    from tstwo_amd.quotients import complexConjugateLineCoeffs
    a, b, c = complexConjugateLineCoeffs(T.CirclePoint(px, py), v, alpha)
    out = (orc.QM31 * 3)()
    OL.orc_line_coeffs(orc.SPoint(orc.q(px.tup()), orc.q(py.tup())), orc.q(v.tup()), orc.q(alpha.tup()), out)
    assert [a.tup(), b.tup(), c.tup()] == [o.tup() for o in out]
    assert T.generate_secure_powers(alpha, 3) == [T.QM31.one(), alpha, alpha.mul(alpha)]
    assert T.generate_secure_powers(alpha, 0) == []


def test_shard_columns():
    assert T.shard_columns(256, 8, 3) == list(range(96, 128))
    got = sum((T.shard_columns(10, 4, r) for r in range(4)), [])
    assert got == list(range(10))
    assert T.shard_columns(3, 4, 3) == []


_WORKER = r"""
import os, sys, hashlib
sys.path.insert(0, os.environ["TSTWO_ROOT"])
import numpy as np
import torch.distributed as dist
from tstwo_amd.distributed import allgather_roots, commit_sharded
from oracle import oracle as orc     # test infrastructure: stands in for the GPU commit on this GPU-less box
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cols = [np.random.default_rng(100 + c).integers(0, 2**31 - 1, size=64, dtype=np.uint32) for c in range(8)]
class Tree:
    def __init__(self, cs): self.r = orc.merkle_commit(cs, [6] * len(cs))[1]
    def root(self): return self.r
tree, roots = commit_sharded(cols, rank, world, Tree)
expect = [orc.merkle_commit(cols[4 * r:4 * r + 4], [6] * 4)[1] for r in range(world)]
assert roots == expect, (rank, roots, expect)
assert roots[rank] == tree.root()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_allgather_roots_world2_gloo(tmp_path):
    """N>1 path on CPU: 2 ranks, column shards [0,4) and [4,8), roots all-gathered in rank order (gloo)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, TSTWO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


_WORKER_ROWS = r"""
import hashlib, os, sys
sys.path.insert(0, os.environ["TSTWO_ROOT"])
import numpy as np
import torch.distributed as dist
from tstwo_amd.distributed import commit_rows_sharded, shard_rows
from oracle import oracle as orc     # test infrastructure: stands in for the GPU commit on this GPU-less box
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
LOG = 7
cols = [np.random.default_rng(200 + c).integers(0, 2**31 - 1, size=1 << LOG, dtype=np.uint32) for c in range(4)]
small = np.random.default_rng(300).integers(0, 2**31 - 1, size=1 << (LOG - 2), dtype=np.uint32)
class Tree:
    def __init__(self, cs): self.r = orc.merkle_commit(cs, [int(c.size).bit_length() - 1 for c in cs])[1]
    def root(self): return self.r
def rows(c):
    s, n = shard_rows(c.size, world, rank)
    return c[s:s + n]
tree, subroots, root = commit_rows_sharded([rows(c) for c in cols] + [rows(small)], Tree)
_, expect = orc.merkle_commit(cols + [small], [LOG] * 4 + [LOG - 2])
assert root == expect, (rank, root.hex(), expect.hex())
assert subroots[rank] == tree.root() and len(subroots) == world
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_row_sharded_merkle_world2_gloo(tmp_path):
    """SURVEY 8(e) row sharding on CPU: 2 ranks hash contiguous leaf ranges of a mixed-size tree to subtree roots,
    all-gather them (gloo) and compute the top level redundantly; the result equals the single-tree root."""
    script = tmp_path / "worker_rows.py"
    script.write_text(_WORKER_ROWS)
    env = dict(os.environ, TSTWO_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_combine_subtree_roots_and_shard_rows():
    import hashlib
    from tstwo_amd.distributed import combine_subtree_roots, shard_rows
    r = [bytes([i]) * 32 for i in range(4)]
    h = lambda a, b: hashlib.blake2s(a + b).digest()
    assert combine_subtree_roots(r) == h(h(r[0], r[1]), h(r[2], r[3]))
    assert combine_subtree_roots(r[:1]) == r[0]
    assert shard_rows(64, 4, 3) == (48, 16)
    with pytest.raises(ValueError):
        shard_rows(8, 4, 0)          # shards of 2 rows: below the 4-row alignment of the fold kernels
    with pytest.raises(ValueError):
        combine_subtree_roots(r[:3])


def test_allgather_roots_single_process():
    from tstwo_amd.distributed import allgather_roots
    assert allgather_roots(b"\x01" * 32) == [b"\x01" * 32]


def test_blake2s_channel_mirror():
    """channel/blake2.ts: mix_u64(4) from the zero digest reproduces the genuine Rust-side transcript digest
    (test-equivalence/.../comprehensive_rust_test_vectors.json:478); draws are deterministic and reduced."""
    c = T.Blake2sChannel()
    assert c.digest() == bytes(32)
    c.mix_u64(4)
    assert c.digest().hex().startswith("af0e8a72") and c.digest().hex().endswith("2ac5")
    assert c.n_challenges == 1 and c.n_sent == 0
    a = c.draw_felt()
    assert c.n_sent == 1 and all(0 <= v < P for v in a.tup())
    c2 = T.Blake2sChannel(); c2.mix_u64(4)
    assert c2.draw_felt() == a
    b = c.draw_felt()                                                   # Rust semantics: a fresh 8-word draw per felt
    assert c.n_sent == 2 and b != a and c2.draw_felt() == b
    ct = T.Blake2sChannel(ts_compat=True); ct.mix_u64(4)
    assert ct.draw_felt() == a                                          # TS port: second felt = the 4 words left in the queue ...
    ct.mix_root(bytes(32))
    stale = ct.draw_felt()
    assert ct.n_sent == 0                                               # ... even across a mix (no new draw happened)
    c3 = T.Blake2sChannel(); c3.mix_u64(4)
    f9 = c3.draw_felts(9)
    assert c3.n_sent == 5 and f9[0] == a and f9[1] == stale             # draw_felts: consecutive words of the same draws
    d0 = c.digest()
    c.mix_root(bytes(range(32)))
    import hashlib
    assert c.digest() == hashlib.blake2s(d0 + bytes(range(32))).digest() and c.n_sent == 0
    c.mix_felts([T.QM31.from_u32_unchecked(1, 2, 3, 4)])
    with pytest.raises(TypeError):
        c.mix_u32s([2**32])
    cfg = T.FriConfig(2, 1, 3)
    assert cfg.last_layer_domain_size() == 8 and cfg.security_bits() == 3
    with pytest.raises(ValueError):
        T.FriConfig(11, 1, 3)


# ---------------------------------------------------------------- fri.test.ts "FRI Implementation" host-only pieces
def test_fri_host_sparse_evaluation_and_rebuild():
    from tstwo_amd.fri_verifier import (InsufficientWitnessError, SparseEvaluation, accumulate_line,
                                        compute_decommitment_positions_and_rebuild_evals)
    q = lambda v: T.QM31.from_(T.M31(v))
    s = SparseEvaluation([[T.QM31.one()] * 2, [T.QM31.zero()] * 2], [0, 1])
    assert len(s.subset_evals) == 2 and len(s.subset_domain_initial_indexes) == 2
    with pytest.raises(ValueError, match=r"All subset evaluations must have length equal to 2\^FOLD_STEP"):
        SparseEvaluation([[T.QM31.one()], [T.QM31.zero(), T.QM31.one()]], [0, 1])
    with pytest.raises(ValueError, match="Number of subset evaluations must match number of domain indexes"):
        SparseEvaluation([[T.QM31.one()] * 2], [0, 1])
    # rebuild: queries 0 and 2 of a log-2 domain, witnesses fill positions 1 and 3
    pos, sp = compute_decommitment_positions_and_rebuild_evals(T.Queries([0, 2], 2), [q(1), q(3)], iter([q(2), q(4)]), 1)
    assert pos == [0, 1, 2, 3]
    assert [[e.tup()[0] for e in ev] for ev in sp.subset_evals] == [[1, 2], [3, 4]]
    assert sp.subset_domain_initial_indexes == [0, 1]            # bit_reverse(0, 2), bit_reverse(2, 2)
    with pytest.raises(InsufficientWitnessError):
        compute_decommitment_positions_and_rebuild_evals(T.Queries([0, 2], 2), [q(1), q(3)], iter([]), 1)
    layer, colv, alpha = [q(1), q(2)], [q(3), q(4)], q(5)
    accumulate_line(layer, colv, alpha)
    assert [e.tup()[0] for e in layer] == [1 * 25 + 3, 2 * 25 + 4]


def test_fri_host_degree_bounds_and_config():
    assert T.CirclePolyDegreeBound(7).fold_to_line().log_degree_bound == 6
    b = T.LinePolyDegreeBound(5)
    assert b.fold(2).log_degree_bound == 3 and b.fold(6) is None
    cfg = T.FriConfig(3, 2, 10)
    assert cfg.last_layer_domain_size() == 1 << 5 and cfg.security_bits() == 20
    with pytest.raises(ValueError):
        T.FriConfig(11, 2, 1)
    with pytest.raises(ValueError):
        T.FriConfig(1, 0, 1)


def test_fri_host_sparse_fold_matches_oracle():
    """SparseEvaluation.fold_line / fold_circle of one 2-element coset == the oracle's fold of the whole layer at that index."""
    n = 5
    cols = [rand_column(900 + k, 1 << n) for k in range(4)]
    alpha = T.QM31.from_u32_unchecked(19283, 1, 2, 3)
    from tstwo_amd.fri_verifier import SparseEvaluation
    from tstwo_amd.circle import bit_reverse_index
    at = lambda i: T.QM31.from_u32_unchecked(*(int(c[i]) for c in cols))
    # line layer on half_odds(n)
    dom = T.LineDomain(T.Coset.half_odds(n))
    folded = orc.fold_line(cols, n, dom.coset().initial_index.value, alpha.tup())
    for pair in (0, 3, 9, 15):
        s = SparseEvaluation([[at(2 * pair), at(2 * pair + 1)]], [bit_reverse_index(2 * pair, n)])
        assert s.fold_line(alpha, dom)[0].tup() == tuple(int(c[pair]) for c in folded)
    # circle layer on CanonicCoset(n)
    cd = T.CanonicCoset(n).circleDomain()
    zero = [np.zeros(1 << (n - 1), dtype=np.uint32)] * 4
    foldc = orc.fold_circle_into_line(zero, cols, n, cd.halfCoset.initial_index.value, alpha.tup())
    for pair in (0, 5, 12):
        s = SparseEvaluation([[at(2 * pair), at(2 * pair + 1)]], [bit_reverse_index(2 * pair, n)])
        assert s.fold_circle(alpha, cd)[0].tup() == tuple(int(c[pair]) for c in foldc)


def test_line_poly_host():
    """line.test.ts: ordered <-> bit-reversed coefficients; eval_at_point = sum c_k * basis_k(x) with pi(x) = 2x^2 - 1."""
    co = [T.QM31.from_u32_unchecked(k + 1, 2 * k, 3, k * k) for k in range(8)]
    p = T.LinePoly.from_ordered_coefficients(co)
    assert [c.tup() for c in p.into_ordered_coefficients()] == [c.tup() for c in co]
    x = T.QM31.from_u32_unchecked(5, 6, 7, 8)
    pi = lambda v: v.square().double().sub(T.QM31.one())
    basis = [T.QM31.one(), x, pi(x), x.mul(pi(x)), pi(pi(x)), x.mul(pi(pi(x))), pi(x).mul(pi(pi(x))), x.mul(pi(x)).mul(pi(pi(x)))]
    want = T.QM31.zero()
    for c, b in zip(co, basis):
        want = want.add(c.mul(b))
    assert p.eval_at_point(x).tup() == want.tup()
    with pytest.raises(ValueError, match="coeffs length must be power of two"):
        T.LinePoly(co[:3])


def test_blake2s_channel_rust_digest_kats():
    """test/channel/channel_exact_rust_tests.test.ts:84-120 — digests copied from the Rust tests (test_mix_u64, test_mix_u32s)."""
    c = T.Blake2sChannel()
    c.mix_u64(0x1111222233334444)
    c2 = T.Blake2sChannel()
    c2.mix_u32s([0x33334444, 0x11112222])
    assert c.digest() == c2.digest() == bytes([
        0xbc, 0x9e, 0x3f, 0xc1, 0xd2, 0x4e, 0x88, 0x97, 0x95, 0x6d, 0x33, 0x59, 0x32, 0x73, 0x97, 0x24, 0x9d, 0x6b, 0xca, 0xcd, 0x22,
        0x4d, 0x92, 0x74, 0x4, 0xe7, 0xba, 0x4a, 0x77, 0xdc, 0x6e, 0xce])
    c3 = T.Blake2sChannel()
    c3.mix_u32s([1, 2, 3, 4, 5, 6, 7, 8, 9])
    assert c3.digest() == bytes([
        0x70, 0x91, 0x76, 0x83, 0x57, 0xbb, 0x1b, 0xb3, 0x34, 0x6f, 0xda, 0xb6, 0xb3, 0x57, 0xd7, 0xfa, 0x46, 0xb8, 0xfb, 0xe3, 0x2c,
        0x2e, 0x43, 0x24, 0xa0, 0xff, 0xc2, 0x94, 0xcb, 0xf9, 0xa1, 0xc7])
    # channel_time (Rust test_channel_time): draw_random_bytes -> n_sent 1; draw_felts(9) -> 5 more draws
    c4 = T.Blake2sChannel()
    c4.draw_random_bytes()
    assert (c4.n_challenges, c4.n_sent) == (0, 1)
    c4.draw_felts(9)
    assert (c4.n_challenges, c4.n_sent) == (0, 6)
    with pytest.raises(TypeError):
        c4.mix_u32s([-1])
    with pytest.raises(TypeError):
        c4.mix_u64(-1)


def test_get_query_positions_by_log_size_kat():
    """test/fri/get_query_positions_by_log_size.test.ts:5-12."""
    q = T.Queries.from_positions([1, 3, 5, 7], 3)
    res = T.get_query_positions_by_log_size(q, {3, 2})
    assert res[3] == [1, 3, 5, 7] and res[2] == [0, 1, 2, 3]


def test_decommit_requests_planner_against_oracle_tree():
    """vcs.decommit_requests (the pure index walk of vcs/prover.ts:32-109, used by the row-sharded decommit): served from
    an oracle-built tree it yields a decommitment the verifier accepts; a wrong witness is rejected."""
    from tstwo_amd.vcs import decommit_requests
    rng = np.random.default_rng(5)
    logs = [5, 5, 3, 5, 3]
    cols = [rng.integers(0, P, size=1 << lg, dtype=np.uint32) for lg in logs]
    layers, root = orc.merkle_commit(cols, logs)
    queries = {5: [0, 7, 8, 30], 3: [2, 5]}
    hreq, qreq, wreq = decommit_requests(5, logs, queries)
    hashes = [bytes(layers[lg][node]) for lg, node in hreq]
    queried = [T.M31(int(cols[c][node])) for c, node in qreq]
    colwit = [T.M31(int(cols[c][node])) for c, node in wreq]
    dec = T.MerkleDecommitment(hashes, colwit)
    T.MerkleVerifier(T.Blake2sMerkleHasher, root, logs).verify(queries, queried, dec)
    assert len(queried) == 4 * 3 + 2 * 2
    bad = T.MerkleDecommitment([hashes[0][::-1]] + hashes[1:], colwit)
    with pytest.raises(ValueError, match="Root mismatch"):
        T.MerkleVerifier(T.Blake2sMerkleHasher, root, logs).verify(queries, queried, bad)


def test_lazy_proof_sequences_behave_like_lists():
    """M31Values / HashSlices (vcs.py): what a decommitment holds until somebody looks at it — list semantics for readers, and a
    real list from the first mutation on (the verifier tests tamper with proofs)."""
    import copy

    from tstwo_amd.fields import M31
    from tstwo_amd.vcs import HashSlices, M31Values
    m = M31Values([5, 6, 7])
    assert len(m) == 3 and m[0] == M31(5) and m[-1] == M31(7) and m[1:] == [M31(6), M31(7)]
    assert list(m) == [M31(5), M31(6), M31(7)] and m == [M31(5), M31(6), M31(7)]
    assert m + [M31(1)] == [M31(5), M31(6), M31(7), M31(1)] and [M31(1)] + m == [M31(1), M31(5), M31(6), M31(7)]
    with pytest.raises(IndexError):
        m[3]
    d = copy.deepcopy(m)
    d.append(M31(9)); d[0] = M31(0)
    assert len(d) == 4 and d[0] == M31(0) and d.pop() == M31(9) and len(m) == 3 and m[0] == M31(5)
    raw = bytes(range(96))
    h = HashSlices(raw, 3)
    assert len(h) == 3 and h[1] == raw[32:64] and h[-1] == raw[64:] and list(h) == [raw[:32], raw[32:64], raw[64:]]
    g = copy.deepcopy(h)
    g[2] = bytes(32); g.pop(0)
    assert list(g) == [raw[32:64], bytes(32)] and list(h) == [raw[:32], raw[32:64], raw[64:]]


def test_bit_reverse_perm_matches_bit_reverse_index():
    from tstwo_amd.circle import bit_reverse_index, bit_reverse_perm
    for lg in range(0, 11):
        assert bit_reverse_perm(lg).tolist() == [bit_reverse_index(i, lg) for i in range(1 << lg)]
    assert not bit_reverse_perm(5).flags.writeable


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    """bench.py --gpus N under a launcher that started another number of ranks must fail, never print a line for another N
    (round 3 parsed --gpus and ignored it).  The check runs before anything imports torch or touches the GPU library."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "0"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_bench_launcher_relays_rank_failure(tmp_path):
    """launch_ranks: N fresh interpreters with RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout passed through, a failing rank
    turns into a non-zero exit and stops the others."""
    import subprocess
    import textwrap
    sys.path.insert(0, ROOT)
    script = tmp_path / "launch.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        if "WORLD_SIZE" in os.environ:                      # a rank
            r = int(os.environ["RANK"])
            assert os.environ["WORLD_SIZE"] == "3" and os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
            print("line from rank", r, flush=True)
            if sys.argv[1] == "fail" and r == 1:
                sys.exit(7)
            if sys.argv[1] == "fail":
                time.sleep(60)                                # must be stopped by the launcher, not waited for
            sys.exit(0)
        import bench
        bench.__file__ = {str(script)!r}
        sys.exit(bench.launch_ranks(3, sys.argv[1:]))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, str(script), "ok"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0 and p.stdout.strip() == "line from rank 0"
    assert "line from rank 1" in p.stderr and "line from rank 2" in p.stderr
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, str(script), "fail"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 7 and time.time() - t0 < 40 and "rank 1 exited with 7" in p.stderr


def test_shipped_library_has_no_experiment_switches():
    """The drop-in .so must not change results or kernel plans because of its caller's environment (round 3: 36 TSTWO_* variables,
    one of which skipped a CFFT pass).  The shipped build keeps the loader / allocator variables and nothing else; the tuning and
    A/B switches exist only in the experiments build (-DTSTWO_EXPERIMENTS), where they are read once."""
    import re
    from tstwo_amd import _lib as L
    allowed = {"TSTWO_ALLOC", "TSTWO_NO_POOL", "TSTWO_POISON", "TSTWO_ALLOW_UNSAFE_ASYNC_ALLOC", "TSTWO_ASYNC_RELEASE", "TSTWO_RCCL_LIB"}
    names = lambda path: set(m.decode() for m in re.findall(rb"TSTWO_[A-Z0-9_]+", open(path, "rb").read()))
    shipped = {n for n in names(L.LIB_PATH) if not n.startswith(("TSTWO_ERR_", "TSTWO_ALLOC_"))}
    assert shipped <= allowed, sorted(shipped - allowed)
    if os.path.exists(L.LIB_EXP_PATH):
        assert {"TSTWO_CFFT_GENERIC", "TSTWO_MERKLE_SUBTREE", "TSTWO_FRI_NO_TAIL"} <= names(L.LIB_EXP_PATH)
    # and in the sources: no getenv outside context.hip's allocator / comm.hip's loader and the #ifdef TSTWO_EXPERIMENTS block
    csrc = os.path.join(ROOT, "tstwo_amd", "csrc")
    outside = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".cuh", ".h")):
            continue
        depth = 0
        for line in open(os.path.join(csrc, f)):
            t = line.strip()
            if t.startswith("#ifdef TSTWO_EXPERIMENTS"):
                depth += 1
            elif depth and t.startswith(("#else", "#endif")):
                depth -= 1
            elif not depth and "getenv(" in line and not t.startswith("//"):
                assert f in ("context.hip", "comm.hip"), (f, line)
                outside += 1
    assert outside <= 8, outside
