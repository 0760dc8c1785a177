"""GPU parity tests (run with -m gpu on an MI355X): every K1-K4 entry point of libswmarlin.so, called through the
C ABI, against (a) the committed Python big-int fixtures and (b) the C oracle on seeded inputs.  Bit-exact: all
arithmetic on this path is integer."""
import numpy as np
import pytest

from oracle_lib import Oracle, golden, h2i, ints_to_limbs, limbs_to_ints, R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module")
def ctx():
    import simpleworks_amd as swm
    c = swm.Context(0)
    yield c
    c.close()


def _pt(p):
    return None if p is None else (h2i(p[0]), h2i(p[1]))


def _affine_of(ctx, orc, jac):
    xy, inf = ctx.g1_normalize(jac)
    if inf:
        return None
    return orc.points_from_mont(xy.reshape(1, 12))[0]


# ------------------------------------------------------------------------------------------------ primitives
def test_device_field_mul_matches_golden_and_oracle(ctx, orc):
    g = golden("fields.json")
    cases = g["fq"]["cases"]
    a = orc.fq_to_mont(ints_to_limbs([h2i(c["a"]) for c in cases], 6))
    b = orc.fq_to_mont(ints_to_limbs([h2i(c["b"]) for c in cases], 6))
    out = ctx.selftest_mul(0, a, b)
    assert limbs_to_ints(orc.fq_from_mont(out)) == [h2i(c["mul"]) for c in cases]
    cases = g["fr"]["cases"]
    a = orc.fr_mont_from_ints([h2i(c["a"]) for c in cases])
    b = orc.fr_mont_from_ints([h2i(c["b"]) for c in cases])
    out = ctx.selftest_mul(1, a, b)
    assert orc.fr_ints_from_mont(out) == [h2i(c["mul"]) for c in cases]
    # bulk random against the C oracle
    from pyref.prng import fr_array
    x = fr_array(20000, 11)
    y = fr_array(20000, 12)
    from oracle_lib import p64
    ref = np.empty_like(x)
    orc.lib.oracle_fr_mul(p64(x), p64(y), p64(ref), x.shape[0])
    assert np.array_equal(ctx.selftest_mul(1, x, y), ref)
    # Fq bulk: build 6-limb values < q from Fr draws
    xa = np.concatenate([fr_array(6000, 13), fr_array(6000, 14)[:, :2] >> np.uint64(8)], axis=1).copy()
    xb = np.concatenate([fr_array(6000, 15), fr_array(6000, 16)[:, :2] >> np.uint64(8)], axis=1).copy()
    xa[:, 5] &= np.uint64((1 << 56) - 1)
    xb[:, 5] &= np.uint64((1 << 56) - 1)
    ref = np.empty_like(xa)
    orc.lib.oracle_fq_mul(p64(xa), p64(xb), p64(ref), xa.shape[0])
    assert np.array_equal(ctx.selftest_mul(0, xa, xb), ref)


def test_device_lazy_limb_multipliers(ctx):
    """fq28.cuh (the MSM's 14 x 28-bit lazy-carry domain, Montgomery radix 2^392) against Python integers: the plain
    multiplier on operands up to 2^378, and the fused two-product form at its operand bounds (spread-subtracted limbs)."""
    import random
    Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
    rnd = random.Random(5)
    edge = [0, 1, Q - 1, Q, Q + 1, 2 * Q - 1, (1 << 377) - 1, (1 << 378) - 1]
    A = [rnd.randrange(2 * Q) for _ in range(4000)] + [x for x in edge for _ in edge]
    B = [rnd.randrange(2 * Q) for _ in range(4000)] + [y for _ in edge for y in edge]
    inv = pow(1 << 392, -1, Q)
    out = limbs_to_ints(ctx.selftest_mul(2, ints_to_limbs(A, 6), ints_to_limbs(B, 6)))
    assert out == [a * b * inv % Q for a, b in zip(A, B)]
    # which = 6: ((a + 8p)(b + 32p) + (8p - a) b) / 2^392 with all four operands in lazy limb form
    out = limbs_to_ints(ctx.selftest_mul(6, ints_to_limbs(A, 6), ints_to_limbs(B, 6)))
    assert out == [((a + 8 * Q) * (b + 32 * Q) + (8 * Q - a) * b) * inv % Q for a, b in zip(A, B)]


def test_device_group_law(ctx, orc):
    g = golden("g1.json")
    a_pts = [_pt(c["a"]) for c in g["adds"]] + [_pt(c["a"]) for c in g["adds"]]
    b_pts = [_pt(c["b"]) for c in g["adds"]] + [_pt(c["a"]) for c in g["adds"]]  # second half: P + P (doubling)
    exp = [_pt(c["sum"]) for c in g["adds"]] + [_pt(c["dbl_a"]) for c in g["adds"]]
    out = ctx.selftest_g1_add(orc.points_to_mont(a_pts), orc.points_to_mont(b_pts))
    got = [_affine_of(ctx, orc, out[i]) for i in range(len(exp))]
    assert got == exp


# ------------------------------------------------------------------------------------------------ K1
def test_msm_golden(ctx, orc):
    g = golden("msm.json")
    srs = [_pt(p) for p in g["srs_bases"]]
    srs_h = ctx.srs_upload(orc.points_to_mont(srs))
    for c in g["cases"]:
        if c["bases"] == "srs":
            bases = srs_h
        else:
            bases = ctx.srs_upload(orc.points_to_mont([_pt(p) for p in c["bases"]]))
        sc = ints_to_limbs([h2i(s) for s in c["scalars"]], 4)
        res = _affine_of(ctx, orc, ctx.msm_g1(bases, sc))
        assert res == _pt(c["result"]), c["name"]
        if bases is not srs_h:
            bases.free()
    # offset into the resident SRS: sum_i s_i * P_{7+i}
    c = g["cases"][3]  # uniform_31
    sc = ints_to_limbs([h2i(s) for s in c["scalars"]], 4)
    ref = orc.jac_to_affine_int(orc.msm(orc.points_to_mont(srs[7:7 + 31]), sc))
    assert _affine_of(ctx, orc, ctx.msm_g1(srs_h, sc, offset=7)) == ref
    srs_h.free()


@pytest.mark.parametrize("log_n", [10, 13, 16, 17, 18, 19])
def test_msm_vs_oracle(ctx, orc, log_n):
    from pyref.prng import fr_array
    n = 1 << log_n
    g = golden("msm.json")
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(g["tau"]), G)
    bh = ctx.srs_upload(bases)
    sc = fr_array(n, 100 + log_n)
    ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=8))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref
    # Montgomery-form scalars resident in HBM (what the prover feeds after an iNTT)
    d = ctx.to_device(orc.fr_to_mont(sc))
    assert _affine_of(ctx, orc, ctx.msm_g1_dev(bh, d, n, True)) == ref
    d.free()
    # structured scalars: 25 % zeros, 25 % ones, 50 % uniform (bit-heavy witnesses, src/gadgets/traits.rs:150-164)
    st = sc.copy()
    st[0::4] = 0
    st[1::4] = 0
    st[1::4, 0] = 1
    ref = orc.jac_to_affine_int(orc.msm(bases, st, threads=8))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, st)) == ref
    if log_n <= 13:
        ones = np.zeros_like(sc)
        ones[:, 0] = 1
        ref = orc.jac_to_affine_int(orc.msm(bases, ones, threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, ones)) == ref
    bh.free()


def test_msm_linearity_2_20(ctx, orc):
    """Full-size property check (BASELINE config: 2^20 points): MSM(s) + MSM(t) == MSM(s + t mod r)."""
    from pyref.prng import fr_array
    from oracle_lib import p64
    n = 1 << 20
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    s, t = fr_array(n, 201), fr_array(n, 202)
    sm, tm = orc.fr_to_mont(s), orc.fr_to_mont(t)
    um = np.empty_like(sm)
    orc.lib.oracle_fr_add(p64(sm), p64(tm), p64(um), n)
    u = orc.fr_from_mont(um)
    js, jt, ju = ctx.msm_g1(bh, s), ctx.msm_g1(bh, t), ctx.msm_g1(bh, u)
    ssum = np.zeros(18, dtype=np.uint64)
    orc.lib.oracle_g1_add(p64(js), p64(jt), p64(ssum))
    assert orc.jac_to_affine_int(ssum) == orc.jac_to_affine_int(ju)
    assert orc.jac_to_affine_int(ju) is not None
    bh.free()


def test_msm_empty_and_single(ctx, orc):
    g = golden("msm.json")
    srs = [_pt(p) for p in g["srs_bases"]][:8]
    bh = ctx.srs_upload(orc.points_to_mont(srs))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, np.zeros((0, 4), dtype=np.uint64))) is None  # n = 0 -> identity
    one = np.zeros((1, 4), dtype=np.uint64)
    one[0, 0] = 1
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, one, offset=5)) == srs[5]
    import simpleworks_amd as swm
    with pytest.raises(swm.SwmError):  # more scalars than bases behind the offset
        ctx.msm_g1(bh, np.zeros((4, 4), dtype=np.uint64), offset=6)
    bh.free()


def test_msm_pathological_buckets(ctx, orc):
    """All scalars equal / all ones at 2^16: one bucket per window holds every point (oversized-bucket path)."""
    from pyref.prng import fr_array
    n = 1 << 16
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    for val in (1, 0x123456789ABCDEF0123456789ABCDEF):
        sc = np.zeros((n, 4), dtype=np.uint64)
        sc[:, 0] = val & 0xFFFFFFFFFFFFFFFF
        sc[:, 1] = val >> 64
        ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref, hex(val)
    bh.free()


def test_msm_edge_sizes_and_scalars(ctx, orc):
    """SURVEY.md §8d edge cases: n = 1, 31, 32, 33 with scalars drawn from {0, 1, r - 1, one repeated value}."""
    g = golden("msm.json")
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(64, h2i(g["tau"]), G)
    bh = ctx.srs_upload(bases)
    for n in (1, 31, 32, 33):
        for pattern in ([0], [1], [R - 1], [0, 1, R - 1], [0x1234567, 0x1234567]):
            sc = ints_to_limbs([pattern[i % len(pattern)] for i in range(n)], 4)
            ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bases[:n]), sc))
            assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref, (n, pattern)
    bh.free()


def test_msm_structured_2_20(ctx, orc):
    """Full-size schedule (c = 16, 128-point segments) on the inputs that stress it: every point in one bucket per
    window (closed form: bases are [tau^i]G, so MSM(s, s, ..) = [s (tau^n - 1)/(tau - 1)]G) and the 25 % zeros /
    25 % ones / 50 % uniform mix of SURVEY.md §8d against the CPU oracle."""
    from pyref.prng import fr_array
    n = 1 << 20
    tau = h2i(golden("msm.json")["tau"])
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, tau, G)
    bh = ctx.srs_upload(bases)
    geo = (pow(tau, n, R) - 1) * pow(tau - 1, -1, R) % R
    for s in (1, R - 1, 0xDEADBEEFCAFEBABE1234567):
        sc = np.ascontiguousarray(np.tile(ints_to_limbs([s], 4), (n, 1)))
        want = orc.fixed_base_mul(G, ints_to_limbs([s * geo % R], 4), threads=1)
        want_aff = orc.points_from_mont(np.ascontiguousarray(want.reshape(1, 12)))[0]
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == want_aff, hex(s)
    st = fr_array(n, 77)
    st[0::4] = 0
    st[1::4] = 0
    st[1::4, 0] = 1
    ref = orc.jac_to_affine_int(orc.msm(bases, st, threads=8))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, st)) == ref
    bh.free()


# ------------------------------------------------------------------------------------------------ K2
def test_ntt_golden(ctx, orc):
    g = golden("ntt.json")
    for c in g["cases"]:
        x = orc.fr_mont_from_ints([h2i(v) for v in c["in"]])
        y = ctx.ntt_fr(x, c["log_n"], c["inverse"], c["coset"])
        assert orc.fr_ints_from_mont(y) == [h2i(v) for v in c["out"]], c["name"]


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19, 20, 21, 22])
def test_ntt_vs_oracle(ctx, orc, log_n):
    """Every pass plan up to the BASELINE sizes (one, two and three passes; odd and even radices; with and without the
    second scratch buffer), forward / inverse x plain / coset, bit-exact against the C oracle (all host threads)."""
    from pyref.prng import fr_array
    x = orc.fr_to_mont(fr_array(1 << log_n, 300 + log_n))
    for inverse in (0, 1):
        for coset in (0, 1):
            ref = orc.ntt(x, log_n, inverse, coset, threads=orc.lib.oracle_max_threads())
            assert np.array_equal(ctx.ntt_fr(x, log_n, inverse, coset), ref), (log_n, inverse, coset)


def test_ntt_roundtrip_2_22(ctx, orc):
    from pyref.prng import fr_array
    log_n = 22
    x = fr_array(1 << log_n, 322)  # any reduced limbs are valid Montgomery residues
    d = ctx.to_device(x)
    ctx.ntt_fr_dev(d, log_n, False, True)
    y = d.download(x.shape)
    assert not np.array_equal(x, y)
    ctx.ntt_fr_dev(d, log_n, True, True)
    assert np.array_equal(d.download(x.shape), x)
    d.free()


def test_ntt_roundtrip_2_25(ctx, orc):
    """SURVEY.md §8d's largest size: 2^25 elements (1 GiB), forward then inverse is the identity, coset variant too."""
    from pyref.prng import fr_array
    x = np.ascontiguousarray(np.tile(fr_array(1 << 20, 325), (32, 1)))
    x[:, 0] ^= np.arange(x.shape[0], dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.to_device(x)
    for coset in (False, True):
        ctx.ntt_fr_dev(d, 25, False, coset)
        ctx.ntt_fr_dev(d, 25, True, coset)
        assert np.array_equal(d.download(x.shape), x)
    d.free()


def test_ntt_roundtrip_2_24_and_linearity(ctx, orc):
    """BASELINE sizes beyond what the CPU oracle finishes quickly: inverse(forward(x)) == x at 2^24, and
    NTT(x + y) == NTT(x) + NTT(y) at 2^22 (additions on the CPU oracle)."""
    from pyref.prng import fr_array
    from oracle_lib import p64
    x = np.ascontiguousarray(np.tile(fr_array(1 << 20, 324), (16, 1)))
    x[:, 0] ^= np.arange(x.shape[0], dtype=np.uint64)  # break the period of the tiling
    x[:, 3] &= np.uint64((1 << 60) - 1)                 # keep every element < r
    d = ctx.to_device(x)
    ctx.ntt_fr_dev(d, 24, False, False)
    ctx.ntt_fr_dev(d, 24, True, False)
    assert np.array_equal(d.download(x.shape), x)
    d.free()
    n = 1 << 22
    a, b = fr_array(n, 401), fr_array(n, 402)
    c = np.empty_like(a)
    orc.lib.oracle_fr_add(p64(a), p64(b), p64(c), n)
    outs = []
    for v in (a, b, c):
        dv = ctx.to_device(v)
        ctx.ntt_fr_dev(dv, 22, False, True)
        outs.append(dv.download(v.shape))
        dv.free()
    s = np.empty_like(a)
    orc.lib.oracle_fr_add(p64(outs[0]), p64(outs[1]), p64(s), n)
    assert np.array_equal(s, outs[2])


def test_msm_infinity_bases(ctx, orc):
    """The point at infinity (x = y = 0) is a valid base (ark-ec VariableBaseMSM accepts zero bases): first, middle,
    last and only position of a bucket, at a size where whole segments run through the 28-bit fast path."""
    from pyref.prng import fr_array
    g = golden("msm.json")
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    n = 4096
    bases = orc.srs_bases(n, h2i(g["tau"]), G)
    for holes in ([0], [n - 1], [0, 1, 2, 3], list(range(5, n, 97)), list(range(0, n, 2))):
        b = bases.copy()
        b[holes] = 0
        bh = ctx.srs_upload(b)
        # every scalar equal: one bucket per window holds every point, so the holes sit first / inside / last in a chain
        for sc in (fr_array(n, 300), np.ascontiguousarray(np.tile(ints_to_limbs([0xABCDEF123], 4), (n, 1)))):
            keep = np.ones(n, dtype=bool)
            keep[holes] = False
            ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bases[keep]), np.ascontiguousarray(sc[keep]), threads=4))
            assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref, holes[:4]
        # with an offset into the base set the mask is read at the shifted position
        sc = fr_array(64, 301)
        keep = np.array([(3 + i) not in set(holes) for i in range(64)])
        ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bases[3:67][keep]), np.ascontiguousarray(sc[keep])))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc, offset=3)) == ref
        bh.free()
    # a base set large enough for the precomputed-window schedule (>= 2^17 points): the mask is honoured there, too
    n = 1 << 17
    bases = orc.srs_bases(n, h2i(g["tau"]), G)
    holes = [0, 1, 77, n // 2, n - 1] + list(range(1000, n, 4099))
    b = bases.copy()
    b[holes] = 0
    bh = ctx.srs_upload(b)
    keep = np.ones(n, dtype=bool)
    keep[holes] = False
    for sc in (fr_array(n, 302), np.ascontiguousarray(np.tile(ints_to_limbs([0xABCDEF123], 4), (n, 1)))):
        ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bases[keep]), np.ascontiguousarray(sc[keep]), threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref
    bh.free()
    only = np.zeros((1, 12), dtype=np.uint64)
    bh = ctx.srs_upload(only)
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, ints_to_limbs([12345], 4))) is None
    bh.free()


def test_msm_rejects_non_canonical_scalars(ctx, orc):
    """include/swmarlin.h: scalars must be canonical (< r); r itself, r + 1 and 2^256 - 1 fail with INVALID_ARG."""
    import simpleworks_amd as swm
    g = golden("msm.json")
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(64, h2i(g["tau"]), G)
    bh = ctx.srs_upload(bases)
    ok = ints_to_limbs([R - 1] * 64, 4)
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, ok)) is not None
    for bad in (R, R + 1, (1 << 256) - 1, 1 << 255):
        sc = ok.copy()
        sc[17] = ints_to_limbs([bad], 4)[0]
        with pytest.raises(swm.SwmError) as e:
            ctx.msm_g1(bh, sc)
        assert e.value.code == -1 and "canonical" in str(e.value)
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, ok)) is not None  # the context stays usable
    bh.free()


def test_msm_vs_oracle_2_22(ctx, orc):
    """BASELINE configs[3] size: 2^22 points, uniform scalars, bit-exact against the C oracle (all host threads)."""
    from pyref.prng import fr_array
    n = 1 << 22
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    sc = fr_array(n, 422)
    ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=orc.lib.oracle_max_threads()))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref
    bh.free()


def test_entry_points_from_another_thread(ctx, orc):
    """A context is bound to its GPU, not to the thread that created it: every entry point selects the context's device
    (a fresh thread has device 0 current here, but no HIP state of its own; the call must still work and agree)."""
    import threading
    from pyref.prng import fr_array
    x = orc.fr_to_mont(fr_array(1 << 12, 9))
    want = orc.ntt(x, 12, 0, 0, 2)
    out = {}

    def run():
        try:
            out["ntt"] = ctx.ntt_fr(x, 12)
            d = ctx.to_device(x)
            ctx.ntt_fr_dev(d, 12)
            out["dev"] = d.download(x.shape)
            d.free()
        except Exception as e:  # surfaced in the main thread
            out["err"] = e
    t = threading.Thread(target=run)
    t.start()
    t.join()
    assert "err" not in out, out.get("err")
    assert np.array_equal(out["ntt"], want) and np.array_equal(out["dev"], want)


# ------------------------------------------------------------------------------------------------ K3 / K4
def test_spmv_golden_and_random(ctx, orc):
    s = golden("misc.json")["spmv"]
    rowptr, col = np.array(s["rowptr"], dtype=np.uint32), np.array(s["col"], dtype=np.uint32)
    val, z = orc.fr_mont_from_ints([h2i(x) for x in s["val"]]), orc.fr_mont_from_ints([h2i(x) for x in s["z"]])
    assert orc.fr_ints_from_mont(ctx.spmv_fr(rowptr, col, val, z)) == [h2i(x) for x in s["out"]]
    # random sparsity (nnz/row ~ Poisson(3)), empty and ragged rows, coefficients equal to one
    from pyref.prng import fr_array
    rng = np.random.default_rng(5)
    rows, cols = 50000, 40000
    cnt = rng.poisson(3, rows).astype(np.uint32)
    cnt[::7] = 0
    rowptr = np.zeros(rows + 1, dtype=np.uint32)
    rowptr[1:] = np.cumsum(cnt)
    nnz = int(rowptr[-1])
    col = rng.integers(0, cols, nnz, dtype=np.uint32)
    val = orc.fr_to_mont(fr_array(nnz, 41))
    one = orc.fr_mont_from_ints([1])[0]
    val[::3] = one
    z = orc.fr_to_mont(fr_array(cols, 42))
    assert np.array_equal(ctx.spmv_fr(rowptr, col, val, z), orc.spmv(rowptr, col, val, z))
    # a few very long rows among short ones (the shape of a transposed R1CS matrix): the schedule picked from the longest
    # row changes, the bits must not
    cnt = rng.poisson(2, rows).astype(np.uint32)
    cnt[0], cnt[5], cnt[777], cnt[rows - 1] = 30000, 100, 65, 2049
    rowptr[1:] = np.cumsum(cnt)
    nnz = int(rowptr[-1])
    col = rng.integers(0, cols, nnz, dtype=np.uint32)
    val = orc.fr_to_mont(fr_array(nnz, 43))
    val[::5] = one
    assert np.array_equal(ctx.spmv_fr(rowptr, col, val, z), orc.spmv(rowptr, col, val, z))
    # empty matrix
    assert ctx.spmv_fr(np.zeros(5, dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.zeros((0, 4), np.uint64), z).sum() == 0


def test_batch_inverse_and_vec_mul(ctx, orc):
    from oracle_lib import p64
    g = golden("misc.json")["batch_inverse"]
    v = orc.fr_mont_from_ints([h2i(x) for x in g["in"]])
    assert orc.fr_ints_from_mont(ctx.batch_inverse_fr(v)) == [h2i(x) for x in g["out"]]
    from pyref.prng import fr_array
    x = orc.fr_to_mont(fr_array(100003, 51))
    x[17] = 0
    x[100002] = 0
    ref = x.copy()
    orc.lib.oracle_batch_inverse_fr(p64(ref), ref.shape[0])
    assert np.array_equal(ctx.batch_inverse_fr(x), ref)
    # whole lanes / whole workgroups of zeros, a single element, an all-zero vector (zeros stay zero)
    x2 = orc.fr_to_mont(fr_array(3 * 4096 + 5, 53))
    x2[32:64] = 0
    x2[4096:8192] = 0
    for v in (x2, x2[:1].copy(), np.zeros((77, 4), dtype=np.uint64)):
        ref = v.copy()
        orc.lib.oracle_batch_inverse_fr(p64(ref), ref.shape[0])
        assert np.array_equal(ctx.batch_inverse_fr(v), ref)
    y = orc.fr_to_mont(fr_array(100003, 52))
    ref = np.empty_like(x)
    orc.lib.oracle_fr_mul(p64(x), p64(y), p64(ref), x.shape[0])
    assert np.array_equal(ctx.vec_mul_fr(x, y), ref)


def test_msm_table_schedule_offsets_and_shapes(ctx, orc):
    """A resident base set with precomputed-window tables: MSMs at an offset, of uneven length, with structured scalars,
    through the flat schedule (60000 points: the low-latency variant with its lane-group fold of multi-segment buckets)
    and — on the same set — through the per-window schedule that MSMs too small for the shared bucket set keep
    (700 points, read from the first table row)."""
    from pyref.prng import fr_array
    n = 1 << 17
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    big = ctx.srs_upload(bases)
    for off, m in ((12345, 60000), (4321, 700)):
        sub = np.ascontiguousarray(bases[off:off + m])
        for seed, shape in ((1, "uniform"), (2, "bits"), (3, "equal")):
            sc = fr_array(m, 900 + seed)
            if shape == "bits":
                sc[0::2] = 0
                sc[0::4, 0] = 1
            elif shape == "equal":
                sc[:] = sc[7]
            ref = orc.jac_to_affine_int(orc.msm(sub, sc, threads=8))
            assert _affine_of(ctx, orc, ctx.msm_g1(big, sc, offset=off)) == ref, (shape, m)
    big.free()


@pytest.mark.parametrize("n", [512, 1000, 4096, 12288, 40000])
def test_msm_small_sets_low_latency_schedule(ctx, orc, n):
    """Base sets from 512 points carry tables too (c = lg n + 2): short segments, lane-group fold, one bucket per lane.
    Uniform, bit-heavy, all-equal and sparse scalars against the oracle, full length and a ragged sub-range."""
    from pyref.prng import fr_array
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    for seed, shape in ((1, "uniform"), (2, "bits"), (3, "equal"), (4, "sparse")):
        sc = fr_array(n, 7000 + seed + n)
        if shape == "bits":
            sc[0::2] = 0
            sc[0::4, 0] = 1
        elif shape == "equal":
            sc[:] = sc[3]
        elif shape == "sparse":
            sc[:] = 0
            sc[5::97] = fr_array(len(sc[5::97]), 1)
        ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref, shape
        lo, m = n // 3, n - n // 3 - 5
        ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bases[lo:lo + m]), np.ascontiguousarray(sc[:m]), threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, np.ascontiguousarray(sc[:m]), offset=lo)) == ref, shape
    bh.free()


def test_msm_uniform_2_20_vs_oracle(ctx, orc):
    """The headline MSM size against the C restatement of VariableBaseMSM (2^19 and 2^22 are covered above)."""
    from pyref.prng import fr_array
    n = 1 << 20
    G = orc.points_to_mont([_pt(golden("g1.json")["generator"])])
    bases = orc.srs_bases(n, h2i(golden("msm.json")["tau"]), G)
    bh = ctx.srs_upload(bases)
    sc = fr_array(n, 2020)
    ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=8))
    assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref
    bh.free()


def test_msm_bases_outside_the_prime_order_subgroup(ctx, orc):
    """VariableBaseMSM takes ANY curve points.  The twisted Edwards tables are only valid for the prime-order subgroup, so a
    base set with points outside it (here: curve points with small x, cofactor not cleared, mixed with subgroup points and
    an identity) must be detected at upload and served by the XYZZ tables — same result as the oracle."""
    from pyref.prng import fr_array
    from pyref.bls12_377 import Q, fq_sqrt, g1_mul_fast, R
    n = 2048
    pts, x = [], 2
    while len(pts) < n:
        y = fq_sqrt((x * x * x + 1) % Q)
        if y is not None:
            pts.append((x, y))
        x += 1
    assert g1_mul_fast(pts[0], R) is not None          # not in the subgroup
    G = _pt(golden("g1.json")["generator"])
    pts[5] = G
    pts[6] = None                                      # identity base
    pts[7] = g1_mul_fast(G, 12345)
    bases = orc.points_to_mont(pts)
    bh = ctx.srs_upload(bases)
    for seed in (1, 2):
        sc = fr_array(n, 4000 + seed)
        ref = orc.jac_to_affine_int(orc.msm(bases, sc, threads=8))
        assert _affine_of(ctx, orc, ctx.msm_g1(bh, sc)) == ref
    bh.free()
