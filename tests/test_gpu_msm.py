"""End-to-end MSM parity: HIP path vs the CPU oracle on the same inputs, compared as canonical affine
(x, y) -- the reference's criterion (src/metal/msm.rs:604-608, :555-558)."""
import random

import pytest

from oracle import bn254_ref as o
from helpers import h2c_instance_bytes, small_instance

pytestmark = pytest.mark.gpu


def _expect(points, scalars):
    return o.msm_pippenger(scalars, points, 8) if len(points) > 64 else o.msm_naive(scalars, points)


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 100, 257, 1000])
def test_gpu_msm_h2c_small(cfg, msm_pkg, n):
    pts, sc = small_instance(100 + n, n)
    sb, pb = h2c_instance_bytes(pts, sc)
    out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    assert o.decode_jacobian_mont_le(out) == _expect(pts, sc)
    # output is normalised: z == R mod p (or 0)
    z = int.from_bytes(out[64:96], "little")
    assert z in (0, o.MONT_R % o.P)


@pytest.mark.parametrize("c", [3, 4, 7, 8, 11, 13, 15, 16, 17])
def test_window_sizes_agree(cfg, msm_pkg, c):
    pts, sc = small_instance(7, 300)
    sb, pb = h2c_instance_bytes(pts, sc)
    cfg.set_window_size(c)
    try:
        out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    finally:
        cfg.set_window_size(0)
    assert o.decode_jacobian_mont_le(out) == _expect(pts, sc)


def test_edge_scalars_and_points(cfg, msm_pkg):
    rng = random.Random(5)
    pts, sc = small_instance(9, 200)
    # zero scalars, scalar 1, r-1, repeated points, P and -P with equal scalars, identity points
    sc[0] = 0
    sc[1] = 1
    sc[2] = o.R_ORDER - 1
    for i in range(10, 40):
        sc[i] = 0
    pts[50] = pts[51] = pts[52]
    sc[50] = sc[51] = sc[52]                 # same bucket in every window -> P+P doubling path
    pts[60] = o.aff_neg(pts[61])
    sc[60] = sc[61]                          # P + (-P) inside a bucket
    pts[70] = None                           # halo2curves identity (0,0)
    pts[71] = None
    sc[80] = (1 << 14) + 1                   # the reference's "breaking scalar" (prepare_buckets_indices.rs:132-135)
    sb, pb = h2c_instance_bytes(pts, sc)
    for c in (0, 15, 5):
        cfg.set_window_size(c)
        try:
            out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
        finally:
            cfg.set_window_size(0)
        assert o.decode_jacobian_mont_le(out) == o.msm_pippenger(sc, pts, 8)


def test_all_zero_scalars_gives_identity(cfg, msm_pkg):
    pts, _ = small_instance(3, 64)
    sb, pb = h2c_instance_bytes(pts, [0] * 64)
    out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    assert o.decode_jacobian_mont_le(out) is None


def test_all_scalars_equal_skewed_buckets(cfg, msm_pkg):
    """One bucket per window holds every point (load-balance worst case, SURVEY section 7 hard part 2)."""
    pts, _ = small_instance(4, 500)
    k = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % o.R_ORDER
    sb, pb = h2c_instance_bytes(pts, [k] * 500)
    out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    acc = None
    for p in pts:
        acc = o.jac_add(acc, o.to_jac(p))
    assert o.decode_jacobian_mont_le(out) == o.scalar_mul(k, o.to_affine(acc))


def test_layouts_agree(cfg, msm_pkg):
    """ark projective / ark affine / reference BE32 wire layout give the same result as h2c."""
    rng = random.Random(21)
    pts, sc = small_instance(33, 150)
    pts[5] = None
    expect = _expect(pts, sc)
    # ark projective with random z (metal_msm, msm.rs:220)
    from helpers import rand_jac
    proj = b"".join(o.encode_projective_ark(rand_jac(rng, p)) for p in pts)
    sb = b"".join(o.encode_scalar_h2c(k) for k in sc)
    assert o.decode_jacobian_mont_le(msm_pkg.metal_msm(proj, sb, cfg)) == expect
    # ark affine {x, y, infinity}
    aff = b""
    for p in pts:
        if p is None:
            aff += bytes(64) + b"\x01" + bytes(7)
        else:
            aff += o.encode_affine_h2c(p) + bytes(8)
    out = cfg.msm(sb, aff, len(pts), msm_pkg.SCALAR_MONT_LE, msm_pkg.POINT_ARK_AFFINE)
    assert o.decode_jacobian_mont_le(out) == expect
    # reference wire layout: canonical BE32 scalars, Jacobian BE32 points
    import struct
    sbe = b"".join(struct.pack("<8I", *o.encode_scalar_be32(k)) for k in sc)
    pbe = b"".join(struct.pack("<24I", *o.encode_point_be32(rand_jac(rng, p))) for p in pts)
    out = cfg.msm(sbe, pbe, len(pts), msm_pkg.SCALAR_CANON_BE32, msm_pkg.POINT_JAC_BE32)
    assert o.decode_jacobian_mont_le(out) == expect
    # canonical LE scalars
    scl = b"".join(o.int_to_le_bytes32(k) for k in sc)
    pb = b"".join(o.encode_affine_h2c(p) for p in pts)
    out = cfg.msm(scl, pb, len(pts), msm_pkg.SCALAR_CANON_LE, msm_pkg.POINT_H2C_AFFINE)
    assert o.decode_jacobian_mont_le(out) == expect


def test_batch_and_errors(cfg, msm_pkg):
    insts = [small_instance(40 + i, n) for i, n in enumerate([17, 64, 129])]
    enc = [h2c_instance_bytes(p, s) for p, s in insts]
    outs = cfg.msm_batch([e[0] for e in enc], [e[1] for e in enc], [17, 64, 129])
    for (p, s), out in zip(insts, outs):
        assert o.decode_jacobian_mont_le(out) == _expect(p, s)
    with pytest.raises(msm_pkg.MsmError) as ei:
        cfg.msm(b"", b"", 0)
    assert ei.value.status == msm_pkg.INPUT_ERROR
    with pytest.raises(msm_pkg.MsmError):
        cfg.set_window_size(18)


def test_device_generator_matches_oracle(cfg, msm_pkg):
    n = 64
    dp, ds = cfg.generate_instance(o.SEED_BASE + 3, n, True)
    try:
        pb = cfg.to_host(dp, 64 * n)
        sb = cfg.to_host(ds, 32 * n)
        pts, sc = o.gen_instance(o.SEED_BASE + 3, n)
        assert pb == b"".join(o.encode_affine_h2c(p) for p in pts)
        assert sb == b"".join(o.encode_scalar_h2c(k) for k in sc)
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        assert o.decode_jacobian_mont_le(out) == o.msm_naive(sc, pts)
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_persistent_bases_match_the_per_call_path(cfg, msm_pkg):
    """Bases converted once and kept resident (SURVEY §8b staged variant / §8f N4) give the same bytes as the
    per-call conversion the reference does (msm.rs:152-153), for every input layout and for several scalar sets."""
    n = 700
    pts, sc = small_instance(4242, n)
    pts[5] = None                                            # an identity among the bases
    sb, pb = h2c_instance_bytes(pts, sc)
    want = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    assert o.decode_jacobian_mont_le(want) == _expect(pts, sc)
    proj = b"".join(o.encode_projective_ark(o.to_jac(p) if p else None) for p in pts)
    handles = [cfg.bases_upload(pb, n), cfg.bases_upload(proj, n, msm_pkg.POINT_ARK_PROJECTIVE)]
    d_raw = cfg.alloc(len(pb))
    cfg.to_device(d_raw, pb)
    handles.append(cfg.bases_prepare_device(d_raw, n))
    try:
        for h in handles:
            assert cfg.msm_prepared(sb, h, n) == want
        # the same resident bases with other scalars, through the device entry point
        sc2 = [(k * 7 + 1) % o.R_ORDER for k in sc]
        sb2, _ = h2c_instance_bytes(pts, sc2)
        d_sc = cfg.alloc(len(sb2))
        cfg.to_device(d_sc, sb2)
        (out2,) = cfg.msm_batch_device([d_sc], [handles[0]], [n], point_layout=msm_pkg.POINT_PREPARED)
        assert out2 == msm_pkg.gpu_msm_h2c(sb2, pb, cfg)
        cfg.free(d_sc)
        # prepared arrays are device memory: a HOST point buffer under that layout is refused, not dereferenced
        with pytest.raises(msm_pkg.MsmError) as e:
            cfg.msm(sb, pb, n, point_layout=msm_pkg.POINT_PREPARED)
        assert e.value.status == msm_pkg.INPUT_ERROR
        # host scalars + resident bases through the host-buffer batch entry point (only scalars cross PCIe)
        assert cfg.msm_batch([sb, sb2], [handles[0], handles[1]], [n, n], point_layout=msm_pkg.POINT_PREPARED) == \
            [want, out2]
    finally:
        for h in handles:
            cfg.free(h)
        cfg.free(d_raw)


def test_pipelined_submit_wait_and_ticket_errors(cfg, msm_pkg):
    """submit_batch_device / wait_batch: several batches in flight give the same bytes as blocking calls, in any
    collection order; a fifth batch, a stale ticket and a bogus ticket are refused with INPUT_ERROR."""
    n = 3000
    insts = []
    for j in range(4):
        dp, ds = cfg.generate_instance(o.SEED_BASE + 600 + j, n, True)
        insts.append((dp, ds))
    try:
        want = [cfg.msm_batch_device([ds], [dp], [n])[0] for dp, ds in insts]
        handles = [cfg.submit_batch_device([ds, ds], [dp, dp], [n, n - 7]) for dp, ds in insts]   # 4 batches of 2
        with pytest.raises(msm_pkg.MsmError) as e:
            cfg.submit_batch_device([insts[0][1]], [insts[0][0]], [n])                             # no free slot
        assert e.value.status == msm_pkg.INPUT_ERROR
        for j in (2, 0, 3, 1):
            got = cfg.wait_batch(handles[j])
            assert got[0] == want[j]
            assert got[1] == cfg.msm_batch_device([insts[j][1]], [insts[j][0]], [n - 7])[0]
        with pytest.raises(msm_pkg.MsmError) as e:
            cfg.wait_batch(handles[1])                                                             # already collected
        assert e.value.status == msm_pkg.INPUT_ERROR
        with pytest.raises(msm_pkg.MsmError):
            cfg.wait_batch((99, handles[0][1], 1))
    finally:
        for dp, ds in insts:
            cfg.free(dp)
            cfg.free(ds)


def test_one_ctx_from_two_threads(cfg, msm_pkg):
    """Calls on one ctx are serialised internally (the reference takes a process-wide mutex, msm.rs:248-255):
    two host threads hammering the same ctx get the right, identical answers."""
    import threading
    pts, sc = small_instance(777, 900)
    sb, pb = h2c_instance_bytes(pts, sc)
    want = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    assert o.decode_jacobian_mont_le(want) == _expect(pts, sc)
    results, errors = [], []

    def work():
        try:
            for _ in range(10):
                results.append(msm_pkg.gpu_msm_h2c(sb, pb, cfg))
        except Exception as ex:   # noqa: BLE001 - surfaced below
            errors.append(ex)

    ts = [threading.Thread(target=work) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors
    assert len(results) == 20 and all(r == want for r in results)


@pytest.mark.parametrize("n,c", [(700, 0), (700, 7), (700, 15), (700, 16), (3000, 19), (64, 21), (33, 4)])
def test_precomputed_window_tables_match_the_per_call_path(cfg, msm_pkg, n, c):
    """tables 2^(c w) P_i built once; every window then shares one bucket set (u16 digits up to
    c = 15, u32 digits above).  Same bytes as the per-call pipeline for several scalar sets, incl. an identity base,
    repeated points and scalars whose signed digits carry through every window."""
    pts, sc = small_instance(5000 + n + c, n)
    pts[3] = None
    pts[7] = pts[8]
    sc[0], sc[1], sc[2] = o.R_ORDER - 1, (1 << 253) - 1, 0
    sc[7] = sc[8]
    sb, pb = h2c_instance_bytes(pts, sc)
    want = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
    assert o.decode_jacobian_mont_le(want) == _expect(pts, sc)
    t = cfg.tables_build(pb, n, window_size=c)
    try:
        info = cfg.tables_info(t)
        assert info["n"] == n and info["num_windows"] == 254 // info["window_size"] + 1
        assert info["device_bytes"] == 64 * n * info["num_windows"]
        if c:
            assert info["window_size"] == c
        assert cfg.msm_tables(sb, t) == want
        sc2 = [(k * 3 + 5) % o.R_ORDER for k in sc]
        sb2, _ = h2c_instance_bytes(pts, sc2)
        d_sc = cfg.alloc(len(sb2))
        cfg.to_device(d_sc, sb2)
        (got,) = cfg.msm_batch_device([d_sc], [t], [n], point_layout=msm_pkg.POINT_TABLES)
        cfg.free(d_sc)
        assert got == msm_pkg.gpu_msm_h2c(sb2, pb, cfg)
    finally:
        cfg.tables_free(t)


def test_precomputed_tables_errors(cfg, msm_pkg):
    pts, sc = small_instance(91, 50)
    sb, pb = h2c_instance_bytes(pts, sc)
    with pytest.raises(msm_pkg.MsmError):
        cfg.tables_build(pb, 50, window_size=22)
    t = cfg.tables_build(pb, 50)
    try:
        d_sc = cfg.alloc(len(sb))
        cfg.to_device(d_sc, sb)
        with pytest.raises(msm_pkg.MsmError) as e:                       # n differs from the table's
            cfg.msm_batch_device([d_sc], [t], [49], point_layout=msm_pkg.POINT_TABLES)
        assert e.value.status == msm_pkg.INPUT_ERROR
        with pytest.raises(msm_pkg.MsmError):                            # a device pointer is not a table handle
            cfg.msm_batch_device([d_sc], [d_sc], [50], point_layout=msm_pkg.POINT_TABLES)
        cfg.free(d_sc)
    finally:
        cfg.tables_free(t)
    with pytest.raises(msm_pkg.MsmError):
        cfg.tables_free(t)                                               # already freed


def test_known_answers_without_the_oracle(cfg, msm_pkg):
    """The GPU path against constants that do not come from this repository: 2G and 3G on alt_bn128 (the EIP-196
    ecMul / ecAdd vectors).  Only the layout encoding is the oracle module's."""
    g2 = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
          9918110051302171585080402603319702774565515993150576347155970296011118125764)
    g3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
          19321533766552368860946552437480515441416830039777911637913418824951667761761)
    G = (1, 2)
    sb, pb = h2c_instance_bytes([G], [2])
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_msm_h2c(sb, pb, cfg)) == g2
    sb, pb = h2c_instance_bytes([G, g2], [1, 1])
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_msm_h2c(sb, pb, cfg)) == g3
    sb, pb = h2c_instance_bytes([G, g2, g3], [o.R_ORDER - 3, o.R_ORDER - 3, 3])    # (r-3)(G + 2G) + 3(3G) = 0
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_msm_h2c(sb, pb, cfg)) is None
    sb, pb = h2c_instance_bytes([G] * 3, [1, 1, 1])                                  # three equal points: 3G
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_msm_h2c(sb, pb, cfg)) == g3


@pytest.mark.parametrize("c", [0, 5, 15])
def test_doubling_inside_the_mixed_addition_gathers_the_base_again(cfg, msm_pkg, c):
    """The exceptional case q == p of the MIXED addition (accumulator already a general point): the accumulate kernel has
    given q's registers to the next gather by then and gathers q again (pti_madd_tail's reload, k_accumulate.hip).  Bases
    P, P, 2P, 4P, 8P with one scalar share every bucket: P -> (affine start, doubling) 2P -> (mixed addition, doubling)
    4P -> 8P -> 16P; with the scalar r - k every digit changes sign and the re-gathered base must be negated again.
    Surrounded by ordinary points so that the items have neighbours in their waves."""
    pts, sc = small_instance(31, 120)
    P = pts[0]
    chain = [P, P] + [o.scalar_mul(2 ** j, P) for j in (1, 2, 3)]
    for k in (1, 5, (1 << 14) + 1, o.R_ORDER - 5, o.R_ORDER - (1 << 33) - 7):
        allp = pts[1:60] + chain + pts[60:]
        alls = sc[1:60] + [k] * len(chain) + sc[60:]
        sb, pb = h2c_instance_bytes(allp, alls)
        cfg.set_window_size(c)
        try:
            out = msm_pkg.gpu_msm_h2c(sb, pb, cfg)
        finally:
            cfg.set_window_size(0)
        assert o.decode_jacobian_mont_le(out) == o.msm_naive(alls, allp), (c, k)
    # the chain alone: every bucket it touches holds nothing else
    sb, pb = h2c_instance_bytes(chain, [7] * len(chain))
    assert o.decode_jacobian_mont_le(msm_pkg.gpu_msm_h2c(sb, pb, cfg)) == o.scalar_mul(7 * 16, P)


@pytest.mark.parametrize("c", [0, 5, 15, 16, 17])
def test_canonical_scalars_at_or_above_r_are_reduced(cfg, msm_pkg, c):
    """Raw 256-bit integers in the canonical layouts (instance files, FFI callers) may exceed r; the result must be
    that of the scalar mod r for every window size (bit 255 used to be dropped at c = 15 / 17)."""
    pts, sc = small_instance(21, 40)
    raw = list(sc)
    raw[0] = o.R_ORDER + 5
    raw[1] = (1 << 256) - 1
    raw[2] = o.R_ORDER
    raw[3] = (1 << 255) + 12345
    raw[4] = 5 * o.R_ORDER + 1
    pb = b"".join(o.encode_affine_h2c(p) for p in pts)
    sb = b"".join(k.to_bytes(32, "little") for k in raw)
    cfg.set_window_size(c)
    try:
        out = cfg.msm(sb, pb, len(pts), scalar_layout=msm_pkg.SCALAR_CANON_LE)
    finally:
        cfg.set_window_size(0)
    assert o.decode_jacobian_mont_le(out) == o.msm_naive([k % o.R_ORDER for k in raw], pts)


def test_ragged_host_batch_more_instances_than_tickets(cfg, msm_pkg):
    """msm_amd_msm_batch with seven instances of very different sizes: more instances than batch tickets (4) and than
    instances in flight (3), two staging sets sized by the largest instance, every result against the oracle."""
    from oracle import c_oracle as co
    sizes = [1000, 17, 4096, 3, 70000, 256, 1]
    data = [co.gen_instance(o.SEED_BASE + 600 + j, n) for j, n in enumerate(sizes)]
    pts, scs = [d[0] for d in data], [d[1] for d in data]
    outs = cfg.msm_batch(scs, pts, sizes)
    assert len(outs) == len(sizes)
    for j, n in enumerate(sizes):
        assert o.decode_jacobian_mont_le(outs[j]) == o.decode_jacobian_mont_le(co.msm_best(scs[j], pts[j], n, 2)), j
    # the same instances one by one give the same bytes
    assert [msm_pkg.gpu_msm_h2c(scs[j], pts[j], cfg) for j in range(len(sizes))] == outs


def test_lone_single_stream_call_mixed_with_pipelined_batches(cfg, msm_pkg):
    """A single instance submitted while nothing is in flight runs on ONE stream (no cross-stream hand-offs); batches
    submitted right behind it use the stream split and share the workspaces' events with it.  Same bytes as blocking
    calls, in any interleaving."""
    n = 5000
    insts = [cfg.generate_instance(o.SEED_BASE + 700 + j, n, True) for j in range(5)]
    dp = [i[0] for i in insts]
    ds = [i[1] for i in insts]
    try:
        want = [cfg.msm_batch_device([ds[j]], [dp[j]], [n])[0] for j in range(5)]     # five lone calls
        for _ in range(3):
            h0 = cfg.submit_batch_device([ds[0]], [dp[0]], [n])                      # lone: single stream
            h1 = cfg.submit_batch_device(ds[1:4], dp[1:4], [n] * 3)                  # behind it: pipelined
            h2 = cfg.submit_batch_device([ds[4]], [dp[4]], [n])                      # one instance, but NOT lone
            assert cfg.wait_batch(h1) == want[1:4]
            assert cfg.wait_batch(h0) == [want[0]]
            assert cfg.wait_batch(h2) == [want[4]]
            assert cfg.msm_batch_device([ds[2]], [dp[2]], [n]) == [want[2]]         # lone again
    finally:
        for p in dp + ds:
            cfg.free(p)
