"""Parity at the sizes BASELINE.json names, through the C ABI on device-resident synthetic inputs.
2^16 / 2^18: bit-exact against the C oracle (restated halo2curves msm_best) on the same inputs.
2^20: size-independent property -- dlog-structured bases P_i = (a0 + i d) G, so the result must equal
(sum k_i (a0 + i d)) G, which needs no MSM code at all."""
import random

import pytest

from oracle import bn254_ref as o
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [16, 18])
def test_device_generated_instance_vs_oracle(cfg, log_n):
    """configs[0]/[1]: 2^16 (the CPU-runnable reference case) and 2^18, h2c layout."""
    n = 1 << log_n
    seed = o.SEED_BASE + log_n
    dp, ds = cfg.generate_instance(seed, n, True)
    try:
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        pb, sb = co.gen_instance(seed, n)                      # same generator on the host
        assert cfg.to_host(dp, 64 * 4096) == pb[:64 * 4096]    # spot check: device generator == oracle generator
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_log20_dlog_identity(cfg, msm_pkg):
    """configs[2] size (2^20): MSM(k, (a0 + i d) G) == (sum k_i (a0 + i d)) G."""
    n = 1 << 20
    rng = random.Random(2020)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    _pb, sb = co.gen_instance(o.SEED_BASE + 77, 1)             # warm the library
    dp, ds = cfg.generate_instance(o.SEED_BASE + 77, n, True)  # scalars from the device generator
    try:
        sb = cfg.to_host(ds, 32 * n)
        pb, expect = co.dlog_instance(a0, d, sb, n)
        cfg.to_device(dp, pb)
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(expect)
        # reference window policy (c = 15, msm.rs:140) and a different window give the same point
        for c in (15, 12, 16, 17):
            cfg.set_window_size(c)
            try:
                assert cfg.msm_batch_device([ds], [dp], [n])[0] == out
            finally:
                cfg.set_window_size(0)
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_skewed_scalars_large(cfg):
    """All scalars equal at 2^16: every window has a single non-empty bucket holding all points -- exercises
    bucket splitting, the workgroup-tree combine and the fine sort's oversized-region path."""
    n = 1 << 16
    pb, _ = co.gen_instance(o.SEED_BASE + 5, n)
    k = 0x2F0D1E2C3B4A59687766554433221100FFEEDDCCBBAA99887766554433221100 % o.R_ORDER
    sb = o.encode_scalar_h2c(k) * n
    out = cfg.msm(sb, pb, n)
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))


def test_digit_carry_chains(cfg):
    """Scalars whose signed-digit recoding carries through every window (r - 1, 2^253 + ..., all-ones windows)."""
    n = 2048
    pb, _ = co.gen_instance(o.SEED_BASE + 6, n)
    specials = [o.R_ORDER - 1, o.R_ORDER - 2, (1 << 253) - 1, (1 << 253), int("1" * 253, 2), int("10" * 126, 2),
                (1 << 15) - 1, 1 << 14, (1 << 14) + 1, 0x7FFF7FFF7FFF7FFF7FFF7FFF]
    rng = random.Random(4)
    sc = [specials[i % len(specials)] if i % 3 == 0 else rng.randrange(o.R_ORDER) for i in range(n)]
    sb = b"".join(o.encode_scalar_h2c(k) for k in sc)
    for c in (0, 15, 7, 3):
        cfg.set_window_size(c)
        try:
            out = cfg.msm(sb, pb, n)
        finally:
            cfg.set_window_size(0)
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n, 2))


def test_all_bases_equal_and_small_scalars(cfg):
    """SURVEY §8(d) secondary distributions: every base the same point (each bucket is a chain of P + P doublings
    and equal-point additions) and scalars below 2^16 (only the two lowest windows are populated)."""
    n = 1 << 14
    pb1, sb = co.gen_instance(o.SEED_BASE + 8, n)
    same = pb1[:64] * n
    out = cfg.msm(sb, same, n)
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, same, n))
    rng = random.Random(16)
    small = b"".join(o.encode_scalar_h2c(rng.randrange(1 << 16)) for _ in range(n))
    out = cfg.msm(small, pb1, n)
    assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(small, pb1, n))


def test_log20_linearity(cfg):
    """Size-independent property at the headline size: MSM(k1 + k2, P) == MSM(k1, P) + MSM(k2, P), with the
    final addition done by the oracle's group law."""
    n = 1 << 20
    dp, ds1 = cfg.generate_instance(o.SEED_BASE + 31, n, False)      # canonical little-endian scalars
    dq, ds2 = cfg.generate_instance(o.SEED_BASE + 32, n, False)
    try:
        import numpy as np
        k1 = np.frombuffer(cfg.to_host(ds1, 32 * n), dtype="<u8").reshape(n, 4)
        k2 = np.frombuffer(cfg.to_host(ds2, 32 * n), dtype="<u8").reshape(n, 4)
        r = o.R_ORDER
        s = bytearray(32 * n)
        for i in range(n):   # 256-bit modular addition on the host (Python integers)
            a = int.from_bytes(k1[i].tobytes(), "little") + int.from_bytes(k2[i].tobytes(), "little")
            s[32 * i:32 * i + 32] = (a - r if a >= r else a).to_bytes(32, "little")
        d_sum = cfg.alloc(32 * n)
        cfg.to_device(d_sum, bytes(s))
        outs = cfg.msm_batch_device([ds1, ds2, d_sum], [dp, dp, dp], [n, n, n], scalar_layout=1)   # SCALAR_CANON_LE
        cfg.free(d_sum)
        a, b, c = (o.decode_jacobian_mont_le(x) for x in outs)
        assert o.aff_add(a, b) == c
    finally:
        for d in (dp, ds1, dq, ds2):
            cfg.free(d)


def test_point_range_split_on_one_gpu(cfg, msm_pkg):
    """The single-instance split of SURVEY §8e, all ranges on this GPU: partial MSMs over 3 point ranges of a
    2^16 instance, added by msm_amd_sum_points, equal the whole MSM."""
    import importlib
    mg = importlib.import_module("metal-msm-gpu-acceleration_amd.multi_gpu")
    n = 1 << 16
    dp, ds = cfg.generate_instance(o.SEED_BASE + 41, n, True)
    try:
        whole = cfg.msm_batch_device([ds], [dp], [n])[0]
        world = 3
        local = lambda b, e: cfg.msm_batch_device([ds + 32 * b], [dp + 64 * b], [e - b])[0]
        parts = [local(*mg.point_range(r, world, n)) for r in range(world)]
        assert msm_pkg.sum_points(parts) == whole
        assert mg.sharded_msm(local, msm_pkg.sum_points, 0, 1, n) == whole
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_log24_ark_projective_dlog_identity(cfg, msm_pkg):
    """configs[4] (2^24 points, arkworks G1Projective layout, z = one): the dlog identity
    MSM(k, (a0 + i d) G) == (sum k_i (a0 + i d)) G needs no MSM code on the checking side, so it scales to the
    full size.  ~1.5 GiB of bases are built on the host from the oracle's dlog generator."""
    import numpy as np
    n = 1 << 24
    rng = random.Random(2024)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    dp, ds = cfg.generate_instance(o.SEED_BASE + 240, n, True)        # scalars from the device generator
    cfg.free(dp)
    d_proj = None
    try:
        sb = cfg.to_host(ds, 32 * n)
        pb, expect = co.dlog_instance(a0, d, sb, n)                    # affine, 64 B each
        one = np.frombuffer(o.int_to_le_bytes32(o.MONT_R % o.P), dtype=np.uint8)
        proj = np.empty((n, 96), dtype=np.uint8)
        proj[:, :64] = np.frombuffer(pb, dtype=np.uint8).reshape(n, 64)
        proj[:, 64:] = one
        del pb
        d_proj = cfg.alloc(96 * n)
        cfg.to_device(d_proj, proj.tobytes())
        del proj
        out = cfg.msm_batch_device([ds], [d_proj], [n], point_layout=msm_pkg.POINT_ARK_PROJECTIVE)[0]
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(expect)
    finally:
        cfg.free(ds)
        if d_proj is not None:
            cfg.free(d_proj)


def test_log24_ark_affine_dlog_identity(cfg, msm_pkg):
    """BASELINE configs[4], the OTHER arkworks variant SURVEY 8(d) C5 names: 2^24 points as ark_bn254::G1Affine records
    (72 bytes: x, y, infinity flag + padding; limbs_conversion.rs:132-137 goes through into_group(), which honours the
    flag), a few of them flagged infinity = 1 with garbage coordinates.  Checked by the dlog identity with the flagged
    points' terms left out."""
    import numpy as np
    n = 1 << 24
    rng = random.Random(2424)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    dp, ds = cfg.generate_instance(o.SEED_BASE + 241, n, True)
    cfg.free(dp)
    d_aff = None
    try:
        sb = cfg.to_host(ds, 32 * n)
        pb, _ = co.dlog_instance(a0, d, sb, n)
        rec = np.zeros((n, 72), dtype=np.uint8)
        rec[:, :64] = np.frombuffer(pb, dtype=np.uint8).reshape(n, 64)
        del pb
        inf = sorted({0, 1, 4097, n // 3, n - 1} | {rng.randrange(n) for _ in range(11)})
        for i in inf:
            rec[i, 64] = 1                                        # infinity = true; x, y keep a (now meaningless) point
            rec[i, 65:72] = 0xAB                                  # padding bytes are not read
        rec[inf[2], :64] = 0xFF                                   # ... and garbage coordinates under the flag
        d_aff = cfg.alloc(72 * n)
        cfg.to_device(d_aff, rec.tobytes())
        del rec
        out = cfg.msm_batch_device([ds], [d_aff], [n], point_layout=msm_pkg.POINT_ARK_AFFINE)[0]
        # expected: the dlog sum without the flagged terms
        r_inv = pow(o.MONT_R, -1, o.R_ORDER)
        q = np.frombuffer(sb, dtype="<u2").reshape(n, 16).astype(np.uint64)
        keep = np.ones(n, dtype=bool)
        keep[inf] = False
        idx = np.arange(n, dtype=np.uint64)
        s0 = [int(x) for x in q[keep].sum(axis=0)]
        s1 = [int(x) for x in (q[keep] * idx[keep, None]).sum(axis=0)]
        sum_k = sum(v << (16 * j) for j, v in enumerate(s0))
        sum_ik = sum(v << (16 * j) for j, v in enumerate(s1))
        expect = o.scalar_mul((a0 * sum_k + d * sum_ik) * r_inv % o.R_ORDER, o.GEN)
        assert o.decode_jacobian_mont_le(out) == expect
    finally:
        cfg.free(ds)
        if d_aff is not None:
            cfg.free(d_aff)


def test_log26_resident_dlog_identity(cfg, msm_pkg):
    """Four times configs[4]'s size: 2^26 points (4 GiB of bases, 2 GiB of scalars, device-resident), the call runs as
    8 pipelined point ranges of 2^23.  Checked by the dlog identity; 2^28 points (the largest power of two below the
    API's 2^31 - 1 bound whose buffers the tool builds, 349 ms) are exercised by tools/big_msm.py, profiles/r03_big_sizes.txt."""
    n = 1 << 26
    rng = random.Random(2026)
    a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
    dp, ds = cfg.generate_instance(o.SEED_BASE + 260, n, True)
    cfg.free(dp)
    dpts = None
    try:
        sb = cfg.to_host(ds, 32 * n)
        pb, expect = co.dlog_instance(a0, d, sb, n)
        del sb
        dpts = cfg.alloc(64 * n)
        cfg.to_device(dpts, pb)
        del pb
        out = cfg.msm_batch_device([ds], [dpts], [n])[0]
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(expect)
    finally:
        cfg.free(ds)
        if dpts is not None:
            cfg.free(dpts)


def test_headline_shape_two_batches_in_flight_dlog(cfg, msm_pkg):
    """The bench's shape: 2 batches x 5 instances of 2^20 points in flight through submit_batch_device / wait_batch
    (4 workspaces, 4 streams, both reduce streams busy).  Bases are dlog-structured, P_i = (a0 + i d) G, so every one
    of the 10 results is checked against (sum k_i (a0 + i d)) G computed with big integers -- no MSM code on the
    checking side."""
    n, inst = 1 << 20, 5
    rng = random.Random(31337)
    r_inv = pow(o.MONT_R, -1, o.R_ORDER)
    d_pts, params = [], []
    d_sc = [[], []]
    try:
        for j in range(inst):
            a0, d = rng.randrange(o.R_ORDER), rng.randrange(o.R_ORDER)
            dp, ds = cfg.generate_instance(o.SEED_BASE + 900 + j, n, True)
            pb, _ = co.dlog_instance(a0, d, cfg.to_host(ds, 32 * n), n)
            cfg.to_device(dp, pb)
            d_pts.append(dp)
            d_sc[0].append(ds)
            dp2, ds2 = cfg.generate_instance(o.SEED_BASE + 950 + j, n, True)   # second batch: other scalars
            cfg.free(dp2)
            d_sc[1].append(ds2)
            params.append((a0, d))
        ns = [n] * inst
        h0 = cfg.submit_batch_device(d_sc[0], d_pts, ns)
        h1 = cfg.submit_batch_device(d_sc[1], d_pts, ns)          # both batches in flight before any wait
        outs = [cfg.wait_batch(h0), cfg.wait_batch(h1)]
        for b in range(2):
            for j in range(inst):
                a0, d = params[j]
                raw = cfg.to_host(d_sc[b][j], 32 * n)
                s, base = 0, a0
                for i in range(n):
                    s += int.from_bytes(raw[32 * i:32 * i + 32], "little") * base
                    base += d
                s = s * r_inv % o.R_ORDER                          # scalars are Montgomery residues k R mod r
                assert o.decode_jacobian_mont_le(outs[b][j]) == o.scalar_mul(s, o.GEN), (b, j)
    finally:
        for p in d_pts + d_sc[0] + d_sc[1]:
            cfg.free(p)


def test_registered_host_buffers_and_resident_bases_batch(cfg, msm_pkg):
    """msm_amd_host_register (DMA uploads) and msm_amd_msm_batch with MSM_AMD_POINT_PREPARED (bases resident, only
    scalars cross PCIe): same results as the plain host-buffer batch and as the CPU oracle."""
    n, inst = 1 << 16, 3
    data = [co.gen_instance(o.SEED_BASE + 40 + j, n) for j in range(inst)]
    pts, scs = [d[0] for d in data], [d[1] for d in data]
    ns = [n] * inst
    plain = cfg.msm_batch(scs, pts, ns)
    for b in pts + scs:
        cfg.host_register(b)
    try:
        with pytest.raises(msm_pkg.MsmError):
            cfg.host_register(pts[0])                              # registering twice is refused
        assert cfg.msm_batch(scs, pts, ns) == plain
        prepared = [cfg.bases_upload(p, n) for p in pts]
        try:
            assert cfg.msm_batch(scs, prepared, ns, point_layout=msm_pkg.POINT_PREPARED) == plain
        finally:
            for d in prepared:
                cfg.free(d)
    finally:
        for b in pts + scs:
            cfg.host_unregister(b)
    with pytest.raises(msm_pkg.MsmError):
        cfg.host_unregister(pts[0])
    assert o.decode_jacobian_mont_le(plain[0]) == o.decode_jacobian_mont_le(co.msm_best(scs[0], pts[0], n))


@pytest.mark.parametrize("env", [
    {"MSM_AMD_HB": "3", "MSM_AMD_MB": "3"},                          # three-level sort forced at a small size
    {"MSM_AMD_HB": "4", "MSM_AMD_MB": "4", "MSM_AMD_TILED": "1"},    # + tile-staged scatter
    {"MSM_AMD_TILED": "1", "MSM_AMD_TILE_THREADS": "512"},
    {"MSM_AMD_BALLOT": "0"}, {"MSM_AMD_BALLOT": "3"},                # ranking: LDS atomics only / wave multisplit everywhere
    {"MSM_AMD_CH": "16"},                                            # every bucket above 16 points is split and combined
])
def test_sort_and_plan_variants_give_identical_results(cfg, env):
    """The plan knobs are read per call: every sort / ranking / work-item variant must return the same bytes as the
    default plan, on uniform scalars and on a skewed instance (all scalars equal: one bucket per window holds all
    points, one sort region holds the whole window)."""
    import os
    n = 1 << 17
    dp, ds = cfg.generate_instance(o.SEED_BASE + 1234, n, True)
    k = 0x1F0D1E2C3B4A59687766554433221100FFEEDDCCBBAA99887766554433221100 % o.R_ORDER
    d_eq = cfg.alloc(32 * n)
    cfg.to_device(d_eq, o.encode_scalar_h2c(k) * n)
    try:
        want = cfg.msm_batch_device([ds, d_eq], [dp, dp], [n, n])
        old = {key: os.environ.get(key) for key in env}
        os.environ.update(env)
        try:
            got = cfg.msm_batch_device([ds, d_eq], [dp, dp], [n, n])
        finally:
            for key, v in old.items():
                if v is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = v
        assert got == want
    finally:
        cfg.free(dp)
        cfg.free(ds)
        cfg.free(d_eq)


def test_log22_three_level_sort_uniform_and_skewed(cfg):
    """2^22 points take the three-level sort and the tile-staged scatter by default: uniform scalars against the CPU
    oracle, and all-equal scalars (k * sum of the points) where one coarse region, one middle region and one bucket
    per window hold everything."""
    n = 1 << 22
    dp, ds = cfg.generate_instance(o.SEED_BASE + 2222, n, True)
    try:
        pb, sb = cfg.to_host(dp, 64 * n), cfg.to_host(ds, 32 * n)
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))
        k = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % o.R_ORDER
        eq = o.encode_scalar_h2c(k) * n
        cfg.to_device(ds, eq)
        out = cfg.msm_batch_device([ds], [dp], [n])[0]
        assert o.decode_jacobian_mont_le(out) == o.decode_jacobian_mont_le(co.msm_best(eq, pb, n))
    finally:
        cfg.free(dp)
        cfg.free(ds)


@pytest.mark.parametrize("log_n", [10, 14, 16, 18])
def test_lone_call_reduce_groups_agree_with_pipelined_geometry(cfg, log_n):
    """A lone call picks the group size of the row / column sums of the window reduction level by level (short chains,
    more launches); pipelined instances use groups of 16.  Every choice must give the same point: lone calls with the
    minimum group pinned to 4, 8 and 16 (MSM_AMD_REDUCE_GROUP), the default, and the same instance inside a batch of
    three (pipelined geometry); uniform scalars and all-equal scalars (one bucket per window holds everything)."""
    import os
    n = 1 << log_n
    dp, ds = cfg.generate_instance(o.SEED_BASE + 4400 + log_n, n, True)
    k = 0x2A2B2C2D2E2F30313233343536373839404142434445464748495051525354 % o.R_ORDER
    d_eq = cfg.alloc(32 * n)
    cfg.to_device(d_eq, o.encode_scalar_h2c(k) * n)
    try:
        batch = cfg.msm_batch_device([ds, d_eq, ds], [dp, dp, dp], [n, n, n])
        assert batch[0] == batch[2]
        if log_n <= 14:
            pb, sb = cfg.to_host(dp, 64 * n), cfg.to_host(ds, 32 * n)
            assert o.decode_jacobian_mont_le(batch[0]) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))
        old = os.environ.get("MSM_AMD_REDUCE_GROUP")
        try:
            for g in (None, "4", "8", "16"):
                if g is None:
                    os.environ.pop("MSM_AMD_REDUCE_GROUP", None)
                else:
                    os.environ["MSM_AMD_REDUCE_GROUP"] = g
                assert cfg.msm_batch_device([ds], [dp], [n])[0] == batch[0], g
                assert cfg.msm_batch_device([d_eq], [dp], [n])[0] == batch[1], g
        finally:
            if old is None:
                os.environ.pop("MSM_AMD_REDUCE_GROUP", None)
            else:
                os.environ["MSM_AMD_REDUCE_GROUP"] = old
    finally:
        cfg.free(dp)
        cfg.free(ds)
        cfg.free(d_eq)


def _with_env(key, value, fn):
    import os
    old = os.environ.get(key)
    if value is None:
        os.environ.pop(key, None)
    else:
        os.environ[key] = value
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = old


@pytest.mark.parametrize("n", [1 << 15, (1 << 15) + 4099, 40003])
def test_lone_call_split_into_pipelined_point_ranges(cfg, msm_pkg, n):
    """A lone call of many points runs as a pipelined batch of point ranges whose results are added on the host
    (run_split; automatic from 2^23 device-resident / 2^19 host points).  Forced here at oracle-checkable sizes with
    MSM_AMD_SPLIT: every entry point that takes one instance -- device-resident, host buffers (h2c), msm_best, prepared
    bases -- must return the unsplit result, which must be the oracle's; n not divisible by the part count, identity
    points and zero scalars inside the ranges, all-equal scalars."""
    rng = random.Random(n)
    dp, ds = cfg.generate_instance(o.SEED_BASE + 5500 + (n & 0xFFF), n, True)
    try:
        pb, sb = bytearray(cfg.to_host(dp, 64 * n)), bytearray(cfg.to_host(ds, 32 * n))
        for i in rng.sample(range(n), 20):
            pb[64 * i:64 * i + 64] = bytes(64)                  # identity points
        for i in rng.sample(range(n), 20):
            sb[32 * i:32 * i + 32] = bytes(32)                  # zero scalars
        pb, sb = bytes(pb), bytes(sb)
        cfg.to_device(dp, pb)
        cfg.to_device(ds, sb)
        want = _with_env("MSM_AMD_SPLIT", "1", lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
        assert o.decode_jacobian_mont_le(want) == o.decode_jacobian_mont_le(co.msm_best(sb, pb, n))
        d_prep = cfg.bases_upload(pb, n)
        try:
            for parts in ("2", "3", "5", "8"):
                got = _with_env("MSM_AMD_SPLIT", parts, lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
                assert got == want, ("device", parts)
                assert cfg.timings().reserved == min(int(parts), 8), parts
                got = _with_env("MSM_AMD_SPLIT", parts, lambda: msm_pkg.gpu_msm_h2c(sb, pb, cfg))
                assert got == want, ("host", parts)
                got = _with_env("MSM_AMD_SPLIT", parts, lambda: msm_pkg.msm_best(sb, pb, cfg))
                assert got == want, ("msm_best", parts)
                got = _with_env("MSM_AMD_SPLIT", parts, lambda: cfg.msm_prepared(sb, d_prep, n))
                assert got == want, ("prepared", parts)
        finally:
            cfg.free(d_prep)
        k = 0x1111222233334444555566667777888899990000AAAABBBBCCCCDDDDEEEEFFFF % o.R_ORDER
        eq = o.encode_scalar_h2c(k) * n
        cfg.to_device(ds, eq)
        one = _with_env("MSM_AMD_SPLIT", "1", lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
        four = _with_env("MSM_AMD_SPLIT", "4", lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
        assert one == four
    finally:
        cfg.free(dp)
        cfg.free(ds)


def test_lone_call_split_automatic_at_2p23(cfg):
    """2^23 device-resident points split into four pipelined ranges by default; same point as the unsplit call."""
    n = 1 << 23
    dp, ds = cfg.generate_instance(o.SEED_BASE + 2323, n, True)
    try:
        got = cfg.msm_batch_device([ds], [dp], [n])[0]
        assert cfg.timings().reserved == 4
        want = _with_env("MSM_AMD_SPLIT", "1", lambda: cfg.msm_batch_device([ds], [dp], [n])[0])
        assert cfg.timings().reserved == 1
        assert got == want
    finally:
        cfg.free(dp)
        cfg.free(ds)
