"""Instance-file harness: the bincode 1.3 cache format of src/utils/preprocess.rs:30-111 and the
ToLimbs / FromLimbs wire conversions (limbs_conversion.rs:87-195, 282-389).

The expected bytes are rebuilt here with `struct` from the oracle's big-int values, independently of the
library's C++ writer.  Parity note: the reference holds no .bin fixture of this format (its cache lives in the
user's home directory), so the layout is pinned by the serde/bincode definition only -- "parity unpinned" at
the fixture level, like the rest of the path.
"""
import os
import struct

import pytest

from oracle import bn254_ref as o
from helpers import h2c_instance_bytes, small_instance


def expected_file_bytes(instances):
    """instances: list of (points as Jacobian int triples or None, canonical scalars)."""
    out = [struct.pack("<Q", len(instances))]
    for pts, scs in instances:
        out.append(struct.pack("<Q", len(pts)))
        for pj in pts:
            out.append(struct.pack("<Q24I", 24, *o.encode_point_be32(pj)))
        out.append(struct.pack("<Q", len(scs)))
        for k in scs:
            out.append(struct.pack("<Q8I", 8, *o.encode_scalar_be32(k)))
    return b"".join(out)


@pytest.fixture(scope="module")
def pre(msm_pkg):
    import importlib
    return importlib.import_module("metal-msm-gpu-acceleration_amd.preprocess")


def wire_instance(pre, msm_pkg, pts, scs):
    sb, pb = h2c_instance_bytes(pts, scs)
    return pre.to_wire(sb, pb, msm_pkg.SCALAR_MONT_LE, msm_pkg.POINT_H2C_AFFINE, 64)


def test_to_wire_matches_oracle_encoding(pre, msm_pkg):
    pts, scs = small_instance(11, 9)
    scs[0], scs[1] = 0, o.R_ORDER - 1
    inst = wire_instance(pre, msm_pkg, pts, scs)
    for i, (p, k) in enumerate(zip(pts, scs)):
        assert inst.points[96 * i:96 * i + 96] == struct.pack("<24I", *o.encode_point_be32(o.to_jac(p)))
        assert inst.scalars[32 * i:32 * i + 32] == struct.pack("<8I", *o.encode_scalar_be32(k))


def test_to_wire_all_layouts_agree(pre, msm_pkg):
    pts, scs = small_instance(12, 5)
    ref = wire_instance(pre, msm_pkg, pts, scs)
    canon_le = b"".join(o.int_to_le_bytes32(k) for k in scs)
    proj = b"".join(o.encode_projective_ark(o.to_jac(p)) for p in pts)
    ark_aff = b"".join(o.encode_affine_h2c(p) + b"\x00" * 8 for p in pts)
    a = pre.to_wire(canon_le, proj, msm_pkg.SCALAR_CANON_LE, msm_pkg.POINT_ARK_PROJECTIVE, 96)
    b = pre.to_wire(ref.scalars, ark_aff, msm_pkg.SCALAR_CANON_BE32, msm_pkg.POINT_ARK_AFFINE, 72)
    c = pre.to_wire(ref.scalars, ref.points, msm_pkg.SCALAR_CANON_BE32, msm_pkg.POINT_JAC_BE32, 96)
    for x in (a, b, c):
        assert x.points == ref.points and x.scalars == ref.scalars


def test_identity_points_get_z_zero(pre, msm_pkg):
    # h2c encodes the affine identity as (0, 0), ark as infinity = true; both must come out with z = 0
    # (the reference's h2c path gives them z = Mont(1): SURVEY Appendix B.1, not replicated)
    one = wire_instance(pre, msm_pkg, [None], [5])
    assert one.points[64:96] == b"\x00" * 32
    ark = pre.to_wire(one.scalars, b"\x00" * 64 + b"\x01" + b"\x00" * 7, msm_pkg.SCALAR_CANON_BE32,
                      msm_pkg.POINT_ARK_AFFINE, 72)
    assert ark.points[64:96] == b"\x00" * 32


def test_from_wire_roundtrip(pre, msm_pkg):
    import ctypes
    pts, scs = small_instance(13, 6)
    sb, pb = h2c_instance_bytes(pts, scs)
    inst = wire_instance(pre, msm_pkg, pts, scs)
    so = ctypes.create_string_buffer(32 * 6)
    po = ctypes.create_string_buffer(96 * 6)
    assert msm_pkg.lib().msm_amd_from_wire(msm_pkg.SCALAR_MONT_LE, msm_pkg.POINT_ARK_PROJECTIVE, inst.scalars,
                                           inst.points, 6, so, po) == 0
    assert so.raw == sb
    assert po.raw == b"".join(o.encode_projective_ark(o.to_jac(p)) for p in pts)
    # affine outputs are refused rather than silently wrong
    assert msm_pkg.lib().msm_amd_from_wire(msm_pkg.SCALAR_MONT_LE, msm_pkg.POINT_H2C_AFFINE, inst.scalars,
                                           inst.points, 6, so, po) == msm_pkg.INPUT_ERROR


def test_save_is_byte_exact_bincode_and_loads_back(pre, msm_pkg, tmp_path):
    data = [small_instance(21, 7), small_instance(22, 3), ([], [])]
    insts = [wire_instance(pre, msm_pkg, p, s) for p, s in data]
    path = tmp_path / "msm_x.bin"
    pre.save_msm_instances(path, insts)
    assert path.read_bytes() == expected_file_bytes([([o.to_jac(q) for q in p], s) for p, s in data])
    back = pre.load_msm_instances(path)
    assert [len(b) for b in back] == [7, 3, 0]
    for a, b in zip(insts, back):
        assert a.points == b.points and a.scalars == b.scalars


def test_load_accepts_an_independently_written_file(pre, tmp_path):
    pts, scs = small_instance(23, 4)
    jac = [o.to_jac(p) for p in pts]
    path = tmp_path / "ext.bin"
    path.write_bytes(expected_file_bytes([(jac, scs)]))
    (inst,) = pre.load_msm_instances(path)
    assert inst.points == b"".join(struct.pack("<24I", *o.encode_point_be32(q)) for q in jac)
    assert inst.scalars == b"".join(struct.pack("<8I", *o.encode_scalar_be32(k)) for k in scs)


def test_errors_mirror_harness_error(pre, msm_pkg, tmp_path):
    with pytest.raises(pre.FileOpenError):
        pre.load_msm_instances(tmp_path / "missing.bin")
    pts, scs = small_instance(24, 4)
    good = expected_file_bytes([([o.to_jac(p) for p in pts], scs)])
    cases = {
        "truncated": good[:-5],
        "empty": b"",
        "huge_count": struct.pack("<Q", 1 << 60) + good[8:],
        "huge_points": good[:8] + struct.pack("<Q", 1 << 59) + good[16:],
        "bad_inner_len": good[:16] + struct.pack("<Q", 23) + good[24:],
    }
    for name, blob in cases.items():
        p = tmp_path / f"{name}.bin"
        p.write_bytes(blob)
        with pytest.raises(pre.DeserializationError):
            pre.load_msm_instances(p)
    # points.len() != scalars.len() is the reference's assert (preprocess.rs:78)
    three_scalars = struct.pack("<Q", 3) + b"".join(struct.pack("<Q8I", 8, *o.encode_scalar_be32(k)) for k in scs[:3])
    pts_part = good[8:8 + 8 + 4 * 104]
    p = tmp_path / "mismatch.bin"
    p.write_bytes(struct.pack("<Q", 1) + pts_part + three_scalars)
    with pytest.raises(pre.InvalidData):
        pre.load_msm_instances(p)
    with pytest.raises(pre.InvalidData):
        pre.MsmInstance(b"\x00" * 96, b"\x00" * 64)


def test_default_path_naming(pre, tmp_path, monkeypatch):
    assert pre.instance_path(20, 5, tmp_path) == os.path.join(str(tmp_path), "msm_20x5.bin")
    monkeypatch.setenv("HOME", "/somewhere")
    assert pre.instance_path(16, 1) == "/somewhere/.msm_gpu_acceleration/msm_vecs/msm_16x1.bin"
    assert pre.default_msm_vec_repo() == "/somewhere/.msm_gpu_acceleration/msm_vecs"


def test_get_or_create_validates_an_existing_file(pre, msm_pkg, tmp_path):
    pts, scs = small_instance(25, 8)
    inst = wire_instance(pre, msm_pkg, pts, scs)
    pre.save_msm_instances(pre.instance_path(3, 2, tmp_path), [inst, inst])
    got = pre.get_or_create_msm_instances(3, 2, seed=1, dir=tmp_path)          # loads, no GPU needed
    assert len(got) == 2 and got[0].points == inst.points
    pre.save_msm_instances(pre.instance_path(4, 2, tmp_path), [inst, inst])     # wrong size for log 4
    with pytest.raises(pre.InvalidData):
        pre.get_or_create_msm_instances(4, 2, seed=1, dir=tmp_path)


@pytest.mark.gpu
def test_generated_file_runs_and_matches_oracle(pre, msm_pkg, cfg, tmp_path):
    seed = 0xB2540000 + 77
    made = pre.get_or_create_msm_instances(10, 2, seed, dir=tmp_path, config=cfg)
    assert os.path.getsize(pre.instance_path(10, 2, tmp_path)) == 8 + 2 * (16 + 1024 * (104 + 40))
    again = pre.get_or_create_msm_instances(10, 2, seed + 999, dir=tmp_path, config=cfg)   # now read from the file
    for j, (a, b) in enumerate(zip(made, again)):
        assert a.points == b.points and a.scalars == b.scalars
        pts, scs = o.gen_instance(seed + j, 1024)
        assert a.scalars == b"".join(struct.pack("<8I", *o.encode_scalar_be32(k)) for k in scs)
        assert o.decode_jacobian_mont_le(pre.run_instance(cfg, a)) == o.msm_pippenger(scs, pts, 8)
