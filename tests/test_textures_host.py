"""Textures on the host side: the product's own PNG / Radiance decoders against the independent loader in
tests/image_ref.py (PIL + a small RGBE reader), the scene tables of a textured scene against the oracle, and the
"ptk atan2" specification.  No GPU needed."""
import os

import numpy as np
import pytest

import image_ref
import texture_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rle", [True, False])
def test_decoders_match_reference_loader(ptk, tmp_path, rle):
    path = texture_scene.make(tmp_path, rle=rle)
    s = ptk.Scene(path)
    assert s.n_textures == 6 and s.env_map_id == 5
    names = ["albedo.png", "metal.png", "rough.png", "normal.png", "pal.png", "env.hdr"]      # first-use order (scene.cpp:318-337)
    for i, name in enumerate(names):
        want = image_ref.load_texture(os.path.join(str(tmp_path), name))
        got = s.texture(i)
        assert got.shape == want.shape and np.array_equal(got, want), name
    m = s.materials()
    assert list(m["albedoMapID"]) == [-1, 0, 4, 0, -1] and list(m["metallicMapID"]) == [-1, 1, -1, -1, -1]
    assert list(m["roughnessMapID"]) == [-1, 2, -1, -1, -1] and list(m["normalMapID"]) == [-1, 3, 3, -1, -1]


def test_reference_assets_decode(ptk, tmp_path):
    """The reference's own texture assets (real-world PNG filters, RLE .hdr), when the reference tree is present."""
    assets = ["/root/reference/scenes/texture/test/testtexture.png", "/root/reference/scenes/env/Newport_Loft_Ref.hdr"]
    if not all(os.path.exists(a) for a in assets):
        pytest.skip("reference assets not present (GPU box)")
    txt = ("MATERIAL a\nTYPE Lambertian\nALBEDO %s\n\nCAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 1\nLOOKAT 0 0 0\nUP 0 1 0\n\nENV %s\n" % tuple(assets))
    f = tmp_path / "s.txt"; f.write_text(txt)
    s = ptk.Scene(str(f))
    assert s.n_textures == 2 and s.env_map_id == 1
    for i, a in enumerate(assets):
        assert np.array_equal(s.texture(i), image_ref.load_texture(a)), a


def test_textured_scene_tables_match_oracle(ptk, orc, tmp_path):
    path = texture_scene.make(tmp_path)
    s = ptk.Scene(path); o = orc.Scene(path)
    assert (s.n_geoms, s.n_materials, s.n_triangles, s.bvh_size, s.n_lights, s.n_textures, s.env_map_id) == \
           (o.n_geoms, o.n_materials, o.n_triangles, o.bvh_size, o.n_lights, o.n_textures, o.env_map_id)
    for name in ("geoms", "materials", "triangles", "bvh", "lights"):
        a, b = getattr(s, name)(), getattr(o, name)()
        for f in a.dtype.names:
            assert np.array_equal(a[f], b[f], equal_nan=(a[f].dtype.kind == "f")), (name, f)
    tri = s.triangles()
    assert (np.abs(tri["tangent"]).sum(1) > 0).all()            # uv-mapped meshes get tangents (scene.cpp:493-507)


def test_undecodable_and_missing_textures(ptk, tmp_path):
    (tmp_path / "bad.png").write_bytes(b"\xff\xd8\xff\xe0 this is a jpeg, not a png")
    base = "CAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 1\nLOOKAT 0 0 0\nUP 0 1 0\n"
    f = tmp_path / "bad.txt"; f.write_text("MATERIAL a\nTYPE Lambertian\nALBEDO bad.png\n\n" + base)
    with pytest.raises(ptk.PtkError) as e:
        ptk.Scene(str(f))
    assert "decode" in str(e.value)
    # a token that names no file falls back to numbers like the reference (atof of a path is 0)
    g = tmp_path / "missing.txt"; g.write_text("MATERIAL a\nTYPE Lambertian\nALBEDO nosuchfile.png 0.5 0.25\n\n" + base)
    m = ptk.Scene(str(g)).materials()
    assert m["albedoMapID"][0] == -1 and list(m["albedo_s"][0]) == [0.0, 0.5, 0.25]


def test_ptk_atan2_spec(orc):
    rng = np.random.default_rng(3)
    y = np.concatenate([rng.normal(size=200000), [0, 0, 1, -1, 0, 1e-20, -3, 5]]).astype(np.float32)
    x = np.concatenate([rng.normal(size=200000), [1, -1, 0, 0, 0, -1e-20, 0, 5]]).astype(np.float32)
    out = np.zeros_like(x)
    orc.L.orc_ptk_atan2_array(y.ctypes.data, x.ctypes.data, len(x), out.ctypes.data)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(out - ref).max() < 6e-7                       # ~2.5 ulp at pi
    assert out[-8] == 0 and out[-7] == np.float32(np.pi) and out[-6] == np.float32(np.pi / 2) and out[-5] == np.float32(-np.pi / 2) and out[-4] == 0


def test_corrupt_headers_are_rejected_not_allocated(ptk, tmp_path):
    """Found by fuzzing the host loaders under ASan/UBSan: an image header that promises far more pixels than the file holds
    (flipped bytes in IHDR / the .hdr resolution line) and an absurd RES line must be rejected with a message — not drive a
    multi-gigabyte allocation or a minutes-long loop."""
    import struct
    import zlib
    base = "CAMERA\nRES 8 8\nFOVY 45\nITERATIONS 1\nDEPTH 2\nFILE x\nEYE 0 0 1\nLOOKAT 0 0 0\nUP 0 1 0\n"

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + bytes([10 * y + x for x in range(4) for _ in range(3)]) for y in range(4))
    for w, h in ((16, 234881040), (61, 1476395053), (70000, 4)):
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
        (tmp_path / "big.png").write_bytes(png)
        f = tmp_path / "big_png.txt"; f.write_text("MATERIAL a\nTYPE Lambertian\nALBEDO big.png\n\n" + base)
        with pytest.raises(ptk.PtkError) as e:
            ptk.Scene(str(f))
        assert "decode" in str(e.value), (w, h)
    (tmp_path / "big.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 8\n" + bytes(64))
    f = tmp_path / "big_hdr.txt"; f.write_text(base + "\nENV big.hdr\n")
    with pytest.raises(ptk.PtkError) as e:
        ptk.Scene(str(f))
    assert "decode" in str(e.value)
    f = tmp_path / "big_res.txt"; f.write_text(base.replace("RES 8 8", "RES 800000 800000"))
    with pytest.raises(ptk.PtkError) as e:
        ptk.Scene(str(f))
    assert "RES too large" in str(e.value)
    with pytest.raises(ptk.PtkError):                       # the override path goes through the same check
        ptk.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(50000, 50000))
