"""Synthetic textured scene for the texture / normal-map / ENV-map tests: small procedural PNGs (PIL), a Radiance .hdr
written here (flat and RLE scanlines), a uv-mapped OBJ, and a scene file in the fork's format that uses them."""
import os

import numpy as np


def write_hdr(path, rgb, rle):
    """rgb: float32 [h, w, 3], top row first.  Standard RGBE encoding; `rle` chooses new-style run-length scanlines."""
    h, w, _ = rgb.shape
    m = rgb.max(-1)
    e = np.zeros((h, w), np.int32); mant = np.zeros((h, w, 3), np.uint8)
    nz = m > 1e-32
    fr, ex = np.frexp(m[nz])
    sc = (fr * 256.0 / m[nz])
    mant[nz] = np.clip(rgb[nz] * sc[:, None], 0, 255).astype(np.uint8)
    e[nz] = ex + 128
    px = np.concatenate([mant, e[..., None].astype(np.uint8)], -1)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            if not rle:
                f.write(px[y].tobytes()); continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for k in range(4):
                row = px[y, :, k]; x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])])); x += run
                    else:
                        lit = 1
                        while x + lit < w and lit < 128 and not (x + lit + 2 < w and row[x + lit] == row[x + lit + 1] == row[x + lit + 2]):
                            lit += 1
                        f.write(bytes([lit]) + row[x:x + lit].tobytes()); x += lit


def make(tmpdir, res=(96, 80), depth=5, env=True, rle=True):
    from PIL import Image
    rng = np.random.default_rng(42)
    d = str(tmpdir)
    yy, xx = np.mgrid[0:45, 0:61]
    albedo = np.stack([(xx * 4) % 256, (yy * 5) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    albedo[::7] = 230
    Image.fromarray(albedo, "RGB").save(os.path.join(d, "albedo.png"))
    rough = (rng.random((33, 29)) * 200 + 20).astype(np.uint8)
    Image.fromarray(rough, "L").save(os.path.join(d, "rough.png"))
    metal = np.zeros((16, 16, 4), np.uint8); metal[..., 0] = np.linspace(0, 255, 16).astype(np.uint8)[None, :]; metal[..., 3] = 128
    Image.fromarray(metal, "RGBA").save(os.path.join(d, "metal.png"))
    ny, nx = np.mgrid[0:40, 0:40]
    nrm = np.stack([128 + 60 * np.sin(nx * 0.5), 128 + 60 * np.cos(ny * 0.4), 220 + 0 * nx], -1).clip(0, 255).astype(np.uint8)
    Image.fromarray(nrm, "RGB").save(os.path.join(d, "normal.png"))
    pal = Image.fromarray((rng.random((12, 20)) * 6).astype(np.uint8), "P")
    pal.putpalette([30, 60, 200, 220, 180, 40, 90, 200, 90, 250, 250, 250, 10, 10, 10, 160, 60, 160] + [0] * (768 - 18))
    pal.save(os.path.join(d, "pal.png"))
    ey, ex = np.mgrid[0:32, 0:64]
    envimg = np.stack([0.2 + 0.8 * (ex / 63.0), 0.3 + 0.5 * (ey / 31.0), 0.9 - 0.6 * (ex / 63.0)], -1).astype(np.float32)
    envimg[4:8, 10:16] = 40.0                                  # a "sun"
    envimg[20:, :8] = 0.0
    write_hdr(os.path.join(d, "env.hdr"), envimg, rle)
    open(os.path.join(d, "quad.obj"), "w").write(
        "v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 1 0\nf 1/1/1 3/3/1 2/2/1\nf 1/1/1 4/4/1 3/3/1\n")
    open(os.path.join(d, "box.obj"), "w").write(
        "v -0.5 -0.5 0.5\nv 0.5 -0.5 0.5\nv -0.5 0.5 0.5\nv 0.5 0.5 0.5\nv -0.5 0.5 -0.5\nv 0.5 0.5 -0.5\nv -0.5 -0.5 -0.5\nv 0.5 -0.5 -0.5\n"
        "vt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\nvn 0 0 1\nvn 0 1 0\nvn 0 0 -1\nvn 0 -1 0\nvn 1 0 0\nvn -1 0 0\n"
        "f 1/1/1 2/2/1 3/3/1\nf 3/3/1 2/2/1 4/4/1\nf 3/1/2 4/2/2 5/3/2\nf 5/3/2 4/2/2 6/4/2\nf 5/4/3 6/3/3 7/2/3\nf 7/2/3 6/3/3 8/1/3\n"
        "f 7/1/4 8/2/4 1/3/4\nf 1/3/4 8/2/4 2/4/4\nf 2/1/5 8/2/5 4/3/5\nf 4/3/5 8/2/5 6/4/5\nf 7/1/6 1/2/6 5/3/6\nf 5/3/6 1/2/6 3/4/6\n")
    scene = f"""MATERIAL light
TYPE Light
ALBEDO 6 6 6

MATERIAL floor
TYPE MetallicWorkflow
ALBEDO albedo.png
METALLIC metal.png
ROUGHNESS rough.png
NORMAL normal.png
IOR 0

MATERIAL boxmat
TYPE Lambertian
ALBEDO pal.png
NORMAL normal.png

MATERIAL glossy
TYPE Microfacet
ALBEDO albedo.png
ROUGHNESS 0.2

MATERIAL glass
TYPE Dielectric
ALBEDO .95 .95 .95
IOR 1.5

CAMERA
RES {res[0]} {res[1]}
FOVY 40
ITERATIONS 10
DEPTH {depth}
FILE tex
EYE 0 2.2 5
LOOKAT 0 0.8 0
UP 0 1 0

OBJECT lamp
sphere
material light
TRANS 0 4 0
ROTAT 0 0 0
SCALE 1 1 1

OBJECT ground
quad.obj
material floor
TRANS 0 0 0
ROTAT 0 20 0
SCALE 3 1 3

OBJECT crate
box.obj
material boxmat
TRANS -1 0.6 0
ROTAT 0 30 10
SCALE 1.2 1.2 1.2

OBJECT ball
sphere
material glossy
TRANS 1.2 0.7 0.5
ROTAT 0 0 0
SCALE 1.4 1.4 1.4

OBJECT marble
sphere
material glass
TRANS 0 0.5 1.8
ROTAT 0 0 0
SCALE 1 1 1
"""
    if env:
        scene += "\nENV env.hdr"                  # last line without a newline, like textureEnv_test.txt
    path = os.path.join(d, "tex_scene.txt")
    open(path, "w").write(scene)
    return path
