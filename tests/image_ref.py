"""Independent restatement (test infrastructure) of how the reference loads a texture (src/image.cpp:20-39):
stbi_set_flip_vertically_on_load(true); stbi_ldr_to_hdr_gamma(1); stbi_loadf(path, &w, &h, &ch, 3).
PNG/LDR files are decoded with PIL, Radiance .hdr files with the small RGBE reader below.
Returns float32 [h, w, 3] with row 0 = BOTTOM row of the picture."""
import numpy as np


def _read_hdr(path):
    d = open(path, "rb").read()
    pos = 0

    def line():
        nonlocal pos
        e = d.index(b"\n", pos); s = d[pos:e]; pos = e + 1; return s
    head = line()
    assert head.startswith(b"#?RADIANCE") or head.startswith(b"#?RGBE"), "not a Radiance file"
    fmt = False
    while True:
        l = line()
        if not l:
            break
        if l == b"FORMAT=32-bit_rle_rgbe":
            fmt = True
    assert fmt
    dims = line().split()
    assert dims[0] == b"-Y" and dims[2] == b"+X"
    H, W = int(dims[1]), int(dims[3])
    out = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        if 8 <= W < 32768 and d[pos] == 2 and d[pos + 1] == 2 and not (d[pos + 2] & 0x80):
            assert (d[pos + 2] << 8 | d[pos + 3]) == W
            pos += 4
            for k in range(4):
                x = 0
                while x < W:
                    c = d[pos]; pos += 1
                    if c > 128:
                        c -= 128; out[y, x:x + c, k] = d[pos]; pos += 1
                    else:
                        out[y, x:x + c, k] = np.frombuffer(d, np.uint8, c, pos); pos += c
                    x += c
        else:
            out[y] = np.frombuffer(d, np.uint8, 4 * W, pos).reshape(W, 4); pos += 4 * W
    e = out[..., 3].astype(np.int32)
    f1 = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)        # stbi__hdr_convert: ldexp(1.0f, e - (128 + 8))
    rgb = (out[..., :3].astype(np.float32) * f1[..., None]).astype(np.float32)
    rgb[e == 0] = 0
    return rgb


def load_texture(path):
    raw = open(path, "rb").read(2)
    if raw == b"#?":
        top = _read_hdr(path)
    else:
        from PIL import Image
        im = Image.open(path)
        if im.mode in ("I;16", "I;16B", "I", "F"):
            raise ValueError("only 8-bit images are supported")
        if im.mode in ("L", "LA"):
            g = np.asarray(im.convert("L"), np.uint8)
            arr = np.stack([g, g, g], -1)                              # stb: grey replicated
        else:
            arr = np.asarray(im.convert("RGB"), np.uint8)              # palette expanded, alpha dropped
        top = (arr.astype(np.float32) / np.float32(255.0)).astype(np.float32)   # stbi__ldr_to_hdr with gamma 1, scale 1
    return np.ascontiguousarray(top[::-1])                             # flip: row 0 = bottom row
