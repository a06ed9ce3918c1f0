// image.cpp — host texture images: the mirror of the reference's src/image.h:8-51 / src/image.cpp:20-39
// (class image: float RGB pixels loaded with stbi_loadf(path, &w, &h, &ch, 3) after stbi_set_flip_vertically_on_load(true)
// and stbi_ldr_to_hdr_gamma(gamma)), with own decoders instead of the vendored stb_image:
//   * PNG  : 8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced (zlib inflate + the five PNG filters);
//   * .hdr : Radiance RGBE, flat or new-style RLE scanlines, "-Y H +X W" orientation.
// Conversion to float follows stb_image v2.06: LDR byte b -> (float)(pow(b / 255.0f, gamma) * 1.0f) (alpha dropped, grey
// replicated), RGBE -> mantissa * ldexp(1, e - 136) (all zero when e == 0); rows flipped so that row 0 is the bottom row.
#include "image.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <zlib.h>

namespace {

bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.good()) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n) f.read(reinterpret_cast<char*>(out.data()), n);
    return f.good() || f.eof();
}

unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// -> 8-bit RGB, top row first
bool decode_png(const std::vector<unsigned char>& d, int& W, int& H, std::vector<unsigned char>& rgb, std::string& err) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    W = H = 0;
    while (pos + 12 <= d.size()) {
        const unsigned len = be32(&d[pos]);
        const char* tag = reinterpret_cast<const char*>(&d[pos + 4]);
        if (pos + 12 + (size_t)len > d.size()) { err = "truncated PNG chunk"; return false; }
        const unsigned char* body = &d[pos + 8];
        if (!std::strncmp(tag, "IHDR", 4)) { W = (int)be32(body); H = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!std::strncmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::strncmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::strncmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (W <= 0 || H <= 0) { err = "PNG without IHDR"; return false; }
    if (W > 65536 || H > 65536) { err = "PNG dimensions out of range"; return false; }
    if (depth != 8 || interlace != 0) { err = "only 8-bit non-interlaced PNG is supported"; return false; }
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "bad PNG colour type"; return false; }
    const size_t stride = (size_t)W * ch;
    // deflate expands by at most ~1032:1: a header that promises more than the IDAT bytes can hold is corrupt (and must not
    // drive a multi-gigabyte allocation)
    if ((stride + 1) * (size_t)H > idat.size() * 1032 + 4096) { err = "PNG image data shorter than its header"; return false; }
    std::vector<unsigned char> raw((stride + 1) * (size_t)H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { err = "PNG inflate failed"; return false; }
    std::vector<unsigned char> img(stride * (size_t)H);
    for (int y = 0; y < H; ++y) {
        const unsigned char* in = &raw[(stride + 1) * (size_t)y];
        const int ft = in[0];
        unsigned char* cur = &img[stride * (size_t)y];
        const unsigned char* up = y ? &img[stride * (size_t)(y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)ch ? cur[x - ch] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)ch) ? up[x - ch] : 0;
            int v = in[1 + x];
            switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break; case 4: v += paeth(a, b, c); break;
                          default: err = "bad PNG filter"; return false; }
            cur[x] = (unsigned char)v;
        }
    }
    rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const unsigned char* p = &img[i * ch];
        unsigned char* o = &rgb[i * 3];
        if (ctype == 0 || ctype == 4) { o[0] = o[1] = o[2] = p[0]; }
        else if (ctype == 3) { const size_t k = (size_t)p[0] * 3; if (k + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; } o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; }
        else { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
    }
    return true;
}

// -> float RGB, top row first
bool decode_hdr(const std::vector<unsigned char>& d, int& W, int& H, std::vector<float>& rgb, std::string& err) {
    size_t pos = 0;
    auto line = [&](std::string& out) { out.clear(); while (pos < d.size() && d[pos] != '\n') out.push_back((char)d[pos++]); if (pos < d.size()) ++pos; return true; };
    std::string l;
    line(l);
    if (l.compare(0, 10, "#?RADIANCE") != 0 && l.compare(0, 6, "#?RGBE") != 0) { err = "not a Radiance .hdr"; return false; }
    bool fmt = false;
    for (;;) { line(l); if (l.empty()) break; if (l == "FORMAT=32-bit_rle_rgbe") fmt = true; if (pos >= d.size()) break; }
    if (!fmt) { err = "unsupported .hdr format"; return false; }
    line(l);
    if (std::sscanf(l.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) { err = "unsupported .hdr orientation"; return false; }
    // every scanline takes at least 4 bytes flat-per-pixel or 4 + 4 x 2 bytes run-length coded
    if (W > 65536 || H > 65536 || (size_t)H * 8 > d.size() - (pos < d.size() ? pos : d.size()) + 8) { err = "truncated .hdr"; return false; }
    std::vector<unsigned char> scan((size_t)W * 4);
    rgb.resize((size_t)W * H * 3);
    for (int y = 0; y < H; ++y) {
        bool rle = false;
        if (W >= 8 && W < 32768 && pos + 4 <= d.size() && d[pos] == 2 && d[pos + 1] == 2 && !(d[pos + 2] & 0x80)) {
            if ((((int)d[pos + 2]) << 8 | d[pos + 3]) != W) { err = "bad .hdr scanline width"; return false; }
            rle = true; pos += 4;
        }
        if (rle) {
            for (int k = 0; k < 4; ++k) {
                int x = 0;
                while (x < W) {
                    if (pos >= d.size()) { err = "truncated .hdr"; return false; }
                    int count = d[pos++];
                    if (count > 128) { count -= 128; if (pos >= d.size() || x + count > W) { err = "bad .hdr run"; return false; } const unsigned char v = d[pos++]; while (count--) scan[(size_t)(x++) * 4 + k] = v; }
                    else { if (count == 0 || pos + count > d.size() || x + count > W) { err = "bad .hdr run"; return false; } while (count--) scan[(size_t)(x++) * 4 + k] = d[pos++]; }
                }
            }
        } else {
            if (pos + (size_t)W * 4 > d.size()) { err = "truncated .hdr"; return false; }
            std::memcpy(scan.data(), &d[pos], (size_t)W * 4); pos += (size_t)W * 4;
        }
        for (int x = 0; x < W; ++x) {
            const unsigned char* p = &scan[(size_t)x * 4];
            float* o = &rgb[((size_t)y * W + x) * 3];
            if (p[3] != 0) { const float f1 = (float)std::ldexp(1.0f, (int)p[3] - (int)(128 + 8)); o[0] = p[0] * f1; o[1] = p[1] * f1; o[2] = p[2] * f1; }
            else o[0] = o[1] = o[2] = 0.f;
        }
    }
    return true;
}

} // namespace

image::image(int x, int y) : width(x), height(y), pixels((size_t)x * y) {}

image::image(const std::string& filePath, float gamma) {
    std::vector<unsigned char> bytes;
    if (!read_file(filePath, bytes)) { error = "cannot open " + filePath; return; }
    std::vector<float> top;                     // float RGB, top row first
    int W = 0, H = 0;
    const bool hdr = bytes.size() >= 2 && bytes[0] == '#' && bytes[1] == '?';
    if (hdr) {
        if (!decode_hdr(bytes, W, H, top, error)) return;
    } else {
        std::vector<unsigned char> rgb;
        if (!decode_png(bytes, W, H, rgb, error)) return;
        top.resize(rgb.size());
        for (size_t i = 0; i < rgb.size(); ++i) top[i] = (float)(std::pow(rgb[i] / 255.0f, gamma) * 1.0f);    // stbi__ldr_to_hdr
    }
    width = W; height = H;
    pixels.resize((size_t)W * H);
    for (int y = 0; y < H; ++y)                  // stbi_set_flip_vertically_on_load(true): row 0 = bottom row
        for (int x = 0; x < W; ++x) {
            const float* s = &top[((size_t)(H - 1 - y) * W + x) * 3];
            pixels[(size_t)y * W + x] = ptk::vec3(s[0], s[1], s[2]);
        }
}
