// Radiance RGBE (.hdr) encoder producing the byte stream the reference's
// stbi_write_hdr("file.hdr", nx, ny, 4, fb) call (main.cu:856) produces:
// text header, "-Y h +X w", one record per scanline top to bottom.  For
// 8 <= w < 32768 a scanline is "02 02 hi lo" followed by the four byte planes
// (R, G, B, E), each run-length coded: a run is >= 3 equal bytes, emitted as
// (128+n, value) with n <= 127; everything before the next run is emitted as
// literal packets (n, bytes...) with n <= 128.  Narrower or wider images are
// written as flat RGBE quadruples.  Colour to RGBE truncates the mantissa.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/crt_hip.h"

namespace {

inline void to_rgbe(const float rgb[3], unsigned char out[4]) {
    float m = rgb[0] > rgb[1] ? rgb[0] : rgb[1];
    if (rgb[2] > m) m = rgb[2];
    if (m < 1e-32f) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    int e = 0;
    float scale = (float)frexp(m, &e) * 256.0f / m;
    for (int c = 0; c < 3; ++c) {
        int q = (int)(rgb[c] * scale); // truncation, as the 8-bit cast of the reference's writer
        out[c] = (unsigned char)q;
    }
    out[3] = (unsigned char)(e + 128);
}

// Run-length codes one byte plane of a scanline.
void rle_plane(const unsigned char* p, int w, std::vector<unsigned char>& out) {
    int x = 0;
    while (x < w) {
        // start of the next run of at least three equal bytes (w if none)
        int r = x;
        while (r + 2 < w && !(p[r] == p[r + 1] && p[r] == p[r + 2])) ++r;
        const bool run = r + 2 < w;
        if (!run) r = w;
        while (x < r) { // literals
            int n = r - x > 128 ? 128 : r - x;
            out.push_back((unsigned char)n);
            out.insert(out.end(), p + x, p + x + n);
            x += n;
        }
        if (run) {
            while (r < w && p[r] == p[x]) ++r;
            while (x < r) {
                int n = r - x > 127 ? 127 : r - x;
                out.push_back((unsigned char)(128 + n));
                out.push_back(p[x]);
                x += n;
            }
        }
    }
}

bool encode(int w, int h, int comp, const float* data, std::vector<unsigned char>& out) {
    if (w <= 0 || h <= 0 || comp < 1 || comp > 4 || !data) return false;
    char head[160];
    int n = snprintf(head, sizeof head,
                     "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\n"
                     "EXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n",
                     h, w);
    out.assign(head, head + n);
    std::vector<unsigned char> planes((size_t)w * 4);
    const bool rle = w >= 8 && w < 32768;
    for (int y = 0; y < h; ++y) {
        const float* row = data + (size_t)y * w * comp;
        for (int x = 0; x < w; ++x) {
            float rgb[3];
            if (comp >= 3) {
                rgb[0] = row[x * comp + 0];
                rgb[1] = row[x * comp + 1];
                rgb[2] = row[x * comp + 2];
            } else {
                rgb[0] = rgb[1] = rgb[2] = row[x * comp];
            }
            unsigned char q[4];
            to_rgbe(rgb, q);
            if (rle) {
                for (int c = 0; c < 4; ++c) planes[(size_t)c * w + x] = q[c];
            } else {
                out.insert(out.end(), q, q + 4);
            }
        }
        if (rle) {
            const unsigned char sl[4] = {2, 2, (unsigned char)((w >> 8) & 0xff), (unsigned char)(w & 0xff)};
            out.insert(out.end(), sl, sl + 4);
            for (int c = 0; c < 4; ++c) rle_plane(&planes[(size_t)c * w], w, out);
        }
    }
    return true;
}

} // namespace

extern "C" long crt_encode_hdr(int w, int h, int comp, const float* data, unsigned char* out, size_t cap) {
    std::vector<unsigned char> bytes;
    if (!encode(w, h, comp, data, bytes)) return CRT_ERR_INVALID;
    if (out) {
        if (cap < bytes.size()) return CRT_ERR_INVALID;
        memcpy(out, bytes.data(), bytes.size());
    }
    return (long)bytes.size();
}

extern "C" int crt_write_hdr(const char* path, int w, int h, int comp, const float* data) {
    if (!path) return CRT_ERR_INVALID;
    std::vector<unsigned char> bytes;
    if (!encode(w, h, comp, data, bytes)) return CRT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    size_t n = fwrite(bytes.data(), 1, bytes.size(), f);
    int rc = fclose(f);
    return (n == bytes.size() && rc == 0) ? CRT_OK : CRT_ERR_IO;
}

// 8-bit preview (main.cu:816-818): truncating conversion of clamp(c * 255, 0, 255).
extern "C" int crt_to_rgba8(const float* in, size_t n, uint8_t* out) {
    if ((!in || !out) && n != 0) return CRT_ERR_INVALID;
    for (size_t i = 0; i < n; ++i) {
        const float v = in[i] * 255.0f;
        out[i] = v >= 255.0f ? (uint8_t)255 : (v > 0.0f ? (uint8_t)v : (uint8_t)0); // NaN -> 0
    }
    return CRT_OK;
}

extern "C" int crt_write_ppm(const char* path, int w, int h, const uint8_t* rgba8) {
    if (!path || !rgba8 || w <= 0 || h <= 0) return CRT_ERR_INVALID;
    FILE* f = fopen(path, "wb");
    if (!f) return CRT_ERR_IO;
    bool ok = fprintf(f, "P6\n%d %d\n255\n", w, h) > 0;
    std::vector<unsigned char> row((size_t)w * 3u);
    for (int j = 0; j < h && ok; ++j) {
        for (int i = 0; i < w; ++i)
            for (int c = 0; c < 3; ++c) row[(size_t)i * 3u + c] = rgba8[((size_t)j * w + i) * 4u + c];
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    const int rc = fclose(f);
    return (ok && rc == 0) ? CRT_OK : CRT_ERR_IO;
}
