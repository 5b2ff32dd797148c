// png_write.cpp — self-contained 8-bit RGBA PNG writer for headless frame output.
//
// Stands where the reference calls stbi_write_png (reference src/FrameBuffer.cpp:89-107).
// The image data is wrapped in "stored" deflate blocks (no compression) — frames are written
// once per run, so simplicity wins over file size.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "png_write.hpp"

namespace {

uint32_t crcTable[256];
bool crcReady = false;

void crcInit() {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crcTable[n] = c;
    }
    crcReady = true;
}

uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if (!crcReady) crcInit();
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = crcTable[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

void be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}

void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(0, &out[start], out.size() - start));
}

} // namespace

int rt_write_png_rgba(const char *path, int w, int h, const uint8_t *rgba, bool flip) {
    if (w <= 0 || h <= 0 || !rgba) return -1;
    // raw scanlines, filter type 0
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 4 + 1));
    for (int y = 0; y < h; y++) {
        const uint8_t *row = rgba + (size_t)(flip ? h - 1 - y : y) * w * 4;
        raw.push_back(0);
        raw.insert(raw.end(), row, row + (size_t)w * 4);
    }
    // zlib stream of stored blocks
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (last) break;
    }
    be32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)w);
    be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(6); // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return -2;
    const size_t wr = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return wr == out.size() ? 0 : -3;
}
