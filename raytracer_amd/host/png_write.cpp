// png_write.cpp — self-contained 8-bit RGBA PNG writer for headless frame output.
//
// Stands where the reference calls stbi_write_png (reference src/FrameBuffer.cpp:89-107).
// The image data is deflated with the fixed Huffman code over LZ77 matches (our own few dozen lines, below; the reference's
// writer carries its own deflate as well): a 1080p frame of the metric scene is a few hundred KB instead of the 8 MB of
// the stored blocks used until round 3.
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

// RFC 1951 with the fixed code (block type 1).  Matches are found with a hash of three bytes and chains of earlier
// positions inside the 32 KiB window, the longest of up to 64 candidates wins; rendered frames are mostly long runs
// (sky, flat walls), which this finds at once.  Bits go out least significant first; Huffman codes most significant
// bit first, hence the reversal.
struct BitSink {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitSink(std::vector<uint8_t> &o) : out(o) {}
    void put(uint32_t bits, int count) {
        acc |= bits << n;
        n += count;
        while (n >= 8) {
            out.push_back(uint8_t(acc));
            acc >>= 8;
            n -= 8;
        }
    }
    void putCode(uint32_t code, int len) { // a Huffman code: most significant bit first
        uint32_t r = 0;
        for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
        put(r, len);
    }
    void flush() {
        if (n > 0) out.push_back(uint8_t(acc));
        acc = 0;
        n = 0;
    }
};

void putSymbol(BitSink &bs, int sym) { // literal / length alphabet of the fixed code
    if (sym <= 143) bs.putCode(0x30 + sym, 8);
    else if (sym <= 255) bs.putCode(0x190 + (sym - 144), 9);
    else if (sym <= 279) bs.putCode(sym - 256, 7);
    else bs.putCode(0xC0 + (sym - 280), 8);
}

void deflateFixed(std::vector<uint8_t> &out, const std::vector<uint8_t> &in) {
    static const int lenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const int lenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const int distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const int distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitSink bs(out);
    bs.put(1, 1); // last block
    bs.put(1, 2); // fixed Huffman codes
    const size_t n = in.size();
    constexpr int HASH_BITS = 15, WINDOW = 32768, MAX_CHAIN = 64;
    std::vector<int> head(1 << HASH_BITS, -1), prev(n ? n : 1, -1);
    auto hashAt = [&](size_t i) { return ((in[i] << 10) ^ (in[i + 1] << 5) ^ in[i + 2]) & ((1 << HASH_BITS) - 1); };
    size_t i = 0;
    while (i < n) {
        int bestLen = 0, bestDist = 0;
        if (i + 3 <= n) {
            const int maxLen = (int)(n - i < 258 ? n - i : 258);
            int cand = head[hashAt(i)], chain = 0;
            while (cand >= 0 && (int)i - cand <= WINDOW && chain++ < MAX_CHAIN) {
                int l = 0;
                while (l < maxLen && in[cand + l] == in[i + l]) l++;
                if (l > bestLen) {
                    bestLen = l;
                    bestDist = (int)i - cand;
                    if (l == maxLen) break;
                }
                cand = prev[cand];
            }
        }
        const size_t step = bestLen >= 3 ? (size_t)bestLen : 1;
        if (bestLen >= 3) {
            int lc = 28;
            while (lenBase[lc] > bestLen) lc--;
            putSymbol(bs, 257 + lc);
            bs.put((uint32_t)(bestLen - lenBase[lc]), lenExtra[lc]);
            int dc = 29;
            while (distBase[dc] > bestDist) dc--;
            bs.putCode((uint32_t)dc, 5);
            bs.put((uint32_t)(bestDist - distBase[dc]), distExtra[dc]);
        } else {
            putSymbol(bs, in[i]);
        }
        for (size_t k = i; k < i + step && k + 3 <= n; k++) { // every position passed enters the chains
            const int h = hashAt(k);
            prev[k] = head[h];
            head[h] = (int)k;
        }
        i += step;
    }
    putSymbol(bs, 256); // end of block
    bs.flush();
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
    // zlib stream: one deflate block with the fixed Huffman codes over LZ77 matches (deflateFixed below)
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x5E);
    deflateFixed(z, raw);
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size();) { // Adler-32, reduced every 5552 bytes (the largest run that cannot overflow)
        const size_t end = raw.size() - i < 5552 ? raw.size() : i + 5552;
        for (; i < end; i++) {
            a += raw[i];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
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
