// Image files of the host front end without third-party codecs: PNG output as the reference writes it
// (Bitmap::savePNG, src/bitmap.cpp:122-144: sRGB, 8 bits, three channels) and PNG / JPEG / PPM / PFM input for textures
// (Bitmap::load, src/bitmap.cpp:68-93: 8-bit values / 255, sRGB -> linear).  The PNG writer stores the image in
// uncompressed deflate blocks; the reader carries its own inflate (fixed and dynamic Huffman codes) and undoes the five
// scanline filters.  JPEG files go through jpeg_decode.cpp.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pt_host_internal.h"

namespace pt_host {

namespace {

uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}

uint32_t adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) {
        a = (a + p[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back(uint8_t(x >> s));
}

void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    put_be32(out, uint32_t(data.size()));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32(out.data() + start, out.size() - start));
}

// ---- inflate (RFC 1951) ------------------------------------------------------------------------------------------

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t buf = 0;
    int cnt = 0;
    uint32_t bits(int k) {
        while (cnt < k) {
            if (pos >= n) throw std::runtime_error("PNG: compressed data ends early");
            buf |= uint32_t(p[pos++]) << cnt;
            cnt += 8;
        }
        const uint32_t v = k ? buf & ((1u << k) - 1u) : 0u;
        buf >>= k;
        cnt -= k;
        return v;
    }
    void align() {
        buf = 0;
        cnt = 0;
    }
};

struct Huffman {
    uint16_t count[16] = {};
    uint16_t symbol[288] = {};
    void build(const uint8_t* len, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) ++count[len[i]];
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = uint16_t(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = uint16_t(i);
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= int(br.bits(1));
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        throw std::runtime_error("PNG: bad Huffman code");
    }
};

std::vector<uint8_t> inflate(const uint8_t* src, size_t n, size_t expect) {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (n < 6) throw std::runtime_error("PNG: compressed data too short");
    if ((src[0] & 0x0f) != 8 || ((uint32_t(src[0]) << 8) | src[1]) % 31u != 0 || (src[1] & 0x20)) throw std::runtime_error("PNG: not a zlib stream");
    BitReader br{src + 2, n - 2};
    std::vector<uint8_t> out;
    out.reserve(expect);
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) throw std::runtime_error("PNG: compressed data ends early");
            const uint32_t len = br.p[br.pos] | (uint32_t(br.p[br.pos + 1]) << 8), nlen = br.p[br.pos + 2] | (uint32_t(br.p[br.pos + 3]) << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) throw std::runtime_error("PNG: bad stored block");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) throw std::runtime_error("PNG: bad block type");
        Huffman lit, dist;
        if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(l, 288);
            uint8_t d[30];
            std::memset(d, 5, sizeof d);
            dist.build(d, 30);
        } else {
            const int hlit = int(br.bits(5)) + 257, hdist = int(br.bits(5)) + 1, hclen = int(br.bits(4)) + 4;
            if (hlit > 286 || hdist > 30) throw std::runtime_error("PNG: bad code counts");
            uint8_t cl[19] = {};
            for (int i = 0; i < hclen; ++i) cl[order[i]] = uint8_t(br.bits(3));
            Huffman code;
            code.build(cl, 19);
            uint8_t l[320] = {};
            for (int i = 0; i < hlit + hdist;) {
                const int sym = code.decode(br);
                if (sym < 16) {
                    l[i++] = uint8_t(sym);
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (i == 0) throw std::runtime_error("PNG: bad code lengths");
                        val = l[i - 1];
                        rep = 3 + int(br.bits(2));
                    } else if (sym == 17) {
                        rep = 3 + int(br.bits(3));
                    } else {
                        rep = 11 + int(br.bits(7));
                    }
                    if (i + rep > hlit + hdist) throw std::runtime_error("PNG: bad code lengths");
                    while (rep--) l[i++] = uint8_t(val);
                }
            }
            lit.build(l, hlit);
            dist.build(l + hlit, hdist);
        }
        for (;;) {
            const int sym = lit.decode(br);
            if (sym < 256) {
                out.push_back(uint8_t(sym));
            } else if (sym == 256) {
                break;
            } else {
                if (sym > 285) throw std::runtime_error("PNG: bad length symbol");
                const size_t len = len_base[sym - 257] + br.bits(len_extra[sym - 257]);
                const int ds = dist.decode(br);
                if (ds > 29) throw std::runtime_error("PNG: bad distance symbol");
                const size_t d = dist_base[ds] + br.bits(dist_extra[ds]);
                if (d > out.size()) throw std::runtime_error("PNG: distance beyond the start");
                for (size_t k = 0; k < len; ++k) out.push_back(out[out.size() - d]);
            }
        }
    }
    return out;
}

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// sRGB -> linear (Color3f::toLinearRGB, include/pt/color.h:70-83)
float to_linear(float v) { return v <= 0.04045f ? v * (1.0f / 12.92f) : std::pow((v + 0.055f) * (1.0f / 1.055f), 2.4f); }
// linear -> sRGB (Color3f::toSRGB, include/pt/color.h:57-69)
float to_srgb(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f; }

std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return {};
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

}  // namespace

void write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve(size_t(h) * (size_t(w) * 3 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);  // filter type "none"
        raw.insert(raw.end(), rgb + size_t(y) * w * 3, rgb + size_t(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    for (size_t pos = 0; pos < raw.size() || raw.empty();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (raw.empty()) break;
    }
    put_be32(z, adler32(raw.data(), raw.size()));
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, RGB, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(out.data()), std::streamsize(out.size()));
    if (!f) throw std::runtime_error("could not write " + path);
}

// Bitmap::savePNG (src/bitmap.cpp:122-144): toSRGB (when asked), * 255, clamped to [0, 255], truncated
void save_png(const std::string& path, const float* rgb, uint32_t w, uint32_t h, bool srgb) {
    std::vector<uint8_t> out(size_t(w) * h * 3);
    for (size_t i = 0; i < out.size(); ++i) {
        const float v = srgb ? to_srgb(rgb[i]) : rgb[i];
        out[i] = uint8_t(std::min(std::max(255.0f * v, 0.0f), 255.0f));
    }
    write_png_rgb8(path, out.data(), w, h);
}

// 8-bit RGB of a PNG file (grey, palette and alpha variants are expanded / dropped; 16-bit samples keep their high byte)
bool read_png_rgb8(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    const std::vector<uint8_t> file = read_file(path);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return false;
    std::vector<uint8_t> idat, palette;
    int depth = 0, colour = 0, interlace = 0;
    w = h = 0;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const std::string type(reinterpret_cast<const char*>(&file[pos + 4]), 4);
        if (pos + 12 + size_t(len) > file.size()) throw std::runtime_error("PNG: chunk beyond the end of " + path);
        const uint8_t* d = &file[pos + 8];
        if (crc32(&file[pos + 4], size_t(len) + 4) != be32(d + len)) throw std::runtime_error("PNG: checksum mismatch in " + path);
        if (type == "IHDR" && len >= 13) {
            w = be32(d);
            h = be32(d + 4);
            depth = d[8];
            colour = d[9];
            interlace = d[12];
        } else if (type == "PLTE") {
            palette.assign(d, d + len);
        } else if (type == "IDAT") {
            idat.insert(idat.end(), d, d + len);
        } else if (type == "IEND") {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (w == 0 || h == 0) throw std::runtime_error("PNG: no header in " + path);
    if (interlace != 0) throw std::runtime_error("PNG: interlaced files are not supported (" + path + ")");
    const int channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : colour == 6 ? 4 : 0;
    if (!channels || !(depth == 8 || depth == 16 || ((colour == 0 || colour == 3) && (depth == 1 || depth == 2 || depth == 4))))
        throw std::runtime_error("PNG: unsupported colour type / bit depth in " + path);
    const size_t bpp_bits = size_t(channels) * depth, stride = (size_t(w) * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<uint8_t> raw = inflate(idat.data(), idat.size(), (stride + 1) * h);
    if (raw.size() < (stride + 1) * h) throw std::runtime_error("PNG: image data too short in " + path);
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgb.assign(size_t(w) * h * 3, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* line = &raw[size_t(y) * (stride + 1)];
        const int filter = line[0];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = line[1 + i];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error("PNG: bad filter type in " + path);
            }
            cur[i] = uint8_t(v);
        }
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* o = &rgb[(size_t(y) * w + x) * 3];
            auto sample = [&](int ch) -> uint8_t {  // 8-bit value of channel ch of pixel x
                if (depth == 8) return cur[size_t(x) * channels + ch];
                if (depth == 16) return cur[(size_t(x) * channels + ch) * 2];
                const size_t bit = size_t(x) * depth;
                const int v = (cur[bit / 8] >> (8 - depth - int(bit % 8))) & ((1 << depth) - 1);
                return colour == 3 ? uint8_t(v) : uint8_t(v * 255 / ((1 << depth) - 1));
            };
            if (colour == 3) {
                const size_t idx = sample(0);
                if (idx * 3 + 2 >= palette.size()) throw std::runtime_error("PNG: palette index out of range in " + path);
                o[0] = palette[idx * 3];
                o[1] = palette[idx * 3 + 1];
                o[2] = palette[idx * 3 + 2];
            } else if (colour == 0 || colour == 4) {
                o[0] = o[1] = o[2] = sample(0);
            } else {
                o[0] = sample(0);
                o[1] = sample(1);
                o[2] = sample(2);
            }
        }
        prev.swap(cur);
    }
    return true;
}

// Bitmap::load (src/bitmap.cpp:68-93) for the formats this build decodes: linear RGB floats, row 0 first.
// Returns false when the file does not exist or is of another format.
bool load_image_linear(const std::string& path, std::vector<float>& rgb, uint32_t& w, uint32_t& h) {
    std::vector<uint8_t> px;
    if (read_png_rgb8(path, px, w, h) || read_jpeg_rgb8(read_file(path), px, w, h)) {
        rgb.resize(px.size());
        for (size_t i = 0; i < px.size(); ++i) rgb[i] = to_linear(float(px[i]) / 255.f);  // src/bitmap.cpp:79-90
        return true;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    int iw = 0, ih = 0;
    if (magic == "P6") {
        int maxv = 0;
        f >> iw >> ih >> maxv;
        f.get();
        if (iw <= 0 || ih <= 0 || maxv != 255) return false;
        std::vector<unsigned char> raw(size_t(iw) * ih * 3);
        f.read(reinterpret_cast<char*>(raw.data()), std::streamsize(raw.size()));
        if (!f) return false;
        rgb.resize(raw.size());
        for (size_t i = 0; i < raw.size(); ++i) rgb[i] = to_linear(float(raw[i]) / 255.f);
    } else if (magic == "PF") {
        float scale = 0;
        f >> iw >> ih >> scale;
        f.get();
        if (iw <= 0 || ih <= 0 || scale >= 0) return false;  // little-endian only
        rgb.resize(size_t(iw) * ih * 3);
        std::vector<float> row(size_t(iw) * 3);
        for (int y = ih - 1; y >= 0; --y) {  // PFM stores the bottom row first
            f.read(reinterpret_cast<char*>(row.data()), std::streamsize(row.size() * sizeof(float)));
            std::memcpy(&rgb[size_t(y) * iw * 3], row.data(), row.size() * sizeof(float));
        }
        if (!f) return false;
    } else {
        return false;
    }
    w = uint32_t(iw);
    h = uint32_t(ih);
    return true;
}

}  // namespace pt_host
