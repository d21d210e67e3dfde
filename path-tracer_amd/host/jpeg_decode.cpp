// JPEG decoder for the textures the reference's scenes name (Bitmap::load, src/bitmap.cpp:68-93, reads them through
// stb_image): 8-bit baseline, extended sequential and progressive Huffman-coded files (ITU-T T.81), one or three
// components, any sampling factors, restart intervals.  Not decoded: arithmetic coding, lossless, 12-bit, CMYK.
// Coefficients of all scans are collected first, then every block goes through a floating-point inverse DCT; chroma
// planes sampled at half rate are brought up with the triangle filter that libjpeg and stb_image both use (3/4 of the
// nearer, 1/4 of the farther sample), and YCbCr is converted by the JFIF equations.  Different decoders round the inverse
// DCT and the colour conversion differently, so decoded values agree with theirs to within a level or two, not bit for bit.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "pt_host_internal.h"

namespace pt_host {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool defined = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    void prepare() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;      // tables of the current scan
    int blocks_w = 0, blocks_h = 0;   // blocks in the plane (padded to whole MCUs)
    int scan_w = 0, scan_h = 0;       // blocks a scan of this component alone covers
    int dc_pred = 0;
    std::vector<int16_t> coef;        // blocks_w * blocks_h * 64, natural (row-major) order inside a block
    std::vector<uint8_t> plane;       // blocks_w * 8 wide
};

struct Decoder {
    const uint8_t* d;
    size_t n, pos = 0;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, have_frame = false;
    Component comp[3];
    uint16_t qt[4][64] = {};
    bool qt_defined[4] = {};
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    // entropy-coded segment
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    int eobrun = 0;
    int ss = 0, se = 63, ah = 0, al = 0;

    [[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("JPEG: ") + what); }
    int u8() {
        if (pos >= n) bad("file ends early");
        return d[pos++];
    }
    int u16() {
        const int a = u8();
        return (a << 8) | u8();
    }

    // ---- bits ------------------------------------------------------------------------------------------------------
    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xff) {
                    const int b2 = pos + 1 < n ? d[pos + 1] : 0xd9;
                    if (b2 == 0) {
                        pos += 2;
                    } else {
                        hit_marker = true;  // the marker stays in the stream; zeros from here on
                        b = 0;
                    }
                } else {
                    ++pos;
                }
            }
            bitbuf |= uint32_t(b) << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bit() {
        if (bitcnt < 1) fill();
        const int v = int(bitbuf >> 31);
        bitbuf <<= 1;
        --bitcnt;
        return v;
    }
    int bits(int k) {
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        const int v = int(bitbuf >> (32 - k));
        bitbuf <<= k;
        bitcnt -= k;
        return v;
    }
    int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
    int receive_extend(int s) { return s ? extend(bits(s), s) : 0; }
    int decode(const HuffTable& t) {
        int code = bit(), l = 1;
        while (code > t.maxcode[l]) {
            code = (code << 1) | bit();
            if (++l > 16) bad("bad Huffman code");
        }
        return t.vals[t.valptr[l] + code - t.mincode[l]];
    }
    void reset_entropy() {
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
        eobrun = 0;
        for (int c = 0; c < ncomp; ++c) comp[c].dc_pred = 0;
    }

    // ---- blocks ----------------------------------------------------------------------------------------------------
    void block_sequential(Component& c, int16_t* b) {
        const int t = decode(dc[c.td]);
        c.dc_pred += receive_extend(t);
        b[0] = int16_t(c.dc_pred);
        for (int k = 1; k < 64; ++k) {
            const int rs = decode(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 15;
            } else {
                k += r;
                if (k > 63) bad("coefficient index out of range");
                b[kZigzag[k]] = int16_t(receive_extend(s));
            }
        }
    }
    void block_dc_first(Component& c, int16_t* b) {
        const int t = decode(dc[c.td]);
        c.dc_pred += receive_extend(t);
        b[0] = int16_t(c.dc_pred * (1 << al));
    }
    void block_dc_refine(int16_t* b) {
        if (bit()) b[0] = int16_t(b[0] | (1 << al));
    }
    void block_ac_first(Component& c, int16_t* b) {
        if (eobrun > 0) {
            --eobrun;
            return;
        }
        for (int k = ss; k <= se; ++k) {
            const int rs = decode(ac[c.ta]), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += bits(r);
                    break;
                }
                k += 15;
            } else {
                k += r;
                if (k > 63) bad("coefficient index out of range");
                b[kZigzag[k]] = int16_t(receive_extend(s) * (1 << al));
            }
        }
    }
    void refine_nonzero(int16_t& v, int p1, int m1) {
        if (bit() && (v & p1) == 0) v = int16_t(v + (v >= 0 ? p1 : m1));
    }
    void block_ac_refine(Component& c, int16_t* b) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = decode(ac[c.ta]);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = bit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += bits(r);
                    break;
                }
                // over the coefficients that are already non-zero (each takes a correction bit) and r that are still zero
                do {
                    int16_t& v = b[kZigzag[k]];
                    if (v != 0) {
                        refine_nonzero(v, p1, m1);
                    } else if (--r < 0) {
                        break;
                    }
                    ++k;
                } while (k <= se);
                if (s && k <= 63) b[kZigzag[k]] = int16_t(s);
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t& v = b[kZigzag[k]];
                if (v != 0) refine_nonzero(v, p1, m1);
            }
            --eobrun;
        }
    }
    void decode_block(Component& c, int bx, int by) {
        int16_t* b = &c.coef[(size_t(by) * c.blocks_w + bx) * 64];
        if (!progressive)
            block_sequential(c, b);
        else if (ss == 0)
            ah == 0 ? block_dc_first(c, b) : block_dc_refine(b);
        else
            ah == 0 ? block_ac_first(c, b) : block_ac_refine(c, b);
    }

    // ---- segments --------------------------------------------------------------------------------------------------
    void read_dqt(int len) {
        while (len > 0) {
            const int pq = u8(), t = pq & 15, wide = pq >> 4;
            if (t > 3 || wide > 1) bad("bad quantisation table");
            for (int i = 0; i < 64; ++i) qt[t][kZigzag[i]] = uint16_t(wide ? u16() : u8());
            qt_defined[t] = true;
            len -= 1 + 64 * (wide ? 2 : 1);
        }
    }
    void read_dht(int len) {
        while (len > 0) {
            const int tc = u8(), t = tc & 15, cls = tc >> 4;
            if (t > 3 || cls > 1) bad("bad Huffman table");
            HuffTable& h = cls ? ac[t] : dc[t];
            int total = 0;
            for (int l = 1; l <= 16; ++l) total += (h.bits[l] = uint8_t(u8()));
            if (total > 256) bad("bad Huffman table");
            for (int i = 0; i < total; ++i) h.vals[i] = uint8_t(u8());
            h.prepare();
            len -= 17 + total;
        }
    }
    void read_sof(int marker) {
        if (have_frame) bad("more than one frame");
        progressive = marker == 0xc2;
        if (u8() != 8) bad("only 8-bit samples are supported");
        height = u16();
        width = u16();
        ncomp = u8();
        if (width <= 0 || height <= 0) bad("empty image");
        if (ncomp != 1 && ncomp != 3) bad("only grey and three-component files are supported");
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = u8();
            const int hv = u8();
            c.h = hv >> 4;
            c.v = hv & 15;
            c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) bad("bad component");
            hmax = std::max(hmax, c.h);
            vmax = std::max(vmax, c.v);
        }
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
        mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.blocks_w = mcus_x * c.h;
            c.blocks_h = mcus_y * c.v;
            const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
            c.scan_w = (cw + 7) / 8;
            c.scan_h = (ch + 7) / 8;
            c.coef.assign(size_t(c.blocks_w) * c.blocks_h * 64, 0);
        }
        have_frame = true;
    }
    void restart_if_due(int& until_restart) {
        if (restart_interval == 0) return;
        if (--until_restart > 0) return;
        // the bits left in the buffer belong to padding; the next two bytes in the stream are the RSTn marker
        if (!hit_marker) {
            while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7)) ++pos;
        }
        if (pos + 1 < n && d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7) pos += 2;
        reset_entropy();
        until_restart = restart_interval;
    }
    void read_sos() {
        if (!have_frame) bad("scan before the frame header");
        const int ns = u8();
        if (ns < 1 || ns > ncomp) bad("bad scan");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = u8(), tt = u8();
            sc[i] = nullptr;
            for (int k = 0; k < ncomp; ++k)
                if (comp[k].id == id) sc[i] = &comp[k];
            if (!sc[i]) bad("scan names an unknown component");
            sc[i]->td = tt >> 4;
            sc[i]->ta = tt & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) bad("bad table selector");
        }
        ss = u8();
        se = u8();
        const int a = u8();
        ah = a >> 4;
        al = a & 15;
        if (!progressive) {
            ss = 0;
            se = 63;
            ah = al = 0;
        } else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss != 0 && ns != 1) || al > 13) {
            bad("bad progressive scan parameters");
        }
        for (int i = 0; i < ns; ++i) {
            const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
            if ((need_dc && !dc[sc[i]->td].defined) || (need_ac && !ac[sc[i]->ta].defined)) bad("scan uses an undefined Huffman table");
        }
        reset_entropy();
        int until_restart = restart_interval;
        if (ns == 1) {  // a scan of one component covers that component's own blocks, row by row
            Component& c = *sc[0];
            for (int by = 0; by < c.scan_h; ++by)
                for (int bx = 0; bx < c.scan_w; ++bx) {
                    decode_block(c, bx, by);
                    restart_if_due(until_restart);
                }
        } else {
            for (int my = 0; my < mcus_y; ++my)
                for (int mx = 0; mx < mcus_x; ++mx) {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h) decode_block(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v);
                    restart_if_due(until_restart);
                }
        }
        // leave the position at the marker that ended the entropy-coded data
        if (!hit_marker)
            while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] != 0 && !(d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7))) ++pos;
        bitcnt = 0;
        bitbuf = 0;
        hit_marker = false;
    }

    // ---- reconstruction --------------------------------------------------------------------------------------------
    void idct_planes() {
        double cs[8][8];  // cs[x][u] = c(u) / 2 * cos((2x + 1) u pi / 16)
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) cs[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        for (int ci = 0; ci < ncomp; ++ci) {
            Component& c = comp[ci];
            if (!qt_defined[c.tq]) bad("component uses an undefined quantisation table");
            const int stride = c.blocks_w * 8;
            c.plane.assign(size_t(stride) * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    const int16_t* b = &c.coef[(size_t(by) * c.blocks_w + bx) * 64];
                    double f[64], t[64];
                    for (int i = 0; i < 64; ++i) f[i] = double(b[i]) * qt[c.tq][i];
                    for (int v = 0; v < 8; ++v)  // rows: over u
                        for (int x = 0; x < 8; ++x) {
                            double s = 0;
                            for (int u = 0; u < 8; ++u) s += cs[x][u] * f[v * 8 + u];
                            t[v * 8 + x] = s;
                        }
                    for (int x = 0; x < 8; ++x)
                        for (int y = 0; y < 8; ++y) {
                            double s = 0;
                            for (int v = 0; v < 8; ++v) s += cs[y][v] * t[v * 8 + x];
                            const long r = std::lround(s + 128.0);
                            c.plane[size_t(by * 8 + y) * stride + bx * 8 + x] = uint8_t(r < 0 ? 0 : r > 255 ? 255 : r);
                        }
                }
        }
    }
    // one component at full resolution: width x height samples
    std::vector<uint8_t> upsample(const Component& c) {
        const int fx = hmax / c.h, fy = vmax / c.v;
        const int stride = c.blocks_w * 8;
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
        std::vector<uint8_t> out(size_t(width) * height);
        const bool exact = hmax % c.h == 0 && vmax % c.v == 0;
        if (exact && fx == 1 && fy == 1) {
            for (int y = 0; y < height; ++y) std::memcpy(&out[size_t(y) * width], &c.plane[size_t(y) * stride], size_t(width));
        } else if (exact && fx == 2 && fy == 1) {  // 3/4 of the nearer sample, 1/4 of the farther one
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = &c.plane[size_t(y) * stride];
                for (int x = 0; x < width; ++x) {
                    const int i = x >> 1, j = (x & 1) ? std::min(i + 1, cw - 1) : std::max(i - 1, 0);
                    out[size_t(y) * width + x] = uint8_t((3 * in[i] + in[j] + ((x & 1) ? 2 : 1)) >> 2);
                }
            }
        } else if (exact && fx == 2 && fy == 2) {
            for (int y = 0; y < height; ++y) {
                const int r = y >> 1, r2 = (y & 1) ? std::min(r + 1, ch - 1) : std::max(r - 1, 0);
                const uint8_t *in0 = &c.plane[size_t(r) * stride], *in1 = &c.plane[size_t(r2) * stride];
                for (int x = 0; x < width; ++x) {
                    const int i = x >> 1, j = (x & 1) ? std::min(i + 1, cw - 1) : std::max(i - 1, 0);
                    const int near = 3 * in0[i] + in1[i], far = 3 * in0[j] + in1[j];
                    out[size_t(y) * width + x] = uint8_t((3 * near + far + ((x & 1) ? 7 : 8)) >> 4);
                }
            }
        } else if (exact && fx == 1 && fy == 2) {
            for (int y = 0; y < height; ++y) {
                const int r = y >> 1, r2 = (y & 1) ? std::min(r + 1, ch - 1) : std::max(r - 1, 0);
                const uint8_t *in0 = &c.plane[size_t(r) * stride], *in1 = &c.plane[size_t(r2) * stride];
                for (int x = 0; x < width; ++x) out[size_t(y) * width + x] = uint8_t((3 * in0[x] + in1[x] + ((y & 1) ? 2 : 1)) >> 2);
            }
        } else {  // other ratios: the sample whose cell the pixel falls into
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x)
                    out[size_t(y) * width + x] = c.plane[size_t(std::min(y * c.v / vmax, ch - 1)) * stride + std::min(x * c.h / hmax, cw - 1)];
        }
        return out;
    }

    void run(std::vector<uint8_t>& rgb) {
        if (n < 4 || d[0] != 0xff || d[1] != 0xd8) bad("not a JPEG file");
        pos = 2;
        bool done = false, adobe_rgb = false;
        while (!done) {
            int b = u8();
            if (b != 0xff) continue;  // (garbage between segments is skipped)
            while (b == 0xff) b = u8();
            if (b == 0 || (b >= 0xd0 && b <= 0xd7) || b == 0x01) continue;
            if (b == 0xd9) break;
            const int len = u16() - 2;
            if (len < 0 || pos + size_t(len) > n) bad("segment beyond the end of the file");
            const size_t next = pos + size_t(len);
            switch (b) {
                case 0xdb: read_dqt(len); break;
                case 0xc4: read_dht(len); break;
                case 0xc0:
                case 0xc1:
                case 0xc2: read_sof(b); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    bad("lossless, hierarchical and arithmetic-coded files are not supported");
                case 0xdd: restart_interval = u16(); break;
                case 0xee:  // Adobe: colour transform 0 with three components means the planes are R, G, B already
                    if (len >= 12 && std::memcmp(d + pos, "Adobe", 5) == 0) adobe_rgb = d[pos + 11] == 0;
                    break;
                case 0xda:
                    read_sos();  // the header, then the entropy-coded data that follow it
                    continue;    // position already at the next marker
                default: break;
            }
            pos = next;
        }
        if (!have_frame) bad("no frame header");
        idct_planes();
        rgb.resize(size_t(width) * height * 3);
        if (ncomp == 1) {
            const std::vector<uint8_t> y = upsample(comp[0]);
            for (size_t i = 0; i < y.size(); ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = y[i];
            return;
        }
        const std::vector<uint8_t> p0 = upsample(comp[0]), p1 = upsample(comp[1]), p2 = upsample(comp[2]);
        auto clamp8 = [](double v) { const long r = std::lround(v); return uint8_t(r < 0 ? 0 : r > 255 ? 255 : r); };
        for (size_t i = 0; i < p0.size(); ++i) {
            if (adobe_rgb) {
                rgb[3 * i] = p0[i];
                rgb[3 * i + 1] = p1[i];
                rgb[3 * i + 2] = p2[i];
            } else {  // JFIF: full-range BT.601
                const double y = p0[i], cb = p1[i] - 128.0, cr = p2[i] - 128.0;
                rgb[3 * i] = clamp8(y + 1.402 * cr);
                rgb[3 * i + 1] = clamp8(y - 0.344136286 * cb - 0.714136286 * cr);
                rgb[3 * i + 2] = clamp8(y + 1.772 * cb);
            }
        }
    }
};

}  // namespace

// 8-bit RGB of a JPEG file; false when the file is not a JPEG (no SOI marker), std::runtime_error when it is malformed
// or of a kind this decoder does not handle.
bool read_jpeg_rgb8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    if (file.size() < 4 || file[0] != 0xff || file[1] != 0xd8) return false;
    Decoder dec{file.data(), file.size()};
    dec.run(rgb);
    w = uint32_t(dec.width);
    h = uint32_t(dec.height);
    return true;
}

}  // namespace pt_host
