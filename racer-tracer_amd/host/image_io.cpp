// image_io.cpp — see image_io.h.
#include "image_io.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace rthost {

bool file_exists(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    return (bool)f;
}

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) f.read(reinterpret_cast<char *>(out.data()), n);
    return (bool)f || f.eof();
}

// =========================================================== JPEG (T.81)
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    void build() { // T.81 Annex C / F.2.2.3
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
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0; // allocated size in blocks (whole MCUs)
    std::vector<uint8_t> plane;     // blocks_w*8 x blocks_h*8 samples
    int pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill() {
        while (nbits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    if (p < end && *p == 0x00) ++p;       // stuffed zero
                    else { hit_marker = true; --p; byte = 0; } // leave the marker for the caller
                }
            }
            acc |= (uint32_t)byte << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

int huff_decode(BitReader &br, const Huffman &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.get(1);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

// The "slow but accurate" integer inverse DCT of the IJG library (13-bit
// constants, two passes), written out from its published description.
inline int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
void idct_islow(const int *coef /* dequantised, natural order */, uint8_t *out, int stride) {
    const int CB = 13, P1 = 2;
    const long F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
               F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    long ws[64];
    for (int c = 0; c < 8; ++c) {
        const int *in = coef + c;
        long *w = ws + c;
        if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
            long dc = (long)in[0] * (1L << P1); // (a left shift of a negative value is undefined before C++20)
            for (int r = 0; r < 8; ++r) w[8 * r] = dc;
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F_0_541;
        long t2 = z1 + z3 * (-F_1_847), t3 = z1 + z2 * F_0_765;
        z2 = in[0]; z3 = in[32];
        long t0 = (z2 + z3) * (1L << CB), t1 = (z2 - z3) * (1L << CB);
        long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = in[56]; t1 = in[40]; t2 = in[24]; t3 = in[8];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
        long z4 = t1 + t3, z5 = (z3 + z4) * F_1_175;
        t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        w[0] = descale(t10 + t3, CB - P1); w[56] = descale(t10 - t3, CB - P1);
        w[8] = descale(t11 + t2, CB - P1); w[48] = descale(t11 - t2, CB - P1);
        w[16] = descale(t12 + t1, CB - P1); w[40] = descale(t12 - t1, CB - P1);
        w[24] = descale(t13 + t0, CB - P1); w[32] = descale(t13 - t0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const long *w = ws + 8 * r;
        uint8_t *o = out + r * stride;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F_0_541;
        long t2 = z1 + z3 * (-F_1_847), t3 = z1 + z2 * F_0_765;
        long t0 = (w[0] + w[4]) * (1L << CB), t1 = (w[0] - w[4]) * (1L << CB);
        long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
        z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
        long z4 = t1 + t3, z5 = (z3 + z4) * F_1_175;
        t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
        z1 *= -F_0_899; z2 *= -F_2_562; z3 *= -F_1_961; z4 *= -F_0_390;
        z3 += z5; z4 += z5;
        t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
        const int SH = CB + P1 + 3;
        auto clamp = [](int v) { v += 128; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        o[0] = clamp(descale(t10 + t3, SH)); o[7] = clamp(descale(t10 - t3, SH));
        o[1] = clamp(descale(t11 + t2, SH)); o[6] = clamp(descale(t11 - t2, SH));
        o[2] = clamp(descale(t12 + t1, SH)); o[5] = clamp(descale(t12 - t1, SH));
        o[3] = clamp(descale(t13 + t0, SH)); o[4] = clamp(descale(t13 - t0, SH));
    }
}

inline uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

} // namespace

bool decode_jpeg_rgba8(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &width, int &height,
                       std::string &why) {
    const uint8_t *d = file.data();
    size_t n = file.size();
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG file"; return false; }
    uint16_t qt[4][64] = {};
    bool qt_present[4] = {false, false, false, false};
    Huffman hdc[4], hac[4];
    std::vector<Component> comps;
    int restart_interval = 0, hmax = 1, vmax = 1;
    width = height = 0;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) { why = "marker expected"; return false; }
        uint8_t m = d[i + 1];
        if (m == 0xFF) { ++i; continue; }
        if (m == 0xD9) break;
        size_t len = be16(d + i + 2);
        if (len < 2 || i + 2 + len > n) { why = "truncated segment"; return false; }
        const uint8_t *s = d + i + 4;
        size_t sl = len - 2;
        if (m == 0xDB) { // DQT
            size_t k = 0;
            while (k < sl) {
                int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || k + (pq ? 128u : 64u) > sl) { why = "bad DQT"; return false; }
                for (int z = 0; z < 64; ++z) {
                    qt[tq][kZigzag[z]] = pq ? be16(s + k + 2 * z) : s[k + z];
                }
                k += pq ? 128 : 64;
                qt_present[tq] = true;
            }
        } else if (m == 0xC0 || m == 0xC1) { // SOF0/1
            if (sl < 6 || s[0] != 8) { why = "only 8-bit JPEG is supported"; return false; }
            height = be16(s + 1);
            width = be16(s + 3);
            int nc = s[5];
            if ((nc != 1 && nc != 3) || sl < 6 + 3 * (size_t)nc || width <= 0 || height <= 0) { why = "unsupported component count"; return false; }
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; ++c) {
                comps[(size_t)c].id = s[6 + 3 * c];
                comps[(size_t)c].h = s[7 + 3 * c] >> 4;
                comps[(size_t)c].v = s[7 + 3 * c] & 15;
                comps[(size_t)c].tq = s[8 + 3 * c];
                if (comps[(size_t)c].h < 1 || comps[(size_t)c].h > 4 || comps[(size_t)c].v < 1 || comps[(size_t)c].v > 4 || comps[(size_t)c].tq > 3) { why = "bad sampling factors"; return false; }
                if (comps[(size_t)c].h > hmax) hmax = comps[(size_t)c].h;
                if (comps[(size_t)c].v > vmax) vmax = comps[(size_t)c].v;
            }
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            why = "only baseline/sequential Huffman JPEG is supported (this file is progressive, lossless or arithmetic)";
            return false;
        } else if (m == 0xC4) { // DHT
            size_t k = 0;
            while (k + 17 <= sl) {
                int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) { why = "bad DHT"; return false; }
                Huffman &h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[k + (size_t)l]; total += h.bits[l]; }
                k += 17;
                if (total > 256 || k + (size_t)total > sl) { why = "bad DHT"; return false; }
                memcpy(h.vals, s + k, (size_t)total);
                k += (size_t)total;
                h.present = true;
                h.build();
            }
        } else if (m == 0xDD) { // DRI
            if (sl >= 2) restart_interval = be16(s);
        } else if (m == 0xDA) { // SOS -> entropy-coded data follows
            if (comps.empty()) { why = "SOS before SOF"; return false; }
            int ns = s[0];
            if (ns != (int)comps.size() || sl < 1 + 2 * (size_t)ns + 3) { why = "non-interleaved scans are not supported"; return false; }
            for (int k = 0; k < ns; ++k) {
                int cid = s[1 + 2 * k];
                Component *c = nullptr;
                for (auto &cc : comps) if (cc.id == cid) c = &cc;
                if (!c) { why = "scan names an unknown component"; return false; }
                c->td = s[2 + 2 * k] >> 4;
                c->ta = s[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3 || !hdc[c->td].present || !hac[c->ta].present || !qt_present[c->tq]) { why = "scan uses a missing table"; return false; }
            }
            int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
            for (auto &c : comps) {
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
                c.pred = 0;
            }
            BitReader br;
            br.p = d + i + 2 + len;
            br.end = d + n;
            int until_restart = restart_interval;
            for (int my = 0; my < mcus_y; ++my)
                for (int mx = 0; mx < mcus_x; ++mx) {
                    if (restart_interval && until_restart == 0) {
                        br.reset(); // skip to the RSTn marker
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (auto &c : comps) c.pred = 0;
                        until_restart = restart_interval;
                    }
                    for (auto &c : comps)
                        for (int by = 0; by < c.v; ++by)
                            for (int bx = 0; bx < c.h; ++bx) {
                                int coef[64] = {0};
                                int t = huff_decode(br, hdc[c.td]);
                                if (t < 0 || t > 16) { why = "corrupt DC code"; return false; }
                                c.pred += extend(br.get(t), t);
                                coef[0] = c.pred * qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    int rs = huff_decode(br, hac[c.ta]);
                                    if (rs < 0) { why = "corrupt AC code"; return false; }
                                    int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r == 15) { k += 16; continue; }
                                        break; // EOB
                                    }
                                    k += r;
                                    if (k > 63) { why = "corrupt AC run"; return false; }
                                    int nat = kZigzag[k];
                                    coef[nat] = extend(br.get(sz), sz) * qt[c.tq][nat];
                                    ++k;
                                }
                                int stride = c.blocks_w * 8;
                                uint8_t *dst = c.plane.data() + ((size_t)(my * c.v + by) * 8) * (size_t)stride + (size_t)(mx * c.h + bx) * 8;
                                idct_islow(coef, dst, stride);
                            }
                    if (restart_interval) --until_restart;
                }
            // upsample (replication) + colour conversion (JFIF YCbCr, 16-bit fixed point)
            rgba.assign((size_t)width * (size_t)height * 4, 255);
            auto fix = [](double x) { return (int)(x * 65536.0 + 0.5); };
            const int cr_r = fix(1.40200), cb_b = fix(1.77200), cr_g = fix(0.71414), cb_g = fix(0.34414), half = 1 << 15;
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) {
                    uint8_t v[3] = {0, 128, 128};
                    for (size_t c = 0; c < comps.size(); ++c) {
                        const Component &cc = comps[c];
                        int sx = x * cc.h / hmax, sy = y * cc.v / vmax;
                        v[c] = cc.plane[(size_t)sy * (size_t)cc.blocks_w * 8 + (size_t)sx];
                    }
                    uint8_t *o = &rgba[((size_t)y * (size_t)width + (size_t)x) * 4];
                    if (comps.size() == 1) {
                        o[0] = o[1] = o[2] = v[0];
                    } else {
                        int yy = v[0], cb = v[1] - 128, cr = v[2] - 128;
                        auto clamp = [](int t) { return (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t)); };
                        o[0] = clamp(yy + ((cr_r * cr + half) >> 16));
                        o[1] = clamp(yy + ((-cb_g * cb - cr_g * cr + half) >> 16));
                        o[2] = clamp(yy + ((cb_b * cb + half) >> 16));
                    }
                }
            return true;
        }
        i += 2 + len;
    }
    why = "no image data (SOS) found";
    return false;
}

// ============================================================ PNG (RFC 2083)
bool decode_png_rgba8(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &width, int &height,
                      std::string &why) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) { why = "not a PNG file"; return false; }
    size_t i = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    width = height = 0;
    while (i + 12 <= file.size()) {
        uint32_t len = (uint32_t)file[i] << 24 | (uint32_t)file[i + 1] << 16 | (uint32_t)file[i + 2] << 8 | file[i + 3];
        if (i + 12 + (size_t)len > file.size()) { why = "truncated chunk"; return false; }
        const uint8_t *t = &file[i + 4], *p = &file[i + 8];
        if (!memcmp(t, "IHDR", 4) && len >= 13) {
            width = (int)((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]);
            height = (int)((uint32_t)p[4] << 24 | (uint32_t)p[5] << 16 | (uint32_t)p[6] << 8 | p[7]);
            depth = p[8]; ctype = p[9]; interlace = p[12];
        } else if (!memcmp(t, "PLTE", 4)) palette.assign(p, p + len);
        else if (!memcmp(t, "tRNS", 4)) trns.assign(p, p + len);
        else if (!memcmp(t, "IDAT", 4)) idat.insert(idat.end(), p, p + len);
        else if (!memcmp(t, "IEND", 4)) break;
        i += 12 + (size_t)len;
    }
    if (width <= 0 || height <= 0 || depth != 8 || interlace != 0) { why = "only 8-bit non-interlaced PNG is supported"; return false; }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) { why = "unknown PNG colour type"; return false; }
    size_t stride = (size_t)width * (size_t)ch;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { why = "corrupt PNG data"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgba.assign((size_t)width * (size_t)height * 4, 255);
    for (int y = 0; y < height; ++y) {
        const uint8_t *line = &raw[(stride + 1) * (size_t)y];
        int ft = line[0];
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)ch ? cur[x - (size_t)ch] : 0, b = prev[x], c = x >= (size_t)ch ? prev[x - (size_t)ch] : 0;
            int v = line[1 + x], pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { why = "bad PNG filter"; return false; }
            cur[x] = (uint8_t)(v + pred);
        }
        for (int x = 0; x < width; ++x) {
            uint8_t *o = &rgba[((size_t)y * (size_t)width + (size_t)x) * 4];
            const uint8_t *s = &cur[(size_t)x * (size_t)ch];
            if (ctype == 0) { o[0] = o[1] = o[2] = s[0]; }
            else if (ctype == 2) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
            else if (ctype == 3) {
                size_t pi = (size_t)s[0] * 3;
                if (pi + 2 < palette.size()) { o[0] = palette[pi]; o[1] = palette[pi + 1]; o[2] = palette[pi + 2]; }
                if (s[0] < trns.size()) o[3] = trns[s[0]];
            } else if (ctype == 4) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
            else { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; }
        }
        prev.swap(cur);
    }
    return true;
}

bool decode_image_rgba8(const std::string &path, std::vector<uint8_t> &rgba, int &width, int &height, std::string &why) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { why = "No such file or directory (os error 2)"; return false; }
    if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) return decode_jpeg_rgba8(file, rgba, width, height, why);
    if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') return decode_png_rgba8(file, rgba, width, height, why);
    why = "The image format could not be determined";
    return false;
}

bool encode_png_rgba8(const uint8_t *rgba, int width, int height, std::vector<uint8_t> &out, std::string &why) {
    if (!rgba || width <= 0 || height <= 0) { why = "empty image"; return false; }
    size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(stride + 1) * (size_t)y] = 0; // filter type None
        memcpy(&raw[(stride + 1) * (size_t)y + 1], rgba + stride * (size_t)y, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { why = "deflate failed"; return false; }
    z.resize(zlen);
    auto put32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); };
    auto chunk = [&](const char *type, const uint8_t *data, size_t len) {
        put32((uint32_t)len);
        size_t at = out.size();
        out.insert(out.end(), type, type + 4);
        if (len) out.insert(out.end(), data, data + len);
        put32((uint32_t)crc32(0L, &out[at], (uInt)(len + 4)));
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.assign(sig, sig + 8);
    uint8_t ihdr[13] = {(uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8), (uint8_t)width,
                        (uint8_t)(height >> 24), (uint8_t)(height >> 16), (uint8_t)(height >> 8), (uint8_t)height,
                        8, 6, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), z.size());
    chunk("IEND", nullptr, 0);
    return true;
}

// ======================================================= SHA-256 (FIPS 180-4)
std::string sha256_hex_upper(const uint8_t *data, size_t len) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
        0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
        0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
        0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
        0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
        0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
        0xc67178f2};
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    auto block = [&](const uint8_t *p) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), maj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    };
    size_t full = len / 64;
    for (size_t i = 0; i < full; ++i) block(data + 64 * i);
    uint8_t tail[128] = {0};
    size_t rem = len - 64 * full;
    if (rem) memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - (size_t)i] = (uint8_t)(bits >> (8 * i));
    block(tail);
    if (tl == 128) block(tail + 64);
    static const char *hex = "0123456789ABCDEF";
    std::string out;
    for (int i = 0; i < 8; ++i)
        for (int s = 28; s >= 0; s -= 4) out.push_back(hex[(h[i] >> s) & 15]);
    return out;
}

} // namespace rthost
