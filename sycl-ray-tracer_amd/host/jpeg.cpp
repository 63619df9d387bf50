// jpeg.cpp — JPEG (ITU-T T.81) decoder for embedded glTF textures: Huffman coding, 8-bit samples, baseline /
// extended-sequential (SOF0/SOF1) and progressive (SOF2) frames, 1, 3 or 4 (Adobe CMYK / YCCK) components, any sampling factors,
// restart intervals. Output is RGBA8, as the reference's loader gets it from
// stbi_load_from_memory(bytes, size, &w, &h, &comp, 4) (deps/include/tiny_gltf.h:2603-2634 of the reference).
//
// ATTRIBUTION. This file is DERIVED FROM stb_image v2.29 by Sean Barrett and contributors (http://nothings.org/stb; the third-party
// header the reference vendors as deps/include/stb_image.h and its loader calls): the Huffman table construction and fast-path lookup,
// the baseline and progressive (DC / AC, first and refinement scans) block decoders, the islow inverse DCT "derived from jidctint", the
// triangle-filter chroma upsampling and the fixed-point YCbCr conversion are stb_image's algorithms, re-typed here as C++ with exceptions
// and bounds checks, with its control flow, field names and error strings largely kept. stb_image is dual-licensed: MIT (Copyright (c)
// 2017 Sean Barrett) or public domain (unlicense.org); the MIT notice follows as that licence requires.
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation files
//   (the "Software"), to deal in the Software without restriction, including without limitation the rights to use, copy, modify, merge,
//   publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom the Software is furnished to do
//   so, subject to the following conditions: The above copyright notice and this permission notice shall be included in all copies or
//   substantial portions of the Software. THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING
//   BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE
//   AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// It is third-party code outside the hot path (the loader; SURVEY §2 row 18 lists deps/ as third party). Why a
// derived decoder and not a clean-room one: the texels must be the ones the reference's loader produces, and every step with a numerical
// choice has to be stb_image's for that — checked bit for bit against the reference's own copy of stb_image, compiled where it lies
// (by the test suite), in tests/test_ref_stb.py:
//   * inverse DCT: the Loeffler-Ligtenberg-Moschytz "islow" factorisation with 12-bit constants, two extra bits kept
//     after the column pass, +128 level shift folded into the final rounding;
//   * chroma upsampling: the 3:1 "triangle" filter horizontally, vertically or both ((3a+b+2)>>2, (9a+3b+3c+d+8)>>4),
//     nearest for other factors;
//   * YCbCr -> RGB in 20-bit fixed point (1.40200, 0.71414, 0.34414, 1.77200 rounded to 12 bits, the 0.34414 term
//     truncated to its upper 16 bits).
#include "jpeg.h"

#include "png.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

namespace rthost {
namespace {

[[noreturn]] void bad(const char* what) { throw std::runtime_error(std::string("JPEG: ") + what); }

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run may step past 63: land on the last coefficient instead of outside the block
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    // canonical code tables (T.81 Annex C / F.2.2.3) + a 9-bit lookahead
    uint8_t fast[512];
    uint16_t code[256];
    uint8_t values[256], size[257];
    uint32_t maxcode[18];
    int delta[17];
    bool present = false;

    void build(const uint8_t counts[16], const uint8_t* vals, int n) {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < counts[i]; ++j) {
                if (k >= 256) bad("bad Huffman table");
                size[k++] = (uint8_t)(i + 1);
            }
        size[k] = 0;
        std::memcpy(values, vals, (size_t)n);
        uint32_t c = 0;
        k = 0;
        for (int j = 1; j <= 16; ++j) {
            delta[j] = k - (int)c;
            if (size[k] == j) {
                while (size[k] == j) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1u << j)) bad("bad Huffman code lengths");
            }
            maxcode[j] = c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0xffffffffu;
        std::memset(fast, 255, sizeof(fast));
        for (int i = 0; i < k; ++i) {
            const int s = size[i];
            if (s <= 9) {
                const int first = code[i] << (9 - s), m = 1 << (9 - s);
                for (int j = 0; j < m; ++j) fast[first + j] = (uint8_t)i;
            }
        }
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int x = 0, y = 0;   // size in samples
    int w2 = 0, h2 = 0; // size padded to whole MCUs
    int bw = 0, bh = 0; // blocks per row / column (padded), progressive coefficient storage
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff;
};

struct Decoder {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t width = 0, height = 0;
    int ncomp = 0;
    Component comp[4];
    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    Huffman hdc[4], hac[4];
    int restart_interval = 0;
    bool progressive = false, jfif = false;
    int app14_transform = -1;
    int hmax = 1, vmax = 1, mcu_w = 8, mcu_h = 8, mcus_x = 0, mcus_y = 0;
    // scan state
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
    // bit reader
    uint32_t bits = 0;
    int nbits = 0;
    uint8_t marker = 0xff; // 0xff: none pending
    bool nomore = false;

    int get8() { return p < end ? *p++ : 0; }
    int get16() {
        const int a = get8();
        return (a << 8) | get8();
    }

    void grow() {
        do {
            int b = nomore ? 0 : get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8(); // fill bytes
                if (c != 0) {
                    marker = (uint8_t)c;
                    nomore = true;
                    return;
                }
            }
            bits |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        } while (nbits <= 24);
    }

    int decode_symbol(const Huffman& h) {
        if (nbits < 16) grow();
        const int look = (int)(bits >> 23);
        int k = h.fast[look];
        if (k < 255) {
            const int s = h.size[k];
            if (s > nbits) return -1;
            bits <<= s, nbits -= s;
            return h.values[k];
        }
        const uint32_t top = bits >> 16;
        int len = 10;
        while (top >= h.maxcode[len]) ++len;
        if (len == 17) {
            nbits -= 16;
            return -1;
        }
        if (len > nbits) return -1;
        const int idx = (int)((bits >> (32 - len)) & ((1u << len) - 1u)) + h.delta[len];
        if (idx < 0 || idx >= 256) return -1;
        bits <<= len, nbits -= len;
        return h.values[idx];
    }

    // n bits as a signed value in the JPEG "extend" convention (T.81 F.2.2.1)
    int receive_extend(int n) {
        if (n == 0) return 0;
        if (nbits < n) grow();
        if (nbits < n) return 0;
        const uint32_t v = bits >> (32 - n);
        bits <<= n, nbits -= n;
        return (int)v < (1 << (n - 1)) ? (int)v - (1 << n) + 1 : (int)v;
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (nbits < n) grow();
        if (nbits < n) return 0;
        const uint32_t v = bits >> (32 - n);
        bits <<= n, nbits -= n;
        return (int)v;
    }
    int get_bit() { return get_bits(1); }

    void reset_entropy() {
        bits = 0, nbits = 0, nomore = false, marker = 0xff, eob_run = 0;
        for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
    }

    // ---- sequential block ---------------------------------------------------------------------------------------
    void block_sequential(int16_t data[64], Component& c) {
        const Huffman &dc = hdc[c.td], &ac = hac[c.ta];
        const uint16_t* q = quant[c.tq];
        if (nbits < 16) grow();
        const int t = decode_symbol(dc);
        if (t < 0 || t > 15) bad("bad Huffman code");
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        c.dc_pred += diff;
        data[0] = (int16_t)(c.dc_pred * q[0]);
        int k = 1;
        do {
            const int rs = decode_symbol(ac);
            if (rs < 0) bad("bad Huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break; // end of block
                k += 16;
            } else {
                k += r;
                const int zig = kZigzag[k];
                data[zig] = (int16_t)(receive_extend(s) * q[k]);
                ++k;
            }
        } while (k < 64);
    }

    // ---- progressive blocks (T.81 Annex G) ----------------------------------------------------------------------
    void block_prog_dc(int16_t data[64], Component& c) {
        if (spec_end != 0) bad("cannot merge DC and AC in a progressive scan");
        if (nbits < 16) grow();
        if (succ_high == 0) { // first pass
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode_symbol(hdc[c.td]);
            if (t < 0 || t > 15) bad("bad Huffman code");
            const int diff = t ? receive_extend(t) : 0;
            c.dc_pred += diff;
            data[0] = (int16_t)(c.dc_pred * (1 << succ_low));
        } else if (get_bit()) { // refinement
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
        }
    }
    void block_prog_ac(int16_t data[64], const Huffman& ac) {
        if (spec_start == 0) bad("cannot merge DC and AC in a progressive scan");
        if (succ_high == 0) {
            const int shift = succ_low;
            if (eob_run) {
                --eob_run;
                return;
            }
            int k = spec_start;
            do {
                const int rs = decode_symbol(ac);
                if (rs < 0) bad("bad Huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += get_bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int zig = kZigzag[k++];
                    data[zig] = (int16_t)(receive_extend(s) * (1 << shift));
                }
            } while (k <= spec_end);
        } else { // refinement scan
            const int16_t bit = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t* v) {
                if (*v != 0 && get_bit() && (*v & bit) == 0) *v = (int16_t)(*v > 0 ? *v + bit : *v - bit);
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) refine(&data[kZigzag[k]]);
            } else {
                int k = spec_start;
                do {
                    const int rs = decode_symbol(ac);
                    if (rs < 0) bad("bad Huffman code");
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += get_bits(r);
                            r = 64; // force the end of the block, refining what is left
                        }
                        // r == 15: a run of 16 zero-history coefficients
                    } else {
                        if (s != 1) bad("bad Huffman code");
                        s = get_bit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t* v = &data[kZigzag[k++]];
                        if (*v != 0) {
                            refine(v);
                        } else {
                            if (r == 0) {
                                *v = (int16_t)s;
                                break;
                            }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
    }

    // ---- inverse DCT into an 8x8 block of samples ---------------------------------------------------------------
    static uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
    static uint8_t blinn8(int x, int y) { // x * y / 255, rounded (stb_image's stbi__blinn_8x8)
        const unsigned t = (unsigned)(x * y) + 128u;
        return (uint8_t)((t + (t >> 8)) >> 8);
    }
    // 12-bit constants: (int)(c * 4096 + 0.5) on the single-precision constant, truncating — so the negative ones are
    // one step closer to zero than a symmetric rounding would put them (part of the arithmetic being matched)
    static constexpr int fix(float x) { return (int)((double)x * 4096 + 0.5); }
    static void idct(uint8_t* out, int stride, const int16_t d[64]) {
        int ws[64];
        // one 1-D pass of the LL&M factorisation; x* are the even part, t* the odd part (scaled by 4096)
        auto pass = [](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int& x0, int& x1, int& x2, int& x3, int& t0, int& t1,
                       int& t2, int& t3) {
            int p2 = s2, p3 = s6;
            int p1 = (p2 + p3) * fix(0.5411961f);
            int e2 = p1 + p3 * fix(-1.847759065f);
            int e3 = p1 + p2 * fix(0.765366865f);
            p2 = s0, p3 = s4;
            const int e0 = (p2 + p3) * 4096, e1 = (p2 - p3) * 4096;
            x0 = e0 + e3, x3 = e0 - e3, x1 = e1 + e2, x2 = e1 - e2;
            t0 = s7, t1 = s5, t2 = s3, t3 = s1;
            p3 = t0 + t2;
            int p4 = t1 + t3;
            p1 = t0 + t3, p2 = t1 + t2;
            const int p5 = (p3 + p4) * fix(1.175875602f);
            t0 = t0 * fix(0.298631336f), t1 = t1 * fix(2.053119869f), t2 = t2 * fix(3.072711026f), t3 = t3 * fix(1.501321110f);
            p1 = p5 + p1 * fix(-0.899976223f), p2 = p5 + p2 * fix(-2.562915447f);
            p3 = p3 * fix(-1.961570560f), p4 = p4 * fix(-0.390180644f);
            t3 += p1 + p4, t2 += p2 + p3, t1 += p2 + p4, t0 += p1 + p3;
        };
        for (int i = 0; i < 8; ++i) { // columns
            const int16_t* c = d + i;
            int* v = ws + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dc = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
                continue;
            }
            int x0, x1, x2, x3, t0, t1, t2, t3;
            pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512, x1 += 512, x2 += 512, x3 += 512; // keep two extra bits
            v[0] = (x0 + t3) >> 10, v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10, v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10, v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10, v[32] = (x3 - t0) >> 10;
        }
        for (int i = 0; i < 8; ++i) { // rows: remove 2^17 (12 + 2 + 3), round, add the level shift of 128
            const int* v = ws + 8 * i;
            uint8_t* o = out + (size_t)i * (size_t)stride;
            int x0, x1, x2, x3, t0, t1, t2, t3;
            pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
            const int bias = 65536 + (128 << 17);
            x0 += bias, x1 += bias, x2 += bias, x3 += bias;
            o[0] = clamp8((x0 + t3) >> 17), o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17), o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17), o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17), o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    // ---- markers ------------------------------------------------------------------------------------------------
    int next_marker() {
        if (marker != 0xff) {
            const int m = marker;
            marker = 0xff;
            return m;
        }
        int x = get8();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = get8();
        return x;
    }

    void read_table_marker(int m) {
        switch (m) {
        case 0xDD: // DRI
            if (get16() != 4) bad("bad DRI length");
            restart_interval = get16();
            return;
        case 0xDB: { // DQT
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), prec = q >> 4, t = q & 15;
                if ((prec != 0 && prec != 1) || t > 3) bad("bad DQT");
                // kept in STREAM (zigzag) order: the coefficient loops index by k, not by position
                for (int i = 0; i < 64; ++i) quant[t][i] = (uint16_t)(prec ? get16() : get8());
                len -= prec ? 129 : 65;
                quant_present[t] = true;
            }
            if (len != 0) bad("bad DQT length");
            return;
        }
        case 0xC4: { // DHT
            int len = get16() - 2;
            while (len > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) bad("bad DHT header");
                uint8_t counts[16], vals[256];
                int n = 0;
                for (int i = 0; i < 16; ++i) counts[i] = (uint8_t)get8(), n += counts[i];
                if (n > 256) bad("bad DHT");
                len -= 17;
                for (int i = 0; i < n; ++i) vals[i] = (uint8_t)get8();
                (tc == 0 ? hdc[th] : hac[th]).build(counts, vals, n);
                len -= n;
            }
            if (len != 0) bad("bad DHT length");
            return;
        }
        default:
            break;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) { // APPn / COM
            int len = get16();
            if (len < 2) bad("bad segment length");
            len -= 2;
            if (m == 0xE0 && len >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) ok &= get8() == (uint8_t)tag[i];
                len -= 5;
                if (ok) jfif = true;
            } else if (m == 0xEE && len >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) ok &= get8() == (uint8_t)tag[i];
                len -= 6;
                if (ok) {
                    get8(), get16(), get16(); // version, flags0, flags1
                    app14_transform = get8();
                    len -= 6;
                }
            }
            if (len > end - p) bad("truncated segment");
            p += len;
            return;
        }
        bad("unknown marker");
    }

    void read_frame_header() {
        const int len = get16();
        if (get8() != 8) bad("only 8-bit samples are supported");
        height = (uint32_t)get16(), width = (uint32_t)get16();
        if (width == 0 || height == 0) bad("empty image");
        ncomp = get8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) bad("bad component count");
        if (len != 8 + 3 * ncomp) bad("bad SOF length");
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = get8();
            const int q = get8();
            c.h = q >> 4, c.v = q & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) bad("bad sampling factors");
            c.tq = get8();
            if (c.tq > 3) bad("bad quantisation table index");
            hmax = std::max(hmax, c.h), vmax = std::max(vmax, c.v);
        }
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) bad("fractional sampling ratios are not supported");
        mcu_w = hmax * 8, mcu_h = vmax * 8;
        mcus_x = ((int)width + mcu_w - 1) / mcu_w, mcus_y = ((int)height + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.x = ((int)width * c.h + hmax - 1) / hmax, c.y = ((int)height * c.v + vmax - 1) / vmax;
            c.w2 = mcus_x * c.h * 8, c.h2 = mcus_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * (size_t)c.h2, 0);
            if (progressive) {
                c.bw = c.w2 / 8, c.bh = c.h2 / 8;
                c.coeff.assign((size_t)c.w2 * (size_t)c.h2, 0);
            }
        }
    }

    void read_scan_header() {
        const int len = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > ncomp) bad("bad SOS component count");
        if (len != 6 + 2 * scan_n) bad("bad SOS length");
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < ncomp; ++which)
                if (comp[which].id == id) break;
            if (which == ncomp) bad("SOS names an unknown component");
            comp[which].td = q >> 4, comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) bad("bad Huffman table index");
            order[i] = which;
        }
        spec_start = get8(), spec_end = get8();
        const int a = get8();
        succ_high = a >> 4, succ_low = a & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) bad("bad SOS");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) bad("bad SOS");
            spec_end = 63;
        }
    }

    // after each restart interval: the RSTn marker must be next; resets predictions and the bit buffer
    bool at_restart() const { return marker >= 0xD0 && marker <= 0xD7; }

    void decode_scan() {
        reset_entropy();
        for (int i = 0; i < scan_n; ++i) {
            const Component& c = comp[order[i]];
            if (!progressive || spec_start == 0) {
                if (!hdc[c.td].present && !(progressive && succ_high != 0)) bad("missing DC Huffman table");
            }
            if ((!progressive || spec_start != 0) && !hac[c.ta].present) bad("missing AC Huffman table");
            if (!progressive && !quant_present[c.tq]) bad("missing quantisation table");
        }
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto restart_check = [&]() {
            if (--todo <= 0) {
                if (nbits < 24) grow();
                if (!at_restart()) return false; // no marker: the scan simply ends (truncated data decode as zeros)
                reset_entropy();
                todo = restart_interval ? restart_interval : 0x7fffffff;
            }
            return true;
        };
        int16_t block[64];
        if (scan_n == 1) { // non-interleaved: the component's own blocks, row by row, only those covering the image
            Component& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (progressive) {
                        int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * (size_t)c.bw)];
                        if (spec_start == 0) block_prog_dc(data, c);
                        else block_prog_ac(data, hac[c.ta]);
                    } else {
                        block_sequential(block, c);
                        idct(&c.plane[(size_t)c.w2 * (size_t)j * 8 + (size_t)i * 8], c.w2, block);
                    }
                    if (!restart_check()) return;
                }
        } else { // interleaved MCUs
            for (int j = 0; j < mcus_y; ++j)
                for (int i = 0; i < mcus_x; ++i) {
                    for (int k = 0; k < scan_n; ++k) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (progressive) {
                                    block_prog_dc(&c.coeff[64 * ((size_t)bx + (size_t)by * (size_t)c.bw)], c);
                                } else {
                                    block_sequential(block, c);
                                    idct(&c.plane[(size_t)c.w2 * (size_t)by * 8 + (size_t)bx * 8], c.w2, block);
                                }
                            }
                    }
                    if (!restart_check()) return;
                }
        }
    }

    void finish_progressive() { // dequantise and transform every stored block
        for (int n = 0; n < ncomp; ++n) {
            Component& c = comp[n];
            if (!quant_present[c.tq]) bad("missing quantisation table");
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t* data = &c.coeff[64 * ((size_t)i + (size_t)j * (size_t)c.bw)];
                    for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * quant_nat[c.tq][k]);
                    idct(&c.plane[(size_t)c.w2 * (size_t)j * 8 + (size_t)i * 8], c.w2, data);
                }
        }
    }
    uint16_t quant_nat[4][64]; // the same tables indexed by coefficient POSITION (progressive pass)

    void decode() {
        if (get8() != 0xff || get8() != 0xD8) bad("not a JPEG stream");
        int m = next_marker();
        while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
            if (m == 0xff) {
                if (p >= end) bad("no frame header");
                m = next_marker();
                continue;
            }
            if ((m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) bad("unsupported coding process (lossless / arithmetic)");
            read_table_marker(m);
            m = next_marker();
        }
        progressive = m == 0xC2;
        read_frame_header();
        m = next_marker();
        bool saw_scan = false;
        while (m != 0xD9) {
            if (m == 0xDA) {
                read_scan_header();
                decode_scan();
                saw_scan = true;
                if (marker == 0xff) { // skip to the next marker (entropy data that was not consumed, or padding)
                    while (p < end) {
                        const int x = get8();
                        if (x == 0xff) {
                            const int y = get8();
                            if (y != 0 && y != 0xff) {
                                marker = (uint8_t)y;
                                break;
                            }
                            if (y == 0xff) --p;
                        }
                    }
                    if (marker == 0xff) break; // stream ended without EOI
                }
            } else if (m == 0xDC) { // DNL
                const int len = get16();
                const uint32_t lines = (uint32_t)get16();
                if (len != 4 || lines != height) bad("bad DNL");
            } else if (m == 0xff) {
                if (p >= end) break;
            } else {
                read_table_marker(m);
            }
            m = next_marker();
        }
        if (!saw_scan) bad("no scan");
        if (progressive) {
            for (int t = 0; t < 4; ++t)
                for (int k = 0; k < 64; ++k) quant_nat[t][kZigzag[k]] = quant[t][k];
            finish_progressive();
        }
    }

    // ---- upsampling + colour conversion -------------------------------------------------------------------------
    static uint8_t div4(int x) { return (uint8_t)(x >> 2); }
    static uint8_t div16(int x) { return (uint8_t)(x >> 4); }
    // each returns the row to read (either `out` or an input row)
    static const uint8_t* up_none(uint8_t*, const uint8_t* near, const uint8_t*, int, int) { return near; }
    static const uint8_t* up_v2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int) {
        for (int i = 0; i < w; ++i) out[i] = div4(3 * near[i] + far[i] + 2);
        return out;
    }
    static const uint8_t* up_h2(uint8_t* out, const uint8_t* in, const uint8_t*, int w, int) {
        if (w == 1) {
            out[0] = out[1] = in[0];
            return out;
        }
        out[0] = in[0];
        out[1] = div4(in[0] * 3 + in[1] + 2);
        int i = 1;
        for (; i < w - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[i * 2] = div4(n + in[i - 1]);
            out[i * 2 + 1] = div4(n + in[i + 1]);
        }
        out[i * 2] = div4(in[w - 2] * 3 + in[w - 1] + 2);
        out[i * 2 + 1] = in[w - 1];
        return out;
    }
    static const uint8_t* up_hv2(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int) {
        if (w == 1) {
            out[0] = out[1] = div4(3 * near[0] + far[0] + 2);
            return out;
        }
        int t1 = 3 * near[0] + far[0];
        out[0] = div4(t1 + 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
            out[i * 2] = div16(3 * t1 + t0 + 8);
        }
        out[w * 2 - 1] = div4(t1 + 2);
        return out;
    }
    static const uint8_t* up_generic(uint8_t* out, const uint8_t* near, const uint8_t*, int w, int hs) {
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
        return out;
    }

    void to_rgba(std::vector<uint8_t>& rgba) {
        rgba.assign((size_t)width * (size_t)height * 4, 255);
        struct Up {
            const uint8_t* (*fn)(uint8_t*, const uint8_t*, const uint8_t*, int, int);
            int hs, vs, w_lores, ystep, ypos;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } up[4];
        for (int k = 0; k < ncomp; ++k) {
            Up& r = up[k];
            const Component& c = comp[k];
            r.hs = hmax / c.h, r.vs = vmax / c.v;
            r.ystep = r.vs >> 1;
            r.w_lores = ((int)width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = c.plane.data();
            r.buf.assign((size_t)width + 3 + (size_t)r.hs * 2, 0);
            r.fn = r.hs == 1 && r.vs == 1 ? up_none : r.hs == 1 && r.vs == 2 ? up_v2 : r.hs == 2 && r.vs == 1 ? up_h2 : r.hs == 2 && r.vs == 2 ? up_hv2 : up_generic;
        }
        const bool is_rgb = ncomp == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (app14_transform == 0 && !jfif));
        const uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (uint32_t j = 0; j < height; ++j) {
            for (int k = 0; k < ncomp; ++k) {
                Up& r = up[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                row[k] = r.fn(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            uint8_t* out = &rgba[(size_t)j * (size_t)width * 4];
            if (ncomp == 1) {
                for (uint32_t i = 0; i < width; ++i) out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = row[0][i];
            } else if (is_rgb) {
                for (uint32_t i = 0; i < width; ++i) out[4 * i] = row[0][i], out[4 * i + 1] = row[1][i], out[4 * i + 2] = row[2][i];
            } else if (ncomp == 4 && app14_transform == 0) { // Adobe CMYK (stored inverted): channel x K / 255, rounded (stb's blinn_8x8)
                for (uint32_t i = 0; i < width; ++i) {
                    const uint8_t m = row[3][i];
                    out[4 * i] = blinn8(row[0][i], m), out[4 * i + 1] = blinn8(row[1][i], m), out[4 * i + 2] = blinn8(row[2][i], m);
                }
            } else {
                auto fixed = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
                const int kr = fixed(1.40200f), kg_r = fixed(0.71414f), kg_b = fixed(0.34414f), kb = fixed(1.77200f);
                for (uint32_t i = 0; i < width; ++i) {
                    const int yf = (row[0][i] << 20) + (1 << 19);
                    const int cb = row[1][i] - 128, cr = row[2][i] - 128;
                    const int r = (yf + cr * kr) >> 20;
                    const int g = (int)((unsigned)yf + (unsigned)(cr * -kg_r) + ((unsigned)(cb * -kg_b) & 0xffff0000u)) >> 20;
                    const int b = (yf + cb * kb) >> 20;
                    out[4 * i] = clamp8(r), out[4 * i + 1] = clamp8(g), out[4 * i + 2] = clamp8(b);
                }
                if (ncomp == 4 && app14_transform == 2) // Adobe YCCK: the converted channels inverted, times K (a fourth channel under any other
                    for (uint32_t i = 0; i < width; ++i) { // transform is ignored, as stb ignores it)
                        const uint8_t m = row[3][i];
                        out[4 * i] = blinn8(255 - out[4 * i], m), out[4 * i + 1] = blinn8(255 - out[4 * i + 1], m), out[4 * i + 2] = blinn8(255 - out[4 * i + 2], m);
                    }
            }
        }
    }
};

} // namespace

bool is_jpeg(const uint8_t* data, size_t size) { return size >= 3 && data[0] == 0xff && data[1] == 0xD8 && data[2] == 0xff; }

void decode_image_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    if (is_png(data, size)) decode_png_rgba8(data, size, width, height, rgba);
    else if (is_jpeg(data, size)) decode_jpeg_rgba8(data, size, width, height, rgba);
    else throw std::runtime_error("unknown image format (PNG and JPEG are supported)");
}

void decode_jpeg_rgba8(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba) {
    auto d = std::make_unique<Decoder>();
    d->p = data, d->end = data + size;
    std::memset(d->quant, 0, sizeof(d->quant));
    d->decode();
    if ((uint64_t)d->width * d->height > (1ull << 28)) bad("image too large");
    d->to_rgba(rgba);
    width = d->width, height = d->height;
}

} // namespace rthost
