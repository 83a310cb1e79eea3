// Host half of the split JPEG decoder (configs[4] from files): marker parsing + Huffman entropy decoding of baseline
// sequential JPEGs into quantised DCT coefficients, on the reader pool's threads.  The device half (csrc/jpeg.hip)
// dequantises, runs the inverse DCT, upsamples the chroma planes and converts to RGB.
//
// The reference decodes camera frames with Pillow's `Image.open` inside DataLoader workers
// (bevfusion/mmdet3d/datasets/pipelines/loading.py:19-58), i.e. with libjpeg-turbo at its defaults: the accurate integer
// inverse DCT (jidctint.c), "fancy" (triangle) chroma upsampling (jdsample.c) and the 16-bit fixed-point YCbCr -> RGB tables
// (jdcolor.c).  The two halves here restate those published algorithms (libjpeg-turbo is a dependency of Pillow, not part
// of /root/reference) and are held to the installed Pillow's bytes: tests/test_jpeg_host.py (CPU: this decoder + a numpy
// restatement of the device half == PIL), tests/test_jpeg_gpu.py (device == PIL).
//
// Scope: 8-bit baseline (SOF0) Huffman, one interleaved scan, 1 or 3 components (YCbCr / grayscale), sampling factors 1 or
// 2 with the luma at the maximum, restart intervals.  Anything else (progressive, arithmetic, 12-bit, CMYK, RGB component
// ids, multiple scans) returns AL3D_EINVAL with the reason; the loader then hands that file to Pillow.
#include "al3d_common.h"
#include "../../include/al3d.h"
#include <cstring>

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    unsigned char fast_len[512];       // 9-bit prefix -> code length (0: longer than 9 bits)
    unsigned char fast_val[512];
    int mincode[18], maxcode[18], valptr[18];
    unsigned char vals[256];
};

bool build_huff(const unsigned char* counts, const unsigned char* vals, int nvals, Huff& h)
{
    int code = 0, k = 0;
    std::memset(h.fast_len, 0, sizeof(h.fast_len));
    for (int len = 1; len <= 16; ++len) {
        h.valptr[len] = k;
        h.mincode[len] = code;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (k >= nvals || code >= (1 << len)) return false;
            h.vals[k] = vals[k];
            if (len <= 9) {
                const int lo = code << (9 - len);
                for (int f = 0; f < (1 << (9 - len)); ++f) { h.fast_len[lo + f] = (unsigned char)len; h.fast_val[lo + f] = vals[k]; }
            }
        }
        h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.present = true;
    return k == nvals;
}

struct Bits {
    const unsigned char* p;
    const unsigned char* end;
    unsigned long long acc = 0;     // bits left-aligned at bit 63
    int n = 0;                      // valid bits in acc
    bool marker = false;            // a marker was reached: zeros are fed from here on (as libjpeg does)

    inline void fill()
    {
        while (n <= 56) {
            unsigned b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xff) {
                    if (p + 1 < end && p[1] == 0) p += 2;             // stuffed zero
                    else { marker = true; b = 0; }                    // RSTn / EOI / truncated: stop consuming
                } else {
                    ++p;
                }
            } else {
                marker = true;
            }
            acc |= (unsigned long long)b << (56 - n);
            n += 8;
        }
    }
    inline unsigned peek(int k) { return (unsigned)(acc >> (64 - k)); }
    inline void skip(int k) { acc <<= k; n -= k; }
    inline int receive_extend(int s)
    {
        if (s == 0) return 0;
        if (n < s) fill();
        const unsigned v = peek(s);
        skip(s);
        return (v >> (s - 1)) ? (int)v : (int)v - (1 << s) + 1;      // HUFF_EXTEND
    }
    inline int decode(const Huff& h)
    {
        if (n < 16) fill();
        const unsigned f = peek(9);
        const int l = h.fast_len[f];
        if (l) { skip(l); return h.fast_val[f]; }
        int code = (int)peek(10), len = 10;
        while (len <= 16 && code > h.maxcode[len]) { ++len; code = (int)peek(len); }
        if (len > 16) return -1;
        skip(len);
        return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    // after an MCU row / restart interval: drop the partial byte, expect RSTn
    bool restart(int expect)
    {
        acc = 0; n = 0;
        // the reader stopped AT the 0xff of the marker (marker == true) or must skip fill bytes up to it
        while (p + 1 < end && !(p[0] == 0xff && p[1] != 0 && p[1] != 0xff)) ++p;
        if (p + 1 >= end) return false;
        if (p[1] != 0xd0 + (expect & 7)) return false;
        p += 2;
        marker = false;
        return true;
    }
};

struct Parsed {
    int width = 0, height = 0, ncomp = 0;
    int cid[3], hs[3], vs[3], tq[3], td[3], ta[3];
    unsigned short quant[4][64];       // natural order
    bool qpresent[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart_interval = 0;
    const unsigned char* scan = nullptr;
};

inline int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

int parse(const unsigned char* d, int64_t n, Parsed& P)
{
    if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return al3d_fail(AL3D_EINVAL, "jpeg: no SOI marker");
    int64_t i = 2;
    bool sof = false;
    while (i + 4 <= n) {
        if (d[i] != 0xff) return al3d_fail(AL3D_EINVAL, "jpeg: marker expected at byte %lld", (long long)i);
        while (i < n && d[i] == 0xff) ++i;                              // fill bytes
        if (i >= n) break;
        const int m = d[i++];
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
        if (m == 0xd9) return al3d_fail(AL3D_EINVAL, "jpeg: EOI before a scan");
        if (i + 2 > n) break;
        const int len = be16(d + i);
        if (len < 2 || i + len > n) return al3d_fail(AL3D_EINVAL, "jpeg: truncated segment");
        const unsigned char* s = d + i + 2;
        const int sl = len - 2;
        if (m == 0xc0) {
            if (sl < 6) return al3d_fail(AL3D_EINVAL, "jpeg: short SOF");
            if (s[0] != 8) return al3d_fail(AL3D_EINVAL, "jpeg: %d-bit samples (8-bit only)", s[0]);
            P.height = be16(s + 1); P.width = be16(s + 3); P.ncomp = s[5];
            if (P.ncomp != 1 && P.ncomp != 3) return al3d_fail(AL3D_EINVAL, "jpeg: %d components (1 or 3 only)", P.ncomp);
            if (sl < 6 + 3 * P.ncomp || P.width <= 0 || P.height <= 0) return al3d_fail(AL3D_EINVAL, "jpeg: bad SOF");
            for (int c = 0; c < P.ncomp; ++c) {
                P.cid[c] = s[6 + 3 * c]; P.hs[c] = s[7 + 3 * c] >> 4; P.vs[c] = s[7 + 3 * c] & 15; P.tq[c] = s[8 + 3 * c];
                if (P.tq[c] > 3) return al3d_fail(AL3D_EINVAL, "jpeg: bad quantisation table index");
            }
            sof = true;
        } else if (m >= 0xc1 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return al3d_fail(AL3D_EINVAL, "jpeg: SOF%d (only baseline SOF0 is decoded here)", m - 0xc0);
        } else if (m == 0xcc) {
            return al3d_fail(AL3D_EINVAL, "jpeg: arithmetic coding");
        } else if (m == 0xdb) {
            int k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || (pq != 0 && pq != 1) || k + 64 * (pq + 1) > sl) return al3d_fail(AL3D_EINVAL, "jpeg: bad DQT");
                for (int z = 0; z < 64; ++z) {
                    const int v = pq ? be16(s + k + 2 * z) : s[k + z];
                    P.quant[tq][kZigzag[z]] = (unsigned short)v;
                }
                P.qpresent[tq] = true;
                k += 64 * (pq + 1);
            }
        } else if (m == 0xc4) {
            int k = 0;
            while (k < sl) {
                if (k + 17 > sl) return al3d_fail(AL3D_EINVAL, "jpeg: bad DHT");
                const int tc = s[k] >> 4, th = s[k] & 15;
                int nv = 0;
                for (int l = 0; l < 16; ++l) nv += s[k + 1 + l];
                if (tc > 1 || th > 3 || nv > 256 || k + 17 + nv > sl) return al3d_fail(AL3D_EINVAL, "jpeg: bad DHT");
                if (!build_huff(s + k + 1, s + k + 17, nv, tc ? P.ac[th] : P.dc[th]))
                    return al3d_fail(AL3D_EINVAL, "jpeg: inconsistent Huffman table");
                k += 17 + nv;
            }
        } else if (m == 0xdd) {
            if (sl < 2) return al3d_fail(AL3D_EINVAL, "jpeg: bad DRI");
            P.restart_interval = be16(s);
        } else if (m == 0xee) {
            if (sl >= 12 && std::memcmp(s, "Adobe", 5) == 0 && P.ncomp != 1 && s[11] != 1)
                return al3d_fail(AL3D_EINVAL, "jpeg: Adobe colour transform %d (YCbCr only)", s[11]);
        } else if (m == 0xda) {
            if (!sof) return al3d_fail(AL3D_EINVAL, "jpeg: scan before the frame header");
            if (sl < 1 || s[0] != P.ncomp || sl < 1 + 2 * P.ncomp + 3)
                return al3d_fail(AL3D_EINVAL, "jpeg: a scan with %d of %d components (one interleaved scan only)", sl ? s[0] : 0, P.ncomp);
            for (int c = 0; c < P.ncomp; ++c) {
                if (s[1 + 2 * c] != P.cid[c]) return al3d_fail(AL3D_EINVAL, "jpeg: scan component order");
                P.td[c] = s[2 + 2 * c] >> 4; P.ta[c] = s[2 + 2 * c] & 15;
                if (P.td[c] > 3 || P.ta[c] > 3 || !P.dc[P.td[c]].present || !P.ac[P.ta[c]].present || !P.qpresent[P.tq[c]])
                    return al3d_fail(AL3D_EINVAL, "jpeg: a table the scan names is missing");
            }
            const unsigned char* t = s + 1 + 2 * P.ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return al3d_fail(AL3D_EINVAL, "jpeg: not a sequential scan");
            P.scan = d + i + len;
            break;
        }
        i += len;
    }
    if (!P.scan) return al3d_fail(AL3D_EINVAL, "jpeg: no scan found");
    int mh = 0, mv = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        if (P.hs[c] < 1 || P.hs[c] > 2 || P.vs[c] < 1 || P.vs[c] > 2)
            return al3d_fail(AL3D_EINVAL, "jpeg: sampling factor %d x %d (1 or 2 only)", P.hs[c], P.vs[c]);
        mh = P.hs[c] > mh ? P.hs[c] : mh; mv = P.vs[c] > mv ? P.vs[c] : mv;
    }
    if (P.ncomp == 3) {
        if (P.hs[0] != mh || P.vs[0] != mv) return al3d_fail(AL3D_EINVAL, "jpeg: subsampled luma");
        if (P.cid[0] == 'R' && P.cid[1] == 'G' && P.cid[2] == 'B') return al3d_fail(AL3D_EINVAL, "jpeg: RGB component ids");
    }
    return AL3D_OK;
}

void fill_info(const Parsed& P, int* info, unsigned short* quant)
{
    int mh = 1, mv = 1;
    for (int c = 0; c < P.ncomp; ++c) { mh = P.hs[c] > mh ? P.hs[c] : mh; mv = P.vs[c] > mv ? P.vs[c] : mv; }
    // a single-component scan is never interleaved: its MCU is one block whatever the sampling factors say
    const bool one = P.ncomp == 1;
    const int mcu_w = one ? 8 : 8 * mh, mcu_h = one ? 8 : 8 * mv;
    const int mx = (P.width + mcu_w - 1) / mcu_w, my = (P.height + mcu_h - 1) / mcu_h;
    std::memset(info, 0, sizeof(int) * AL3D_JPEG_INFO_INTS);
    info[0] = P.width; info[1] = P.height; info[2] = P.ncomp;
    int off = 0;
    for (int c = 0; c < P.ncomp; ++c) {
        const int h = one ? 1 : P.hs[c], v = one ? 1 : P.vs[c];
        info[3 + c] = h; info[6 + c] = v;
        info[11 + c] = mx * h; info[14 + c] = my * v;
        info[17 + c] = off;
        off += mx * h * my * v;
        for (int z = 0; z < 64; ++z) quant[64 * c + z] = P.quant[P.tq[c]][z];
    }
    info[9] = mx; info[10] = my; info[20] = off; info[21] = P.restart_interval; info[22] = one ? 1 : mh; info[23] = one ? 1 : mv;
}

}  // namespace

extern "C" int al3d_jpeg_header(const unsigned char* data, int64_t nbytes, int* info, unsigned short* quant)
{
    AL3D_REQUIRE(data && info && quant && nbytes > 0, "al3d_jpeg_header: null pointer");
    Parsed P;
    const int rc = parse(data, nbytes, P);
    if (rc != AL3D_OK) return rc;
    fill_info(P, info, quant);
    return AL3D_OK;
}

extern "C" int al3d_jpeg_entropy_decode(const unsigned char* data, int64_t nbytes, short* coefs, int64_t coef_blocks)
{
    AL3D_REQUIRE(data && coefs && nbytes > 0, "al3d_jpeg_entropy_decode: null pointer");
    Parsed P;
    int rc = parse(data, nbytes, P);
    if (rc != AL3D_OK) return rc;
    int info[AL3D_JPEG_INFO_INTS];
    unsigned short q[192];
    fill_info(P, info, q);
    AL3D_REQUIRE(coef_blocks >= info[20], "al3d_jpeg_entropy_decode: coefficient buffer of %lld blocks, the image needs %d",
                 (long long)coef_blocks, info[20]);
    std::memset(coefs, 0, (size_t)info[20] * 128);
    Bits B;
    B.p = P.scan; B.end = data + nbytes;
    int pred[3] = {0, 0, 0};
    const int mx = info[9], my = info[10], ri = P.restart_interval;
    int left = ri, rst = 0;
    for (int y = 0; y < my; ++y) {
        for (int x = 0; x < mx; ++x) {
            if (ri && left == 0) {
                if (!B.restart(rst)) return al3d_fail(AL3D_EINVAL, "jpeg: restart marker RST%d not found", rst & 7);
                rst = (rst + 1) & 7; left = ri;
                pred[0] = pred[1] = pred[2] = 0;
            }
            for (int c = 0; c < P.ncomp; ++c) {
                const int h = info[3 + c], v = info[6 + c], bw = info[11 + c];
                const Huff& hd = P.dc[P.td[c]];
                const Huff& ha = P.ac[P.ta[c]];
                for (int by = 0; by < v; ++by) {
                    for (int bx = 0; bx < h; ++bx) {
                        short* blk = coefs + ((int64_t)info[17 + c] + (int64_t)(y * v + by) * bw + (x * h + bx)) * 64;
                        int s = B.decode(hd);
                        if (s < 0 || s > 11) return al3d_fail(AL3D_EINVAL, "jpeg: corrupt DC code");
                        pred[c] += B.receive_extend(s);
                        blk[0] = (short)pred[c];
                        for (int k = 1; k < 64;) {
                            const int rs = B.decode(ha);
                            if (rs < 0) return al3d_fail(AL3D_EINVAL, "jpeg: corrupt AC code");
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;                            // EOB
                                k += 16;                                       // ZRL
                                continue;
                            }
                            k += r;
                            if (k > 63) return al3d_fail(AL3D_EINVAL, "jpeg: coefficient index out of range");
                            blk[kZigzag[k]] = (short)B.receive_extend(s);
                            ++k;
                        }
                    }
                }
            }
            if (ri) --left;
        }
    }
    return AL3D_OK;
}
