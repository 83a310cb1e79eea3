// Device half of the split JPEG decoder (configs[4] from files; host half: csrc/jpeg_host.cpp).
//
// Input: the quantised DCT coefficients of a batch of baseline JPEGs of ONE geometry (natural order, int16, component planes
// of whole blocks, as al3d_jpeg_entropy_decode writes them) + each image's quantisation tables.  Output: 8-bit RGB
// [n][H][W][3] -- the bytes Pillow's `Image.open(...).convert("RGB")` gives, i.e. libjpeg-turbo at its defaults, restated
// from its published algorithms (libjpeg-turbo is Pillow's dependency, not part of /root/reference; the reference's call
// site is bevfusion/mmdet3d/datasets/pipelines/loading.py:19-58):
//   * jidctint.c  jpeg_idct_islow: dequantise, two passes of the 13-bit fixed-point LL&M inverse DCT (PASS1_BITS 2), +128, clamp;
//   * jdsample.c  "fancy" upsampling: h2v1 (3/4, 1/4 along x), h2v2 (the 9/16, 3/16, 3/16, 1/16 triangle), h1v2, with the
//                 edge rules of the library (first / last column special cases, the rows above the first and below the last
//                 REAL chroma row are replicas of it -- not the padding rows the blocks carry);
//   * jdcolor.c   YCbCr -> RGB with the 16-bit fixed-point tables (FIX(1.40200) etc., ONE_HALF folded into the Cb term).
// Integer arithmetic throughout: bit-identical to Pillow on every image tests/test_jpeg_gpu.py decodes.
#include "al3d_common.h"
#include "../../include/al3d.h"

#define JP_CONST_BITS 13
#define JP_PASS1_BITS 2
#define JP_FIX_0_298631336 2446
#define JP_FIX_0_390180644 3196
#define JP_FIX_0_541196100 4433
#define JP_FIX_0_765366865 6270
#define JP_FIX_0_899976223 7373
#define JP_FIX_1_175875602 9633
#define JP_FIX_1_501321110 12299
#define JP_FIX_1_847759065 15137
#define JP_FIX_1_961570560 16069
#define JP_FIX_2_053119869 16819
#define JP_FIX_2_562915447 20995
#define JP_FIX_3_072711026 25172

struct JpegGeom {
    int width, height, ncomp;
    int hs[3], vs[3];
    int bw[3], bh[3];               // blocks per row / block rows of each component plane
    int boff[3];                    // first block of each component inside an image's coefficient array
    int total_blocks;
    int max_h, max_v;
    int64_t poff[3];                // byte offset of each component's sample plane inside an image's plane buffer
    int64_t plane_bytes;
};

__device__ __forceinline__ int jp_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ unsigned char jp_clamp(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one 1-D pass of jpeg_idct_islow on eight values; SHIFT = the pass's descale amount
template <int SHIFT, int PRE>
__device__ __forceinline__ void jp_idct8(const int (&in)[8], int (&out)[8])
{
    // even part
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * JP_FIX_0_541196100;
    const int tmp2 = z1 + z3 * (-JP_FIX_1_847759065);
    const int tmp3 = z1 + z2 * JP_FIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    const int tmp0 = (z2 + z3) << PRE;
    const int tmp1 = (z2 - z3) << PRE;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    int t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * JP_FIX_1_175875602;
    t0 *= JP_FIX_0_298631336; t1 *= JP_FIX_2_053119869; t2 *= JP_FIX_3_072711026; t3 *= JP_FIX_1_501321110;
    z1 *= -JP_FIX_0_899976223; z2 *= -JP_FIX_2_562915447; z3 *= -JP_FIX_1_961570560; z4 *= -JP_FIX_0_390180644;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    out[0] = jp_descale(tmp10 + t3, SHIFT); out[7] = jp_descale(tmp10 - t3, SHIFT);
    out[1] = jp_descale(tmp11 + t2, SHIFT); out[6] = jp_descale(tmp11 - t2, SHIFT);
    out[2] = jp_descale(tmp12 + t1, SHIFT); out[5] = jp_descale(tmp12 - t1, SHIFT);
    out[3] = jp_descale(tmp13 + t0, SHIFT); out[4] = jp_descale(tmp13 - t0, SHIFT);
}

// one thread per 8 x 8 block: 64 coefficients -> 64 samples of the component's plane
__global__ __launch_bounds__(128) void jpeg_idct_kernel(const short* __restrict__ coefs, const unsigned short* __restrict__ quant,
                                                        JpegGeom g, int nimg, unsigned char* __restrict__ planes)
{
    const int64_t t = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (t >= (int64_t)nimg * g.total_blocks) return;
    const int img = (int)(t / g.total_blocks), blk = (int)(t % g.total_blocks);
    const int c = (g.ncomp > 2 && blk >= g.boff[2]) ? 2 : ((g.ncomp > 1 && blk >= g.boff[1]) ? 1 : 0);
    const int lb = blk - g.boff[c], brow = lb / g.bw[c], bcol = lb % g.bw[c];
    const short* src = coefs + t * 64;
    const unsigned short* q = quant + ((int64_t)img * 3 + c) * 64;
    int ws[64];
    // pass 1: columns (results scaled up by 2^PASS1_BITS)
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)src[r * 8 + col] * (int)q[r * 8 + col];
        jp_idct8<JP_CONST_BITS - JP_PASS1_BITS, JP_CONST_BITS>(in, out);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
    }
    // pass 2: rows; remove the scaling, divide by 8, level shift, clamp
    unsigned char* dst = planes + (int64_t)img * g.plane_bytes + g.poff[c] + ((int64_t)brow * 8) * (g.bw[c] * 8) + bcol * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int in[8], out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) in[k] = ws[r * 8 + k];
        jp_idct8<JP_CONST_BITS + JP_PASS1_BITS + 3, JP_CONST_BITS>(in, out);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (unsigned)jp_clamp(out[k] + 128) << (8 * k);
            hi |= (unsigned)jp_clamp(out[4 + k] + 128) << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + (int64_t)r * (g.bw[c] * 8)) = make_uint2(lo, hi);
    }
}

// chroma sample at full-resolution position (x, y) under libjpeg's fancy upsampling; dw x dh = the component's REAL size
__device__ __forceinline__ int jp_upsample(const unsigned char* __restrict__ pl, int pitch, int dw, int dh, int hr, int vr, int x, int y)
{
    if (hr == 1 && vr == 1) return pl[(int64_t)y * pitch + x];
    if (hr == 2 && vr == 1) {
        const unsigned char* row = pl + (int64_t)y * pitch;
        const int cx = x >> 1, v = row[cx];
        if (!(x & 1)) return cx == 0 ? v : (3 * v + row[cx - 1] + 1) >> 2;
        return cx == dw - 1 ? v : (3 * v + row[cx + 1] + 2) >> 2;
    }
    const int cy = y >> 1;
    int fy = (y & 1) ? cy + 1 : cy - 1;                       // the farther row: replicas beyond the real first / last row
    fy = fy < 0 ? 0 : (fy > dh - 1 ? dh - 1 : fy);
    const unsigned char* r0 = pl + (int64_t)cy * pitch;
    const unsigned char* r1 = pl + (int64_t)fy * pitch;
    if (hr == 1) {                                            // h1v2
        const int s = 3 * r0[x] + r1[x];
        return (s + ((y & 1) ? 2 : 1)) >> 2;
    }
    const int cx = x >> 1;                                    // h2v2
    const int cur = 3 * r0[cx] + r1[cx];
    if (!(x & 1)) {
        if (cx == 0) return (cur * 4 + 8) >> 4;
        const int last = 3 * r0[cx - 1] + r1[cx - 1];
        return (cur * 3 + last + 8) >> 4;
    }
    if (cx == dw - 1) return (cur * 4 + 7) >> 4;
    const int next = 3 * r0[cx + 1] + r1[cx + 1];
    return (cur * 3 + next + 7) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const unsigned char* __restrict__ planes, JpegGeom g, int nimg,
                                                       unsigned char* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t px = (int64_t)g.width * g.height;
    if (t >= px * nimg) return;
    const int img = (int)(t / px);
    const int x = (int)(t % g.width), y = (int)((t / g.width) % g.height);
    const unsigned char* pb = planes + (int64_t)img * g.plane_bytes;
    const int yv = pb[g.poff[0] + (int64_t)y * (g.bw[0] * 8) + x];
    unsigned char* o = out + t * 3;
    if (g.ncomp == 1) { o[0] = o[1] = o[2] = (unsigned char)yv; return; }
    int cv[2];
#pragma unroll
    for (int c = 1; c < 3; ++c) {
        const int hr = g.max_h / g.hs[c], vr = g.max_v / g.vs[c];
        const int dw = (g.width * g.hs[c] + g.max_h - 1) / g.max_h, dh = (g.height * g.vs[c] + g.max_v - 1) / g.max_v;
        cv[c - 1] = jp_upsample(pb + g.poff[c], g.bw[c] * 8, dw, dh, hr, vr, x, y);
    }
    const int cb = cv[0] - 128, cr = cv[1] - 128;
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF folded into the Cb -> G term
    const int r = yv + ((91881 * cr + 32768) >> 16);
    const int b = yv + ((116130 * cb + 32768) >> 16);
    const int gg = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    o[0] = jp_clamp(r); o[1] = jp_clamp(gg); o[2] = jp_clamp(b);
}

static int jp_geom(const int* info, JpegGeom& g)
{
    g.width = info[0]; g.height = info[1]; g.ncomp = info[2];
    AL3D_REQUIRE(g.width > 0 && g.height > 0 && (g.ncomp == 1 || g.ncomp == 3), "al3d_jpeg: bad header info");
    int64_t off = 0;
    for (int c = 0; c < 3; ++c) {
        const bool on = c < g.ncomp;
        g.hs[c] = on ? info[3 + c] : 1; g.vs[c] = on ? info[6 + c] : 1;
        g.bw[c] = on ? info[11 + c] : 0; g.bh[c] = on ? info[14 + c] : 0; g.boff[c] = on ? info[17 + c] : 0;
        g.poff[c] = off;
        off += (int64_t)g.bw[c] * g.bh[c] * 64;
        if (on) AL3D_REQUIRE(g.hs[c] >= 1 && g.hs[c] <= 2 && g.vs[c] >= 1 && g.vs[c] <= 2 && g.bw[c] > 0 && g.bh[c] > 0,
                             "al3d_jpeg: unsupported sampling factors");
    }
    g.total_blocks = info[20]; g.max_h = info[22]; g.max_v = info[23];
    g.plane_bytes = al3d_align(off, 16);
    AL3D_REQUIRE(g.total_blocks == (int)(off / 64) && g.max_h >= 1 && g.max_v >= 1, "al3d_jpeg: inconsistent header info");
    if (g.ncomp == 3) AL3D_REQUIRE(g.hs[0] == g.max_h && g.vs[0] == g.max_v, "al3d_jpeg: subsampled luma");
    return AL3D_OK;
}

extern "C" int64_t al3d_jpeg_workspace_bytes(const int* info, int nimg)
{
    if (!info || nimg <= 0) return 0;
    JpegGeom g;
    if (jp_geom(info, g) != AL3D_OK) return -1;
    return g.plane_bytes * nimg;
}

extern "C" int al3d_jpeg_idct_rgb_u8(const short* coefs, const unsigned short* quant, const int* info, int nimg,
                                     unsigned char* out_rgb, void* workspace, void* stream)
{
    AL3D_REQUIRE(info, "al3d_jpeg_idct_rgb_u8: null info");
    JpegGeom g;
    int rc = jp_geom(info, g);
    if (rc != AL3D_OK) return rc;
    if (nimg <= 0) return AL3D_OK;
    AL3D_REQUIRE(coefs && quant && out_rgb && workspace, "al3d_jpeg_idct_rgb_u8: null pointer");
    AL3D_REQUIRE(((uintptr_t)workspace & 15) == 0, "al3d_jpeg_idct_rgb_u8: 16-byte aligned workspace");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = (int64_t)nimg * g.total_blocks;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)al3d_cdiv(nb, 128)), dim3(128), 0, s, coefs, quant, g, nimg,
                       (unsigned char*)workspace);
    const int64_t npx = (int64_t)nimg * g.width * g.height;
    hipLaunchKernelGGL(jpeg_rgb_kernel, dim3((unsigned)al3d_cdiv(npx, 256)), dim3(256), 0, s, (const unsigned char*)workspace, g, nimg,
                       out_rgb);
    AL3D_CHECK_LAUNCH("jpeg_idct_rgb");
    return AL3D_OK;
}
