// plan.hip -- static inference schedule executor for SMAP.forward (model/smap.py:403-419)
// plus the non-GEMM kernels of the backbone:
//   stem_kernel     ResNet_top conv 7x7 s2 p3 + folded BN + ReLU   (smap.py:83-84, 88-90)
//   maxpool_kernel  MaxPool2d(3, 2, 1)                              (smap.py:86)
//   upadd_kernel    relu(a + bilinear_align_corners(t))             (smap.py:213-217)
//   headsum_kernel  res4 + up(res3) + up(res2) -> fp32 NCHW         (smap.py:221-229, 417-419)
// The schedule itself (which op, which buffers) is built by the Python host
// (smap_amd/plan.py) and handed over as a flat array of smap_op; this file only
// launches it, op after op, on one HIP stream -- capturable into a hipGraph.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <vector>
#include <type_traits>
#include "smap_hip.h"
#include "plan.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : -(1000 + (int)e); }

// ------------------------------------------------------------------ stem --
// ResNet_top conv 7x7 s2 p3, 3 -> 64, + folded BN + ReLU, fp32 NCHW image -> NHWC fp16.
// Implicit GEMM on the matrix cores with the im2col done by the LDS read addresses:
//   K order = (kh, c, kw padded 7->8): one 8-wide K granule = 8 consecutive input columns of one
//   (kh, c) row of the patch, i.e. 16 contiguous bytes of the fp16 patch in LDS; 21 granules + 1
//   zero granule = 22 = 11 MFMA K-steps of 16.  Weights (A operand, [64][176] fp16) and the
//   37x38x3 input patch (B operand) both sit in LDS; D[n][pixel] puts 4 consecutive channels of
//   one pixel in a lane, so the NHWC store is 8 bytes per lane with no transposition.
// One 16x16 output tile per workgroup, 64 pixels x 64 channels per wave (4 accumulators).
#ifndef SMAP_STEM_STORE16
#define SMAP_STEM_STORE16 1
#endif
constexpr int ST_T = 16, ST_PH = ST_T * 2 + 5, ST_PW = 40;     // tile edge, patch rows, padded patch row (halves)
constexpr int ST_K = 176;                                      // 22 granules x 8
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// The images of a schedule may come as up to SMAP_MAX_INPUTS separate [frames_per,3,H,W] buffers (smap_plan_run_inputs: a
// launch that coalesces several of the caller's batches reads them where they are -- no gather copy in front of the stem).
struct StemIn {
    const float* p[SMAP_MAX_INPUTS];
    int frames_per;                     // frames per buffer; frame b lives in p[b / frames_per] at index b % frames_per
};
__device__ __forceinline__ const float* stem_frame(const StemIn& in, int b, int H, int W)
{
    const int k = b / in.frames_per;
    const float* base = in.p[0];        // compile-time indices only: a run-time index into a by-value kernel argument would
#pragma unroll                          // be promoted to scratch / LDS (EXPERIMENTS R3.6)
    for (int j = 1; j < SMAP_MAX_INPUTS; ++j) base = k == j ? in.p[j] : base;
    return base + (size_t)(b - k * in.frames_per) * 3 * H * W;
}

// X3 (smap_op.precision = 1): image and weights as fp16 hi/lo pairs, three MFMAs per K step, output pixel = [hi(64) | lo(64)]
// (see conv.hip).  The weight blob then holds [64][176] hi followed by [64][176] lo of (w * 2^s); acc_scale = 2^-s.
template <bool X3>
__global__ __launch_bounds__(256) void stem_kernel(const StemIn in, const _Float16* __restrict__ wk,
                                                   const float* __restrict__ bias, _Float16* __restrict__ out,
                                                   int H, int W, int Ho, int Wo, float acc_scale, int flip_from)
{
    constexpr int NPL = X3 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) _Float16 s_w[NPL * 64 * ST_K];
    __shared__ __attribute__((aligned(16))) _Float16 s_p[NPL * 3 * ST_PH * ST_PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, oy0 = blockIdx.y * ST_T, ox0 = blockIdx.x * ST_T;
    for (int i = tid; i < NPL * 64 * ST_K / 8; i += 256)
        reinterpret_cast<half8*>(s_w)[i] = reinterpret_cast<const half8*>(wk)[i];
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    // flip-TTA: frames >= flip_from are the x-mirrored images of frames 0.. (the mirror lives in this index, not in memory)
    const bool mirror = flip_from > 0 && b >= flip_from;
    const int bsrc = mirror ? b - flip_from : b;
    const float* __restrict__ img = stem_frame(in, bsrc, H, W);
    for (int i = tid; i < 3 * ST_PH * ST_PW; i += 256) {
        const int c = i / (ST_PH * ST_PW), r = i - c * ST_PH * ST_PW;
        const int py = r / ST_PW, px = r - py * ST_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = img[((size_t)c * H + iy) * W + (mirror ? W - 1 - ix : ix)];
        const _Float16 hi = (_Float16)v;
        s_p[i] = hi;
        if (X3) s_p[3 * ST_PH * ST_PW + i] = (_Float16)(v - (float)hi);
    }
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[2][2];                                          // [channel tile][pixel tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // this lane's two pixels (B operand columns): pixel tile t covers output rows 4*wave+2t, +1
    int pbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int py = 4 * wave + 2 * t + (l31 >> 4), px = l31 & 15;
        pbase[t] = (py * 2) * ST_PW + px * 2;                  // halves; + (c*ST_PH + kh)*ST_PW per granule
    }
#pragma unroll
    for (int ks = 0; ks < ST_K / 16; ++ks) {
        int g = ks * 2 + lhi;                                  // K granule of this half-wave
        const int gg = g < 21 ? g : 20;                        // granule 21 has zero weights: any in-bounds read
        const int kh = gg / 3, c = gg - kh * 3;
        const int goff = (c * ST_PH + kh) * ST_PW;
        half8 wf[NPL][2], pf[NPL][2];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[pl][nt] = *reinterpret_cast<const half8*>(s_w + pl * 64 * ST_K + (nt * 32 + l31) * ST_K + g * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t) {                      // 8 consecutive columns, 4-byte aligned: 4 x b32
                const unsigned* q = reinterpret_cast<const unsigned*>(s_p + pl * 3 * ST_PH * ST_PW + pbase[t] + goff);
                union { unsigned u[4]; half8 h; } cv;
                cv.u[0] = q[0]; cv.u[1] = q[1]; cv.u[2] = q[2]; cv.u[3] = q[3];
                pf[pl][t] = cv.h;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (X3) {
                    acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[NPL - 1][nt], pf[0][t], acc[nt][t], 0, 0, 0);
                    acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nt], pf[NPL - 1][t], acc[nt][t], 0, 0, 0);
                }
                acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nt], pf[0][t], acc[nt][t], 0, 0, 0);
            }
    }
    // D[n][pixel]: lane = pixel (col l31), reg r -> channel nt*32 + (r&3) + 8*(r>>2) + 4*lhi.  The half-wave swap of convp.hip's
    // register epilogue turns a lane's 4 + 4 channels (8 apart) into 8 consecutive ones: 16-byte NHWC stores instead of 8-byte ones
    // (SMAP_STEM_STORE16 = 0: the round-1 form).
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int oy = oy0 + 4 * wave + 2 * t + (l31 >> 4), ox = ox0 + (l31 & 15);
        const bool ok = oy < Ho && ox < Wo;
        _Float16* op = out + (((size_t)b * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0)) * (NPL * 64);
#if SMAP_STEM_STORE16
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xf = acc[nt][t][8 * j + e], yf = acc[nt][t][8 * j + 4 + e];    // channels 16j + e + 4 lhi and + 8
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    v[e] = __uint_as_float(s0);
                    v[4 + e] = __uint_as_float(s1);
                }
                const int n0 = nt * 32 + 16 * j + 8 * lhi;     // this lane now holds channels n0 .. n0 + 7 of its pixel
                const float4 b0 = *reinterpret_cast<const float4*>(bias + n0), b1 = *reinterpret_cast<const float4*>(bias + n0 + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                half8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = X3 ? v[e] * acc_scale + bb[e] : v[e] + bb[e];
                    x = x < 0.f ? 0.f : x;                     // NaN stays NaN (torch's ReLU)
                    h[e] = (_Float16)x;
                    l[e] = (_Float16)(x - (float)h[e]);
                }
                if (!ok) continue;
                *reinterpret_cast<half8*>(op + n0) = h;
                if (X3) *reinterpret_cast<half8*>(op + 64 + n0) = l;
            }
#else
        if (!ok) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = nt * 32 + 8 * q + 4 * lhi;
                const float4 bv = *reinterpret_cast<const float4*>(bias + n0);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = X3 ? acc[nt][t][4 * q + e] * acc_scale + bb[e] : acc[nt][t][4 * q + e] + bb[e];
                    v = v < 0.f ? 0.f : v;                 // NaN stays NaN (torch's ReLU)
                    h[e] = (_Float16)v;
                    l[e] = (_Float16)(v - (float)h[e]);
                }
                *reinterpret_cast<half4*>(op + n0) = h;
                if (X3) *reinterpret_cast<half4*>(op + 64 + n0) = l;
            }
#endif
    }
}

// ------------------------------------------------------- stem + max-pool --
// ResNet_top in one kernel (smap.py:83-86): the 7x7 s2 conv + BN + ReLU of stem_kernel, then the 3x3 s2 p1 max-pool
// straight from LDS.  stem_kernel is bound by its 109 MB (B = 8) output write and maxpool_kernel reads that tensor
// back; here only the pooled 27 MB leave the CU.  A workgroup owns 8 x 7 pooled pixels = 17 x 15 conv pixels (255 of
// the 256 MFMA pixel columns; the one-pixel halo is recomputed: +14 % conv work on a store-bound kernel).  Conv pixels
// outside the image take the value 0 in the LDS tile: every real value is >= 0 after the ReLU, so a 0 never changes a
// window maximum -- the same result as MaxPool2d's -inf padding.
constexpr int SP_PY = 8, SP_PX = 7, SP_SY = 2 * SP_PY + 1, SP_SX = 2 * SP_PX + 1;       // pooled tile, conv tile (17 x 15)
constexpr int SP_PH = (SP_SY - 1) * 2 + 7, SP_PW = 40;                                  // input patch rows (39), padded row
static_assert(SP_SY * SP_SX <= 256 && (SP_SX - 1) * 2 + 8 <= SP_PW, "stem+pool tile");

// X3 / flip_from: as in stem_kernel (hi/lo planes, three MFMAs per K step, mirrored read of frames >= flip_from); the conv
// tile then stays in LDS as fp32 (the maximum is taken over the exact value, then re-split).
template <bool X3>
__global__ __launch_bounds__(256) void stem_pool_kernel(const StemIn in, const _Float16* __restrict__ wk,
                                                        const float* __restrict__ bias, _Float16* __restrict__ out,
                                                        int H, int W, int Hs, int Ws, int Ho, int Wo, float acc_scale, int flip_from)
{
    constexpr int NPL = X3 ? 2 : 1;
    typedef typename std::conditional<X3, float, _Float16>::type tile_t;
    __shared__ __attribute__((aligned(16))) _Float16 s_w[NPL * 64 * ST_K];
    __shared__ __attribute__((aligned(16))) _Float16 s_p[NPL * 3 * SP_PH * SP_PW];
    __shared__ __attribute__((aligned(16))) tile_t s_t[256 * 64];                       // conv tile, [pixel slot][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, py0 = blockIdx.y * SP_PY, px0 = blockIdx.x * SP_PX;
    const int sy0 = py0 * 2 - 1, sx0 = px0 * 2 - 1;                                     // first conv pixel of the tile
    for (int i = tid; i < NPL * 64 * ST_K / 8; i += 256)
        reinterpret_cast<half8*>(s_w)[i] = reinterpret_cast<const half8*>(wk)[i];
    const int iy0 = sy0 * 2 - 3, ix0 = sx0 * 2 - 3;
    const bool mirror = flip_from > 0 && b >= flip_from;
    const int bsrc = mirror ? b - flip_from : b;
    const float* __restrict__ img = stem_frame(in, bsrc, H, W);
    for (int i = tid; i < 3 * SP_PH * SP_PW; i += 256) {
        const int c = i / (SP_PH * SP_PW), r = i - c * SP_PH * SP_PW;
        const int py = r / SP_PW, px = r - py * SP_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = img[((size_t)c * H + iy) * W + (mirror ? W - 1 - ix : ix)];
        const _Float16 hi = (_Float16)v;
        s_p[i] = hi;
        if (X3) s_p[3 * SP_PH * SP_PW + i] = (_Float16)(v - (float)hi);
    }
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[2][2];                                          // [channel tile][pixel tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // pixel slot s = wave*64 + t*32 + l31 -> conv pixel (s / 15, s % 15) of the 17 x 15 tile (slot 255 is a spare)
    int pbase[2], slot[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int sl = wave * 64 + t * 32 + l31;
        const int sc = sl < SP_SY * SP_SX ? sl : 0;
        const int py = sc / SP_SX, px = sc - py * SP_SX;
        slot[t] = sl;
        pbase[t] = (py * 2) * SP_PW + px * 2;
    }
#pragma unroll
    for (int ks = 0; ks < ST_K / 16; ++ks) {
        int g = ks * 2 + lhi;
        const int gg = g < 21 ? g : 20;
        const int kh = gg / 3, c = gg - kh * 3;
        const int goff = (c * SP_PH + kh) * SP_PW;
        half8 wf[NPL][2], pf[NPL][2];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[pl][nt] = *reinterpret_cast<const half8*>(s_w + pl * 64 * ST_K + (nt * 32 + l31) * ST_K + g * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned* q = reinterpret_cast<const unsigned*>(s_p + pl * 3 * SP_PH * SP_PW + pbase[t] + goff);
                union { unsigned u[4]; half8 h; } cv;
                cv.u[0] = q[0]; cv.u[1] = q[1]; cv.u[2] = q[2]; cv.u[3] = q[3];
                pf[pl][t] = cv.h;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (X3) {
                    acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[NPL - 1][nt], pf[0][t], acc[nt][t], 0, 0, 0);
                    acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nt], pf[NPL - 1][t], acc[nt][t], 0, 0, 0);
                }
                acc[nt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nt], pf[0][t], acc[nt][t], 0, 0, 0);
            }
    }
    // bias + ReLU -> conv tile in LDS (zero where the conv pixel lies outside the image)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int sl = slot[t];
        const int py = sl / SP_SX, px = sl - py * SP_SX;
        const int sy = sy0 + py, sx = sx0 + px;
        const bool live = sl < SP_SY * SP_SX && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = nt * 32 + 8 * q + 4 * lhi;
                const float4 bv = *reinterpret_cast<const float4*>(bias + n0);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = X3 ? acc[nt][t][4 * q + e] * acc_scale + bb[e] : acc[nt][t][4 * q + e] + bb[e];
                    v = (!live || v < 0.f) ? 0.f : v;          // NaN stays NaN
                    s_t[sl * 64 + n0 + e] = (tile_t)v;
                }
            }
    }
    __syncthreads();
    // 3x3 s2 max over the conv tile: pooled pixel (py, px) reads conv rows 2py..2py+2, cols 2px..2px+2 of the tile
    for (int i = tid; i < SP_PY * SP_PX * 8; i += 256) {
        const int cg = i & 7, pp = i >> 3;
        const int py = pp / SP_PX, px = pp - py * SP_PX;
        const int oy = py0 + py, ox = px0 + px;
        if (oy >= Ho || ox >= Wo) continue;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (float)s_t[((2 * py) * SP_SX + 2 * px) * 64 + cg * 8 + e];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (dy == 0 && dx == 0) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = (float)s_t[((2 * py + dy) * SP_SX + 2 * px + dx) * 64 + cg * 8 + e];
                    m[e] = (x > m[e] || x != x) ? x : m[e];    // ATen's rule: NaN sticks
                }
            }
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)m[e];
        _Float16* op = out + (((size_t)b * Ho + oy) * Wo + ox) * (NPL * 64) + cg * 8;
        *reinterpret_cast<half8*>(op) = h;
        if (X3) {
            half8 l;
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = (_Float16)(m[e] - (float)h[e]);
            *reinterpret_cast<half8*>(op + 64) = l;
        }
    }
}

// --------------------------------------------------------------- maxpool --
// NHWC fp16, 3x3 stride 2 pad 1 (padding never wins: only in-range taps are read).
// X3: pixel = [hi(C) | lo(C)]; the maximum is taken over hi + lo (exact in fp32) and re-split (the same pair comes back).
template <bool X3>
__global__ void maxpool_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int B, int H, int W,
                               int C, int Ho, int Wo)
{
    constexpr int NPL = X3 ? 2 : 1;
    const int cg_n = C / 8;
    const long long total = (long long)B * Ho * Wo * cg_n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cg_n);
        long long pix = idx / cg_n;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), b = (int)(pix / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const _Float16* ip = in + (((size_t)b * H + iy) * W + ix) * (NPL * C) + cg * 8;
                const half8 v = *reinterpret_cast<const half8*>(ip);
                if (X3) {
                    const half8 vl = *reinterpret_cast<const half8*>(ip + C);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float x = (float)v[e] + (float)vl[e]; m[e] = (x > m[e] || x != x) ? x : m[e]; }   // ATen's rule: NaN sticks
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float x = (float)v[e]; m[e] = (x > m[e] || x != x) ? x : m[e]; }
                }
            }
        }
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)m[e];
        _Float16* op = out + (((size_t)b * Ho + oy) * Wo + ox) * (NPL * C) + cg * 8;
        *reinterpret_cast<half8*>(op) = h;
        if (X3) {
            half8 l;
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = (_Float16)(m[e] - (float)h[e]);
            *reinterpret_cast<half8*>(op + C) = l;
        }
    }
}

// bilinear align-corners index rule: lerp_index() in plan.h (shared with the conv epilogue)

// out = relu(a + up(t)); a/out [B,Ho,Wo,C] fp16, t [B,h,w,C] fp16; 8 channels per thread.
__global__ void upadd_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ t,
                             _Float16* __restrict__ out, int B, int Ho, int Wo, int C, int h, int w, int relu)
{
    const int cg_n = C / 8;
    const long long total = (long long)B * Ho * Wo * cg_n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % cg_n);
        long long pix = idx / cg_n;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), b = (int)(pix / Ho);
        const Lerp ly = lerp_index(oy, h, Ho), lx = lerp_index(ox, w, Wo);
        const _Float16* tb = t + (size_t)b * h * w * C + cg * 8;
        const half8 v00 = *reinterpret_cast<const half8*>(tb + ((size_t)ly.i0 * w + lx.i0) * C);
        const half8 v01 = *reinterpret_cast<const half8*>(tb + ((size_t)ly.i0 * w + lx.i1) * C);
        const half8 v10 = *reinterpret_cast<const half8*>(tb + ((size_t)ly.i1 * w + lx.i0) * C);
        const half8 v11 = *reinterpret_cast<const half8*>(tb + ((size_t)ly.i1 * w + lx.i1) * C);
        const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + cg * 8;
        const half8 av = *reinterpret_cast<const half8*>(a + o);
        half8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float up = ly.l0 * (lx.l0 * (float)v00[e] + lx.l1 * (float)v01[e]) +
                             ly.l1 * (lx.l0 * (float)v10[e] + lx.l1 * (float)v11[e]);
            float v = (float)av[e] + up;
            if (relu) v = v < 0.f ? 0.f : v;
            r[e] = (_Float16)v;
        }
        *reinterpret_cast<half8*>(out + o) = r;
    }
}

// fp32 NHWC heads (channel stride Cs) -> fp32 NCHW [B,C,Ho,Wo] = ((s0 + up(s1)) + up(s2)).
// One 32-pixel row segment per workgroup; LDS transposes pixel-major -> channel-major so
// that both the NHWC reads and the NCHW writes are coalesced.
struct HeadSrc { const float* p[3]; int h[3], w[3]; int n; };
constexpr int HS_PX = 32;

// ((s0 + up(s1)) + up(s2)) for four consecutive channels of pixel (b, y, x): the value the reference's
// `res4 + res3 + res2` (smap.py:417) has there.  A source at the output resolution is its own bilinear image.
// Written with the source index k as a COMPILE-TIME constant (unrolled, `k < s.n` as a predicate).  The round-1/2 version looped
// over k with a private `Lerp ly[3]`; the compiler promoted that array to LDS (indexed through the dispatch packet) and the kernel
// returned ~15 wrong values in 10-30 % of the launches that overlapped another stream's kernels -- never alone (EXPERIMENTS
// R3.6, tools/experiments/headsum_runtime_index_repro.hip).  No run-time-indexed private arrays in this library's kernels.
template <int K>
__device__ __forceinline__ float4 head_source4(const HeadSrc& s, const Lerp& ly, int b, int y, int x, int c, int Ho, int Wo, int Cs)
{
    const int sh = s.h[K], sw = s.w[K];
    const float* base = s.p[K] + (size_t)b * sh * sw * Cs + c;
    if (sh == Ho && sw == Wo) return *reinterpret_cast<const float4*>(base + ((size_t)y * Wo + x) * Cs);
    const Lerp lx = lerp_index(x, sw, Wo);
    const float4 v00 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * sw + lx.i0) * Cs);
    const float4 v01 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i0 * sw + lx.i1) * Cs);
    const float4 v10 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * sw + lx.i0) * Cs);
    const float4 v11 = *reinterpret_cast<const float4*>(base + ((size_t)ly.i1 * sw + lx.i1) * Cs);
    float4 up;
    up.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
    up.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
    up.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
    up.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    return up;
}

__device__ __forceinline__ float4 head_value4(const HeadSrc& s, const Lerp& ly0, const Lerp& ly1, const Lerp& ly2, int b, int y, int x,
                                              int c, int Ho, int Wo, int Cs)
{
    float4 v = head_source4<0>(s, ly0, b, y, x, c, Ho, Wo, Cs);
    if (s.n > 1) {
        const float4 u = head_source4<1>(s, ly1, b, y, x, c, Ho, Wo, Cs);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (s.n > 2) {
        const float4 u = head_source4<2>(s, ly2, b, y, x, c, Ho, Wo, Cs);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    return v;
}

// fp32 NHWC heads (channel stride Cs) -> fp32 NCHW [B,C,Ho,Wo] = ((s0 + up(s1)) + up(s2)).
// One 32-pixel row segment per workgroup; LDS transposes pixel-major -> channel-major so
// that both the NHWC reads and the NCHW writes are coalesced.
// FLIP (smap_op.flip_from > 0): the flip-TTA merge of test.py:55-70 in the same pass -- the workgroup also evaluates the
// maps of the mirrored frame b + flip_from at the mirrored pixels W-1-x and writes
//     out[b,c] = v[b,c] + s_c * v_mirror[pair[c]]   (s_c = -1 on PAF-x channels), halved for c >= n_kpt,
// the same fp32 operations, in the same order, as the reference's channel loop (and as smap_flip_merge).
// scale (smap_op.scale_hms): the value is stored as v / 255 (channels < n_kpt) or v / 127 -- test.py:111-112's in-place division of the maps
// before the association, the same fp32 IEEE division as smap_scale_hms (csrc/assoc.hip), fused into this store.
template <bool FLIP>
__global__ __launch_bounds__(256) void headsum_kernel(HeadSrc s, float* __restrict__ out, int Ho, int Wo, int C,
                                                      int Cs, int flip_from, const int* __restrict__ pair, int n_kpt,
                                                      int* __restrict__ status, int scale)
{
    __shared__ float tile[(FLIP ? 2 : 1) * 48 * (HS_PX + 1)];
    float* tile2 = tile + 48 * (HS_PX + 1);
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HS_PX, tid = threadIdx.x;
    const Lerp ly0 = lerp_index(y, s.h[0], Ho);
    const Lerp ly1 = lerp_index(y, s.n > 1 ? s.h[1] : Ho, Ho), ly2 = lerp_index(y, s.n > 2 ? s.h[2] : Ho, Ho);
    // four channels per thread (16-byte loads; Cs is a multiple of 8)
    const int G = Cs >> 2;
    for (int idx = tid; idx < HS_PX * G; idx += 256) {
        const int px = idx / G, c = (idx - px * G) * 4;
        const int x = x0 + px;
        if (x >= Wo || c >= C) continue;
        const float4 v = head_value4(s, ly0, ly1, ly2, b, y, x, c, Ho, Wo, Cs);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < C) tile[(c + e) * (HS_PX + 1) + px] = vv[e];
        if (FLIP) {
            const float4 m = head_value4(s, ly0, ly1, ly2, b + flip_from, y, Wo - 1 - x, c, Ho, Wo, Cs);
            const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < C) tile2[(c + e) * (HS_PX + 1) + px] = mm[e];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < C * HS_PX; idx += 256) {
        const int c = idx / HS_PX, px = idx - c * HS_PX;
        const int x = x0 + px;
        if (x >= Wo) continue;
        float v = tile[c * (HS_PX + 1) + px];
        if (FLIP) {
            const float f = tile2[pair[c] * (HS_PX + 1) + px];
            const bool neg = c >= n_kpt && ((c - n_kpt) & 1) == 0;
            v = v + (neg ? f * -1.f : f);
            if (c >= n_kpt) v = v * 0.5f;
        }
        if (scale) v = v / (c < n_kpt ? 255.f : 127.f);
        out[(((size_t)b * C + c) * Ho + y) * Wo + x] = v;
        // inf / NaN: an activation left the fp16 range upstream.  Status word b / 31: bit 0 = some frame of the word, bit 1 + b % 31 =
        // output frame b; word 0's bit 0 = some frame of the launch (include/smap_hip.h)
        if (status && !(fabsf(v) <= 3.4028234e38f)) {
            atomicOr(reinterpret_cast<unsigned*>(status) + b / 31, 1u | (2u << (b % 31)));
            if (b >= 31) atomicOr(reinterpret_cast<unsigned*>(status), 1u);
        }
    }
}

// SMAP_OP_TAPSUM: out[b,0,y,x] = bias + sum_{kh,kw} t[b, y+kh-1, x+kw-1][3 kh + kw] (zero outside the map): the stencil half of a 3x3 conv with
// one output channel whose per-pixel dot products came out of the producing conv launch (conv.hip TAPDOT).  Taps summed in the fixed order
// 0..8.  One thread per output pixel; the nine reads of a wave are nine coalesced 64-byte-strided gathers of neighbouring pixels.
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ t, const float* __restrict__ bias, float* __restrict__ out,
                                                     int B, int H, int W, int ts, int* __restrict__ status)
{
    const long long total = (long long)B * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
    float v = bias[0];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int yy = y + kh - 1, xx = x + kw - 1;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v += t[(((size_t)b * H + yy) * W + xx) * ts + kh * 3 + kw];
        }
    out[i] = v;
    if (status && !(fabsf(v) <= 3.4028234e38f)) {
        atomicOr(reinterpret_cast<unsigned*>(status) + b / 31, 1u | (2u << (b % 31)));
        if (b >= 31) atomicOr(reinterpret_cast<unsigned*>(status), 1u);
    }
}

inline int grid_for(long long total, int block)
{
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

struct smap_plan {
    std::vector<smap_op> ops;
    std::vector<int64_t> windows;          // arena offsets of the zero pages this schedule's conv launches address through
    struct Ticket { int64_t off, bytes; int op; };
    std::vector<Ticket> tickets;           // arena byte range of every split-K op's ticket slice (zeroed per op by smap_plan_run)
    // lanes (smap_op.lane): side streams 1 .. SMAP_MAX_LANES - 1 and one event per op some other lane waits for; created on first use
    bool lanes_on = false, lanes_ready = false;
    int n_lanes = 1;
    hipStream_t side[SMAP_MAX_LANES] = {};
    std::vector<hipEvent_t> ev;            // per op, null unless signalled
    std::vector<char> signalled;
    hipEvent_t join_ev[SMAP_MAX_LANES] = {};
};

// ZERO PAGES and WINDOWS.  The conv kernels address their input with (64-bit uniform base in SGPRs) + (32-bit byte offset per
// lane); offset 0..SMAP_ZERO_PAGE of that base must read as zeros (padding taps and rows past M fetch their 16 bytes there).
// The base of a launch is the WINDOW of its input: the arena offset in_off rounded down to a multiple of SMAP_WINDOW (4 GiB); as
// no tensor crosses a window boundary, an input tensor anywhere in an arena of any size is within 32 bits of its base.  Arena contract: bytes
// [k * SMAP_WINDOW, k * SMAP_WINDOW + SMAP_ZERO_PAGE) are reserved for every k >= 0 (no tensor overlaps them); smap_plan_run
// clears the ones its launches use on the stream before the first op.
constexpr int64_t SMAP_ZERO_PAGE = 16384;  // >= max Cin * 2 bytes + 16 (+ the lo-plane offset, <= 4096, in split precision):
                                           // a padding tap reads zero page + chunk*128 (+ lo offset)
constexpr int64_t SMAP_WINDOW = (int64_t)1 << 32;
inline int64_t window_of(int64_t off) { return off & ~(SMAP_WINDOW - 1); }
// does [off, off + bytes) touch a reserved zero page?
inline bool hits_zero_page(int64_t off, int64_t bytes)
{
    if (off < 0 || bytes <= 0) return false;
    const int64_t k0 = off / SMAP_WINDOW, k1 = (off + bytes - 1) / SMAP_WINDOW;
    if (off < k0 * SMAP_WINDOW + SMAP_ZERO_PAGE) return true;
    return k1 > k0;                        // crosses the start of the next window = its zero page
}

static int validate(const smap_op& o)
{
    if (o.B <= 0 || o.H <= 0 || o.W <= 0 || o.Ho <= 0 || o.Wo <= 0 || o.Cout <= 0) return SMAP_E_ARG;
    if (o.precision != 0 && o.precision != 1) return SMAP_E_ARG;
    if (o.precision == 1 && o.kind != SMAP_OP_CONV && o.kind != SMAP_OP_STEM && o.kind != SMAP_OP_MAXPOOL && o.kind != SMAP_OP_HEADSUM &&
        o.kind != SMAP_OP_STEMPOOL && o.kind != SMAP_OP_TAPSUM)
        return SMAP_E_ARG;                               // UPADD has no split-precision instance
    switch (o.kind) {
        case SMAP_OP_CONV: {
            int bm, bn;
            if (smap_conv_tile_dims(o.tile, &bm, &bn)) return SMAP_E_ARG;
            if (o.Cin % 64 || o.Cin * 2 + 16 > SMAP_ZERO_PAGE || o.cout_pad % bn || o.cout_pad < o.Cout) return SMAP_E_ARG;
            if (o.precision == 1) {
                if (!smap_conv_tile_has_x3(o.tile) || o.in_stride_c % 16 || (!o.out_fp32 && o.out_stride_c % 16)) return SMAP_E_ARG;
                if (o.Cin * 2 + o.in_stride_c + 16 > SMAP_ZERO_PAGE || !(o.acc_scale > 0.f)) return SMAP_E_ARG;
                if (o.in_c_off + o.Cin > o.in_stride_c / 2) return SMAP_E_ARG;
            }
            if (o.ksize != 1 && o.ksize != 3) return SMAP_E_ARG;
            if (o.w_pairs != 0 && o.w_pairs != 1) return SMAP_E_ARG;
            if (o.tile >= 30 && o.tile < 50 && (o.ksize != 3 || o.stride != 1 || o.pad != 1 || o.res_off >= 0 || o.add1_off >= 0 ||
                                 o.add2_off >= 0 || o.aux_off[0] >= 0))
                return SMAP_E_ARG;                       // halo-tiled kernel: plain 3x3 stride-1 convs only
            if ((o.tile >= 80 && o.tile < 100) != (o.tail_cout > 0)) return SMAP_E_ARG;
            if ((o.tile >= 90 && o.tile < 100) != (o.head_cin > 0)) return SMAP_E_ARG;
            if (o.tile >= 90 && o.tile < 100) {          // whole Bottleneck (convb.hip): split precision, P = 64 planes, 256 output channels
                const bool first = o.tile == 92 || o.tile == 93;       // a layer's FIRST block: 64 input channels, 1x1 shortcut conv instead of + x
                if (o.precision != 1 || o.ksize != 3 || o.stride != 1 || o.pad != 1 || o.out_fp32 || o.aux_off[0] >= 0) return SMAP_E_ARG;
                const int planes = o.tile == 94 ? 128 : 64;            // 94: csrc/convc.hip, 128 planes / 512 channels
                if (o.Cin != planes || o.Cout != planes || o.cout_pad != planes || o.head_cin != (first ? 64 : 4 * planes) || o.tail_cout != 4 * planes ||
                    o.tail_cout_pad != 4 * planes)
                    return SMAP_E_ARG;
                if (o.in_stride_c != 2 * o.head_cin || o.in_c_off != 0) return SMAP_E_ARG;
                if (first ? (o.res_off >= 0 || o.add1_off >= 0 || o.add2_off >= 0 || o.short_w_off < 0 || !(o.short_acc_scale > 0.f))
                          : (o.res_off != o.in_off || o.short_acc_scale != 0.f))     // identity block: the residual IS the input
                    return SMAP_E_ARG;
                if (o.head_w_off < 0 || o.head_bias_off < 0 || o.tail_w_off < 0 || o.tail_bias_off < 0) return SMAP_E_ARG;
                if (!(o.head_acc_scale > 0.f) || !(o.tail_acc_scale > 0.f) || o.out_stride_c < o.tail_cout) return SMAP_E_ARG;
                if ((int64_t)o.B * o.Ho * o.Wo * o.tail_cout * 2 >= ((int64_t)1 << 31)) return SMAP_E_ARG;
            } else if (o.short_acc_scale != 0.f) return SMAP_E_ARG;      // (a zero-initialised op has no shortcut conv)
            if (o.tile >= 80 && o.tile < 90) {           // 3x3 + fused 1x1 tail: the op's Cout is the tile's whole N extent
                const int bn2 = smap_conv_tile_tail_bn(o.tile);
                if (o.ksize != 3 || o.stride != 1 || o.pad != 1 || o.out_fp32 || o.aux_off[0] >= 0 || o.Cout != bn || o.cout_pad != bn)
                    return SMAP_E_ARG;
                if (o.tail_cout % 8 || o.tail_cout_pad % bn2 || o.tail_cout_pad < o.tail_cout || o.tail_w_off < 0 || o.tail_bias_off < 0)
                    return SMAP_E_ARG;
                if (o.precision == 1 && !(o.tail_acc_scale > 0.f)) return SMAP_E_ARG;
                if (o.out_stride_c < o.tail_cout) return SMAP_E_ARG;
                if ((int64_t)o.B * o.Ho * o.Wo * o.tail_cout * (1 + o.precision) >= ((int64_t)1 << 31)) return SMAP_E_ARG;
            }
            if (o.tile == 56 && (o.precision != 1 || o.out_fp32 || o.aux_off[0] >= 0 || o.add1_off >= 0 || o.add2_off >= 0 || o.Cout % 8 || o.ksplit > 1))
                return SMAP_E_ARG;                       // register-epilogue tile of conv.hip: split precision, fp16 outputs, residual + ReLU only
            if (o.tile >= 60 && o.tile < 80 && (o.out_fp32 || o.aux_off[0] >= 0 || o.Cout % 8 || o.cout_pad > 2048))
                return SMAP_E_ARG;                       // persistent kernel: register epilogue, fp16 outputs, no fused bilinear add, bias table of 2048 channels in LDS
            if (o.in_stride_c % 8 || o.in_c_off % 8 || o.out_stride_c % 8 || o.out_c_off % 8) return SMAP_E_ARG;
            if (o.out_stride_c < ((o.Cout + 7) & ~7) && o.tap_n == 0) return SMAP_E_ARG;      // (tap-dot: `out` is the [M][16] tap tensor)
            if ((o.res_off >= 0 || o.add1_off >= 0 || o.add2_off >= 0 || o.aux_off[0] >= 0) && o.Cout % 8) return SMAP_E_ARG;
            if (o.aux_off[0] >= 0 && (o.aux_h[0] <= 0 || o.aux_w[0] <= 0)) return SMAP_E_ARG;
            if (o.in_off < SMAP_ZERO_PAGE || o.out_off < SMAP_ZERO_PAGE || o.w_off < 0 || o.bias_off < 0) return SMAP_E_ARG;
            {   // conv A-operand addresses are 32-bit byte offsets from the input's window base; no tensor of the op may lie on a
                // reserved zero page
                const int64_t in_bytes = (int64_t)o.B * o.H * o.W * o.in_stride_c * 2;
                const int64_t M = (int64_t)o.B * o.Ho * o.Wo;
                const int64_t out_bytes = M * o.out_stride_c * (o.out_fp32 ? 4 : 2);
                const int64_t dense = M * ((o.tail_cout > 0 ? o.tail_cout : ((o.Cout + 7) & ~7))) * 2 * (1 + o.precision);
                if (o.in_off - window_of(o.in_off) + in_bytes > ((int64_t)1 << 32)) return SMAP_E_ARG;
                const int64_t up_bytes = o.aux_off[0] >= 0 ? (int64_t)o.B * o.aux_h[0] * o.aux_w[0] * ((o.Cout + 7) & ~7) * 2 * (1 + o.precision) : 0;
                if (hits_zero_page(o.in_off, in_bytes) || hits_zero_page(o.out_off, out_bytes) || hits_zero_page(o.res_off, dense) ||
                    hits_zero_page(o.add1_off, dense) || hits_zero_page(o.add2_off, dense) || hits_zero_page(o.aux_off[0], up_bytes))
                    return SMAP_E_ARG;
            }
            if ((int64_t)o.cout_pad * o.ksize * o.ksize * o.Cin * 2 * (1 + o.precision) > ((int64_t)1 << 32)) return SMAP_E_ARG;
            if (o.ksplit < 0 || o.ksplit > 16) return SMAP_E_ARG;
            if (o.ksplit > 1) {                          // split K: conv.hip's tiles; scratch and tickets inside the arena, off the zero pages
                const int bk = smap_conv_tile_bk(o.tile, o.precision);
                if (!smap_conv_tile_has_splitk(o.tile) || bk <= 0 || o.ksplit > o.ksize * o.ksize * o.Cin / bk) return SMAP_E_ARG;
                const int64_t tiles = (((int64_t)o.B * o.Ho * o.Wo + bm - 1) / bm) * (o.cout_pad / bn);
                const int64_t pbytes = tiles * o.ksplit * bm * bn * 4, cbytes = tiles * 4;
                if (o.kpart_off < SMAP_ZERO_PAGE || o.kcount_off < SMAP_ZERO_PAGE || o.kpart_off % 16 || o.kcount_off % 4) return SMAP_E_ARG;
                if (hits_zero_page(o.kpart_off, pbytes) || hits_zero_page(o.kcount_off, cbytes)) return SMAP_E_ARG;
            }
            if (o.tap_n != 0) {                          // tap-dot epilogue: tile 54, one N tile of 256 channels, t = fp32 [M][16]
                if (o.tap_n != 9 || o.tile != 54 || o.cout_pad != 256 || o.Cout != 256 || o.ksize != 1 || o.stride != 1 || !o.out_fp32 || o.out_stride_c != 16 ||
                    o.out_c_off != 0 || o.res_off >= 0 || o.add1_off >= 0 || o.add2_off >= 0 || o.aux_off[0] >= 0 || o.seg_n[0] != 0 || o.ksplit > 1 || o.in2_C != 0 ||
                    o.tap_w_off < 0 || !(o.tap_scale > 0.f))
                    return SMAP_E_ARG;
            }
            if (o.in2_C < 0) return SMAP_E_ARG;
            if (o.in2_C > 0) {                           // second input along K: conv.hip's tiles 20 / 50 / 51, 1x1 stride 1 on the first input, plain epilogue
                if (!smap_conv_tile_has_dual(o.tile) || o.ksize != 1 || o.stride != 1 || o.pad != 0 || o.ksplit > 1 || o.seg_n[0] != 0 || o.aux_off[0] >= 0 ||
                    o.add1_off >= 0 || o.add2_off >= 0 || o.out_fp32 || o.in_c_off != 0)
                    return SMAP_E_ARG;
                if (o.in2_C % 64 || o.in2_stride < 1 || o.in2_stride > 2 || o.in2_H <= 0 || o.in2_W <= 0 || o.in2_off < SMAP_ZERO_PAGE) return SMAP_E_ARG;
                if (o.Ho != (o.in2_H - 1) / o.in2_stride + 1 || o.Wo != (o.in2_W - 1) / o.in2_stride + 1) return SMAP_E_ARG;
                if (o.in2_stride_c % 8 || o.in2_stride_c < o.in2_C * (1 + o.precision) || (o.precision == 1 && o.in2_stride_c % 16)) return SMAP_E_ARG;
                if (o.in2_C * 2 + (o.precision ? o.in2_stride_c : 0) + 16 > SMAP_ZERO_PAGE) return SMAP_E_ARG;      // padding rows read the zero page (both planes)
                const int64_t b2 = (int64_t)o.B * o.in2_H * o.in2_W * o.in2_stride_c * 2;
                if (window_of(o.in2_off) != window_of(o.in_off) || o.in2_off - window_of(o.in_off) + b2 > ((int64_t)1 << 32) || hits_zero_page(o.in2_off, b2))
                    return SMAP_E_ARG;
                if ((int64_t)o.cout_pad * (o.Cin + o.in2_C) * 2 * (1 + o.precision) > ((int64_t)1 << 32)) return SMAP_E_ARG;
                if (o.in2_mode != 0 && o.in2_mode != 1) return SMAP_E_ARG;
                if (o.in2_mode == 1 && (!smap_conv_tile_has_relusum(o.tile) || o.in2_stride != 1 || o.res_off >= 0 || o.relu != 0 || o.in2_bias_off < 0 || (o.precision == 1 && !(o.in2_acc_scale > 0.f))))
                    return SMAP_E_ARG;               // relu(W1 x + b1) + relu(W2 x2 + b2): its own activations, no residual
            } else if (o.in2_mode != 0) return SMAP_E_ARG;
            if (o.seg_n[0] == 0 && o.seg_n[1] != 0) return SMAP_E_ARG;
            if (o.seg_n[0] != 0) {                       // N segments: conv.hip's tiles, 1x1, fp16 outputs; every segment starts on an N tile
                const bool igemm = (o.tile >= 0 && o.tile < 30) || (o.tile >= 50 && o.tile < 60);
                if (!igemm || o.ksize != 1 || o.out_fp32 || o.out_c_off != 0) return SMAP_E_ARG;
                int prev = 0;
                for (int j = 0; j < 2 && o.seg_n[j] != 0; ++j) {
                    if (o.seg_n[j] <= prev || o.seg_n[j] % bn || o.seg_n[j] >= o.cout_pad) return SMAP_E_ARG;
                    const int end = (j == 0 && o.seg_n[1] != 0) ? o.seg_n[1] : o.cout_pad;
                    if (o.seg_cout[j] <= 0 || o.seg_cout[j] % 8 || o.seg_n[j] + o.seg_cout[j] > end) return SMAP_E_ARG;
                    if (o.seg_out_stride_c[j] % 8 || (o.precision == 1 && o.seg_out_stride_c[j] % 16)) return SMAP_E_ARG;
                    if (o.seg_out_stride_c[j] < o.seg_cout[j] * (1 + o.precision)) return SMAP_E_ARG;
                    if (o.precision == 1 && !(o.seg_acc_scale[j] > 0.f)) return SMAP_E_ARG;
                    const int64_t sb = (int64_t)o.B * o.Ho * o.Wo * o.seg_out_stride_c[j];
                    if (sb >= ((int64_t)1 << 31) || o.seg_out_off[j] < SMAP_ZERO_PAGE || hits_zero_page(o.seg_out_off[j], sb * 2)) return SMAP_E_ARG;
                    prev = o.seg_n[j];
                }
                if (((o.Cout + 7) & ~7) > o.seg_n[0]) return SMAP_E_ARG;
            }
            // epilogues address outputs / residuals / addends / the low-res tensor with 32-bit ELEMENT offsets from their bases
            if ((int64_t)o.B * o.Ho * o.Wo * o.out_stride_c >= ((int64_t)1 << 31)) return SMAP_E_ARG;
            if ((int64_t)o.B * o.Ho * o.Wo * ((o.Cout + 7) & ~7) * (1 + o.precision) >= ((int64_t)1 << 31)) return SMAP_E_ARG;
            if (o.Ho != (o.H + 2 * o.pad - o.ksize) / o.stride + 1) return SMAP_E_ARG;
            if (o.Wo != (o.W + 2 * o.pad - o.ksize) / o.stride + 1) return SMAP_E_ARG;
            return 0;
        }
        case SMAP_OP_STEM:
            if (o.Cin != 3 || o.Cout != 64 || o.Ho != (o.H + 6 - 7) / 2 + 1 || o.Wo != (o.W + 6 - 7) / 2 + 1)
                return SMAP_E_ARG;
            if (o.flip_from < 0 || (o.flip_from > 0 && o.B != 2 * o.flip_from)) return SMAP_E_ARG;
            return 0;
        case SMAP_OP_STEMPOOL: {
            const int hs = (o.H + 6 - 7) / 2 + 1, ws = (o.W + 6 - 7) / 2 + 1;
            if (o.Cin != 3 || o.Cout != 64 || o.Ho != (hs + 2 - 3) / 2 + 1 || o.Wo != (ws + 2 - 3) / 2 + 1) return SMAP_E_ARG;
            if (o.flip_from < 0 || (o.flip_from > 0 && o.B != 2 * o.flip_from)) return SMAP_E_ARG;
            return 0;
        }
        case SMAP_OP_MAXPOOL:
            if (o.Cin % 8 || o.Cin != o.Cout || o.Ho != (o.H + 2 - 3) / 2 + 1 || o.Wo != (o.W + 2 - 3) / 2 + 1)
                return SMAP_E_ARG;
            return 0;
        case SMAP_OP_UPADD:
            if (o.Cout % 8 || o.aux_off[0] < 0 || o.aux_h[0] <= 0 || o.aux_w[0] <= 0) return SMAP_E_ARG;
            return 0;
        case SMAP_OP_TAPSUM:
            if (o.Cout != 1 || o.Cin < 9 || o.Cin % 4 || o.Ho != o.H || o.Wo != o.W || o.aux_off[0] < SMAP_ZERO_PAGE || o.ext_off < 0 || o.bias_off < 0) return SMAP_E_ARG;
            if (o.status_off < 0 || o.status_off % 4 || hits_zero_page(o.aux_off[0], (int64_t)o.B * o.H * o.W * o.Cin * 4)) return SMAP_E_ARG;
            return 0;
        case SMAP_OP_HEADSUM:
            if (o.n_aux < 1 || o.n_aux > 3 || o.Cout > 48 || o.Cin < o.Cout || o.ext_off < 0) return SMAP_E_ARG;
            if (o.flip_from < 0 || (o.flip_from > 0 && (o.w_off < 0 || o.in_c_off < 0 || o.in_c_off > o.Cout))) return SMAP_E_ARG;
            if (o.status_off < 0 || o.status_off % 4) return SMAP_E_ARG;
            if ((o.scale_hms != 0 && o.scale_hms != 1) || (o.scale_hms && (o.in_c_off < 0 || o.in_c_off > o.Cout))) return SMAP_E_ARG;
            return 0;
        default:
            return SMAP_E_ARG;
    }
}

#ifdef SMAP_TIMELINE
// Diagnostics build (tools/build_ablate.py --timeline): every conv launch gets a slice of a caller-provided buffer
// (env SMAP_TIMELINE_PTR = device address, SMAP_TIMELINE_CAP = launches it holds) in which its workgroups leave their
// start / end times on the device-wide 100 MHz clock and the CU they ran on (csrc/plan.h).
constexpr long long TL_SLICE = 4LL * (1 + 16384);               // int64 per launch: metadata row + up to 16384 workgroups
static int tl_next = 0;
static long long* tl_base() { const char* e = getenv("SMAP_TIMELINE_PTR"); return e ? reinterpret_cast<long long*>(strtoull(e, nullptr, 0)) : nullptr; }
static int tl_cap() { const char* e = getenv("SMAP_TIMELINE_CAP"); return e ? atoi(e) : 0; }
extern "C" __attribute__((visibility("default"))) void smap_timeline_reset(void) { tl_next = 0; }
extern "C" __attribute__((visibility("default"))) int smap_timeline_count(void) { return tl_next; }
#endif

extern "C" {

int smap_sizeof_op(void) { return (int)sizeof(smap_op); }

int smap_plan_create(const smap_op* ops, int n_ops, smap_plan** plan)
{
    if (!ops || !plan || n_ops <= 0 || n_ops > 4096) return SMAP_E_ARG;
    for (int i = 0; i < n_ops; ++i)
        if (int rc = validate(ops[i])) return rc;
    smap_plan* p = new (std::nothrow) smap_plan();
    if (!p) return SMAP_E_ARG;
    p->ops.assign(ops, ops + n_ops);
    p->signalled.assign(n_ops, 0);
    for (int i = 0; i < n_ops; ++i) {                    // lanes: every wait names an EARLIER op of ANOTHER lane
        const smap_op& o = ops[i];
        if (o.lane < 0 || o.lane >= SMAP_MAX_LANES || o.n_wait < 0 || o.n_wait > 4) { delete p; return SMAP_E_ARG; }
        if (o.lane + 1 > p->n_lanes) p->n_lanes = o.lane + 1;
        for (int k = 0; k < o.n_wait; ++k) {
            const int w = o.wait_op[k];
            if (w < 0 || w >= i || ops[w].lane == o.lane) { delete p; return SMAP_E_ARG; }
            p->signalled[w] = 1;
        }
    }
    p->windows.push_back(0);
    for (int i = 0; i < n_ops; ++i)
        if (ops[i].kind == SMAP_OP_CONV) {
            const int64_t w = window_of(ops[i].in_off);
            bool have = false;
            for (int64_t x : p->windows) have = have || x == w;
            if (!have) p->windows.push_back(w);
            if (ops[i].ksplit > 1) {
                int bm = 0, bn = 1;
                smap_conv_tile_dims(ops[i].tile, &bm, &bn);
                const int64_t tiles = (((int64_t)ops[i].B * ops[i].Ho * ops[i].Wo + bm - 1) / bm) * (ops[i].cout_pad / bn);
                p->tickets.push_back({ops[i].kcount_off, tiles * 4, i});
            }
        }
    // Split-K tickets: smap_plan_run zeroes EACH op's own slice (not a span from the lowest to the highest ticket: a foreign blob may put
    // tensors in between), and a slice overlaps neither another op's slice nor any tensor / scratch range an op of the schedule touches --
    // the tickets live for the whole schedule, whatever the packer reuses around them.
    for (size_t t = 0; t < p->tickets.size(); ++t) {
        const int64_t lo = p->tickets[t].off, hi = lo + p->tickets[t].bytes;
        for (size_t u = 0; u < t; ++u)
            if (lo < p->tickets[u].off + p->tickets[u].bytes && p->tickets[u].off < hi) { delete p; return SMAP_E_ARG; }
        for (int i = 0; i < n_ops; ++i) {
            const smap_op& o = ops[i];
            if (o.kind != SMAP_OP_CONV && o.kind != SMAP_OP_STEM && o.kind != SMAP_OP_MAXPOOL && o.kind != SMAP_OP_UPADD && o.kind != SMAP_OP_STEMPOOL) continue;
            const int64_t M = (int64_t)o.B * o.Ho * o.Wo, pl = 1 + o.precision;
            const int64_t c8 = o.tail_cout > 0 ? o.tail_cout : ((o.Cout + 7) & ~7);
            auto hit = [&](int64_t off, int64_t bytes) { return off >= 0 && bytes > 0 && off < hi && lo < off + bytes; };
            bool bad = hit(o.out_off, M * (o.kind == SMAP_OP_CONV ? (int64_t)o.out_stride_c * (o.out_fp32 ? 4 : 2) : c8 * 2 * pl));
            if (o.kind == SMAP_OP_CONV || o.kind == SMAP_OP_MAXPOOL || o.kind == SMAP_OP_UPADD)
                bad = bad || hit(o.in_off, (int64_t)o.B * o.H * o.W * (o.kind == SMAP_OP_CONV ? (int64_t)o.in_stride_c * 2 : (int64_t)o.Cin * 2 * pl));
            if (o.kind == SMAP_OP_CONV) {
                bad = bad || hit(o.res_off, M * c8 * 2 * pl) || hit(o.add1_off, M * c8 * 2 * pl) || hit(o.add2_off, M * c8 * 2 * pl);
                if (o.in2_C > 0) bad = bad || hit(o.in2_off, (int64_t)o.B * o.in2_H * o.in2_W * o.in2_stride_c * 2);
                bad = bad || hit(o.aux_off[0], (int64_t)o.B * o.aux_h[0] * o.aux_w[0] * c8 * 2 * pl);
                for (int j = 0; j < 2; ++j)
                    if (o.seg_n[j] > 0) bad = bad || hit(o.seg_out_off[j], M * o.seg_out_stride_c[j] * 2);
                if (o.ksplit > 1) {
                    int bm = 0, bn = 1;
                    smap_conv_tile_dims(o.tile, &bm, &bn);
                    const int64_t tiles = ((M + bm - 1) / bm) * (o.cout_pad / bn);
                    bad = bad || hit(o.kpart_off, tiles * o.ksplit * bm * bn * 4);
                }
            }
            if (bad) { delete p; return SMAP_E_ARG; }
        }
    }
    *plan = p;
    return 0;
}

static void lanes_release(smap_plan* p);

void smap_plan_destroy(smap_plan* plan)
{
    if (!plan) return;
    lanes_release(plan);
    delete plan;
}

static void lanes_release(smap_plan* p)
{
    for (hipEvent_t& e : p->ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (int l = 0; l < SMAP_MAX_LANES; ++l) {
        if (p->join_ev[l]) { (void)hipEventDestroy(p->join_ev[l]); p->join_ev[l] = nullptr; }
        if (p->side[l]) { (void)hipStreamDestroy(p->side[l]); p->side[l] = nullptr; }
    }
    p->lanes_ready = false;
}

// side streams + events of a plan whose lanes are switched on (device = the current one).  All or nothing: a failure half way releases
// what was created, so that the next attempt starts clean (round 5 created them lazily inside the const run and leaked on failure).
static int lanes_setup(smap_plan* p)
{
    if (p->lanes_ready) return 0;
    hipError_t e = hipSuccess;
    for (int l = 1; l < p->n_lanes && e == hipSuccess; ++l) e = hipStreamCreateWithFlags(&p->side[l], hipStreamNonBlocking);
    for (int l = 0; l < p->n_lanes && e == hipSuccess; ++l) e = hipEventCreateWithFlags(&p->join_ev[l], hipEventDisableTiming);
    p->ev.assign(p->ops.size(), nullptr);
    for (size_t i = 0; i < p->ops.size() && e == hipSuccess; ++i)
        if (p->signalled[i]) e = hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming);
    if (e != hipSuccess) { lanes_release(p); return hip_rc(e); }
    p->lanes_ready = true;
    return 0;
}

// Lanes on / off.  Switching them ON creates the side streams and events HERE (on the current device), not inside a run: a first run inside
// a stream capture must not create streams, smap_plan_run stays const, and a failure is this call's return code.  A plan with lanes on has
// ONE caller at a time (its side streams and events are shared by whoever runs it: executors of one plan on several host threads are not
// supported with lanes; without lanes a plan is immutable and may be run concurrently).
int smap_plan_set_lanes(smap_plan* plan, int on)
{
    if (!plan) return SMAP_E_ARG;
    if (on && plan->n_lanes > 1)
        if (int rc = lanes_setup(plan)) return rc;
    plan->lanes_on = on != 0;
    return 0;
}

static int run_ops(const smap_plan* plan, int first, int count, const float* const* inputs, int n_inputs, void* arena,
                   const void* weights, float* out, void* stream)
{
    if (!plan || !arena || !weights || first < 0 || count < 0 || first + count > (int)plan->ops.size())
        return SMAP_E_ARG;
    if (n_inputs < 0 || n_inputs > SMAP_MAX_INPUTS || (n_inputs > 0 && !inputs)) return SMAP_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    char* ar = static_cast<char*>(arena);
    const char* wb = static_cast<const char*>(weights);
    auto A = [&](int64_t off) -> _Float16* { return off < 0 ? nullptr : reinterpret_cast<_Float16*>(ar + off); };
    StemIn sin;
    for (int j = 0; j < SMAP_MAX_INPUTS; ++j) sin.p[j] = j < n_inputs ? inputs[j] : nullptr;
    sin.frames_per = 1;
    const bool have_input = n_inputs > 0 && inputs[0];
    for (int64_t w : plan->windows)
        if (hipError_t e = hipMemsetAsync(ar + w, 0, SMAP_ZERO_PAGE, st); e != hipSuccess) return hip_rc(e);
    // split-K tickets: zero before the first op (the kernels leave them at zero; an aborted run may not) -- the slices of the ops THIS run
    // launches, one memset each when they are apart, one for the lot when the packer laid them end to end (engine.py does)
    for (size_t t = 0; t < plan->tickets.size();) {
        if (plan->tickets[t].op < first || plan->tickets[t].op >= first + count) { ++t; continue; }
        int64_t lo = plan->tickets[t].off, hi = lo + plan->tickets[t].bytes;
        size_t u = t + 1;
        while (u < plan->tickets.size() && plan->tickets[u].off == hi && plan->tickets[u].op < first + count) hi += plan->tickets[u++].bytes;
        if (hipError_t e = hipMemsetAsync(ar + lo, 0, (size_t)(hi - lo), st); e != hipSuccess) return hip_rc(e);
        t = u;
    }
    for (int i = first; i < first + count; ++i)          // the status word (one per schedule) starts every run at 0
        if ((plan->ops[i].kind == SMAP_OP_HEADSUM || plan->ops[i].kind == SMAP_OP_TAPSUM) && plan->ops[i].status_off > 0 && out) {
            if (hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(out) + plan->ops[i].status_off, 0, 4 * (size_t)SMAP_STATUS_WORDS(plan->ops[i].B), st); e != hipSuccess) return hip_rc(e);
            break;
        }
    // lanes: only for whole-schedule runs (a partial range may start behind a fork)
    const bool lanes = plan->lanes_on && plan->lanes_ready && plan->n_lanes > 1 && first == 0 && count == (int)plan->ops.size();
    hipStream_t const st0 = st;
    // (error returns behind the fork go through `fail`: the side lanes are joined into the caller's stream first, so that nothing of this run
    //  is still in flight on a stream the caller does not know about)
    auto join = [&]() -> int {
        for (int l = 1; l < plan->n_lanes; ++l) {
            if (hipError_t e = hipEventRecord(plan->join_ev[l], plan->side[l]); e != hipSuccess) return hip_rc(e);
            if (hipError_t e = hipStreamWaitEvent(st0, plan->join_ev[l], 0); e != hipSuccess) return hip_rc(e);
        }
        return 0;
    };
    auto fail = [&](int rc) -> int { if (lanes) (void)join(); return rc; };
    if (lanes) {
        // the side lanes start behind what the caller's stream holds so far (its earlier work, the memsets above)
        if (hipError_t e = hipEventRecord(plan->join_ev[0], st0); e != hipSuccess) return hip_rc(e);
        for (int l = 1; l < plan->n_lanes; ++l)
            if (hipError_t e = hipStreamWaitEvent(plan->side[l], plan->join_ev[0], 0); e != hipSuccess) return hip_rc(e);
    }
    for (int i = first; i < first + count; ++i) {
        const smap_op& o = plan->ops[i];
        hipError_t e = hipSuccess;
        if (lanes) {
            st = o.lane > 0 ? plan->side[o.lane] : st0;
            for (int k = 0; k < o.n_wait; ++k)
                if (hipError_t e2 = hipStreamWaitEvent(st, plan->ev[o.wait_op[k]], 0); e2 != hipSuccess) return fail(hip_rc(e2));
        }
        switch (o.kind) {
            case SMAP_OP_CONV: {
                ConvArgs a;
                a.arena = ar + window_of(o.in_off);          // 64-bit base of the launch; lane offsets are 32-bit from here
                a.in_off = o.in_off - window_of(o.in_off);
                a.w = reinterpret_cast<const _Float16*>(wb + o.w_off);
                a.bias = reinterpret_cast<const float*>(wb + o.bias_off);
                a.out = ar + o.out_off;
                a.res = A(o.res_off);
                a.add1 = A(o.add1_off);
                a.add2 = A(o.add2_off);
                a.up = A(o.aux_off[0]);
                a.up_h = o.aux_h[0];
                a.up_w = o.aux_w[0];
#ifdef SMAP_TIMELINE
                a.tl = nullptr;
                if (tl_base() && tl_next < tl_cap()) {
                    a.tl = tl_base() + (long long)tl_next++ * TL_SLICE;
                    a.tl_meta[0] = i; a.tl_meta[1] = 0; a.tl_meta[2] = (long long)(uintptr_t)stream; a.tl_meta[3] = o.tile;
                }
#endif
#ifdef SMAP_TRACE
                a.dbg = getenv("SMAP_TRACE_PTR") ? reinterpret_cast<long long*>(strtoull(getenv("SMAP_TRACE_PTR"), nullptr, 0)) : nullptr;
#endif
                a.H = o.H; a.W = o.W; a.Cin = o.Cin; a.in_stride_c = o.in_stride_c; a.in_c_off = o.in_c_off;
                a.Ho = o.Ho; a.Wo = o.Wo; a.Cout8 = (o.Cout + 7) & ~7;
                a.ksize = o.ksize; a.stride = o.stride; a.pad = o.pad; a.relu = o.relu;
                a.out_stride_c = o.out_stride_c; a.out_c_off = o.out_c_off; a.out_fp32 = o.out_fp32;
                a.M = o.B * o.Ho * o.Wo;
                a.K = o.ksize * o.ksize * o.Cin + (o.in2_C > 0 ? o.in2_C : 0);      // (a second input extends K)
                a.Cin2 = o.in2_C > 0 ? o.in2_C : 0;
                a.in2_off = o.in2_C > 0 ? o.in2_off - window_of(o.in_off) : 0;       // same base as the first input (validate: same window)
                a.H2 = o.in2_H; a.W2 = o.in2_W; a.in2_stride_c = o.in2_stride_c; a.stride2 = o.in2_stride; a.in2_lo = o.in2_stride_c / 2;
                a.bias_b = (o.in2_C > 0 && o.in2_mode == 1) ? reinterpret_cast<const float*>(wb + o.in2_bias_off) : nullptr;
                a.acc_scale_b = o.in2_acc_scale;
                a.x3 = o.precision;
                a.in_lo = o.in_stride_c / 2;
                a.out_lo = o.out_stride_c / 2;
                a.w_lo = (long long)o.cout_pad * a.K * 2;
                a.acc_scale = o.acc_scale;
                a.w_pairs = o.w_pairs;
                a.w2 = o.tail_cout > 0 ? reinterpret_cast<const _Float16*>(wb + o.tail_w_off) : nullptr;
                a.bias2 = o.tail_cout > 0 ? reinterpret_cast<const float*>(wb + o.tail_bias_off) : nullptr;
                a.tail_cout8 = o.tail_cout;
                a.tail_chunks = o.tail_cout > 0 ? o.tail_cout_pad / smap_conv_tile_tail_bn(o.tile) : 0;
                a.tail_acc_scale = o.tail_acc_scale;
                a.w0 = o.head_cin > 0 ? reinterpret_cast<const _Float16*>(wb + o.head_w_off) : nullptr;
                a.bias0 = o.head_cin > 0 ? reinterpret_cast<const float*>(wb + o.head_bias_off) : nullptr;
                a.head_cin = o.head_cin;
                a.acc_scale0 = o.head_acc_scale;
                a.wd = o.head_cin > 0 && o.short_acc_scale > 0.f ? reinterpret_cast<const _Float16*>(wb + o.short_w_off) : nullptr;
                a.acc_scale_d = o.short_acc_scale;
                a.tap_n = o.tap_n;
                a.tap_w = o.tap_n > 0 ? reinterpret_cast<const _Float16*>(wb + o.tap_w_off) : nullptr;
                a.tap_scale = o.tap_scale;
                a.ksplit = o.ksplit > 1 ? o.ksplit : 1;
                a.kpart = o.ksplit > 1 ? reinterpret_cast<float*>(ar + o.kpart_off) : nullptr;
                a.kcount = o.ksplit > 1 ? reinterpret_cast<unsigned*>(ar + o.kcount_off) : nullptr;
                a.seg_n1 = o.seg_n[0] > 0 ? o.seg_n[0] : INT32_MAX;
                a.seg_n2 = o.seg_n[1] > 0 ? o.seg_n[1] : INT32_MAX;
                a.seg_out1 = o.seg_n[0] > 0 ? ar + o.seg_out_off[0] : nullptr;
                a.seg_out2 = o.seg_n[1] > 0 ? ar + o.seg_out_off[1] : nullptr;
                a.seg_cout8_1 = o.seg_cout[0]; a.seg_cout8_2 = o.seg_cout[1];
                a.seg_stride1 = o.seg_out_stride_c[0]; a.seg_stride2 = o.seg_out_stride_c[1];
                a.seg_relu1 = o.seg_relu[0]; a.seg_relu2 = o.seg_relu[1];
                a.seg_scale1 = o.seg_acc_scale[0]; a.seg_scale2 = o.seg_acc_scale[1];
                int bm, bn;
                smap_conv_tile_dims(o.tile, &bm, &bn);
                a.m_tiles = (a.M + bm - 1) / bm;
                a.n_tiles = o.cout_pad / bn;
                e = smap_launch_conv(a, o.tile, st);
                break;
            }
            case SMAP_OP_STEM: {
                const int frames = o.flip_from > 0 ? o.flip_from : o.B;       // frames the input holds
                if (!have_input || frames % n_inputs) return fail(SMAP_E_ARG);
                for (int j = 0; j < n_inputs; ++j) if (!inputs[j]) return fail(SMAP_E_ARG);
                sin.frames_per = frames / n_inputs;
                dim3 grid((o.Wo + ST_T - 1) / ST_T, (o.Ho + ST_T - 1) / ST_T, o.B);
                if (o.precision)
                    hipLaunchKernelGGL(stem_kernel<true>, grid, dim3(256), 0, st, sin,
                                       reinterpret_cast<const _Float16*>(wb + o.w_off),
                                       reinterpret_cast<const float*>(wb + o.bias_off), A(o.out_off), o.H, o.W, o.Ho,
                                       o.Wo, o.acc_scale, o.flip_from);
                else
                    hipLaunchKernelGGL(stem_kernel<false>, grid, dim3(256), 0, st, sin,
                                       reinterpret_cast<const _Float16*>(wb + o.w_off),
                                       reinterpret_cast<const float*>(wb + o.bias_off), A(o.out_off), o.H, o.W, o.Ho,
                                       o.Wo, 1.f, o.flip_from);
                e = hipGetLastError();
                break;
            }
            case SMAP_OP_STEMPOOL: {
                const int frames = o.flip_from > 0 ? o.flip_from : o.B;
                if (!have_input || frames % n_inputs) return fail(SMAP_E_ARG);
                for (int j = 0; j < n_inputs; ++j) if (!inputs[j]) return fail(SMAP_E_ARG);
                sin.frames_per = frames / n_inputs;
                const int hs = (o.H + 6 - 7) / 2 + 1, ws = (o.W + 6 - 7) / 2 + 1;
                dim3 grid((o.Wo + SP_PX - 1) / SP_PX, (o.Ho + SP_PY - 1) / SP_PY, o.B);
                if (o.precision)
                    hipLaunchKernelGGL(stem_pool_kernel<true>, grid, dim3(256), 0, st, sin,
                                       reinterpret_cast<const _Float16*>(wb + o.w_off),
                                       reinterpret_cast<const float*>(wb + o.bias_off), A(o.out_off), o.H, o.W, hs, ws,
                                       o.Ho, o.Wo, o.acc_scale, o.flip_from);
                else
                    hipLaunchKernelGGL(stem_pool_kernel<false>, grid, dim3(256), 0, st, sin,
                                       reinterpret_cast<const _Float16*>(wb + o.w_off),
                                       reinterpret_cast<const float*>(wb + o.bias_off), A(o.out_off), o.H, o.W, hs, ws,
                                       o.Ho, o.Wo, 1.f, o.flip_from);
                e = hipGetLastError();
                break;
            }
            case SMAP_OP_MAXPOOL: {
                const long long total = (long long)o.B * o.Ho * o.Wo * (o.Cin / 8);
                if (o.precision)
                    hipLaunchKernelGGL(maxpool_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, st, A(o.in_off),
                                       A(o.out_off), o.B, o.H, o.W, o.Cin, o.Ho, o.Wo);
                else
                    hipLaunchKernelGGL(maxpool_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, st, A(o.in_off),
                                       A(o.out_off), o.B, o.H, o.W, o.Cin, o.Ho, o.Wo);
                e = hipGetLastError();
                break;
            }
            case SMAP_OP_UPADD: {
                const long long total = (long long)o.B * o.Ho * o.Wo * (o.Cout / 8);
                hipLaunchKernelGGL(upadd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, A(o.in_off),
                                   A(o.aux_off[0]), A(o.out_off), o.B, o.Ho, o.Wo, o.Cout, o.aux_h[0], o.aux_w[0],
                                   o.relu);
                e = hipGetLastError();
                break;
            }
            case SMAP_OP_TAPSUM: {
                if (!out) return fail(SMAP_E_ARG);
                const long long total = (long long)o.B * o.H * o.W;
                int* status = o.status_off > 0 ? reinterpret_cast<int*>(reinterpret_cast<char*>(out) + o.status_off) : nullptr;
                hipLaunchKernelGGL(tapsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float*>(ar + o.aux_off[0]),
                                   reinterpret_cast<const float*>(wb + o.bias_off), reinterpret_cast<float*>(reinterpret_cast<char*>(out) + o.ext_off),
                                   o.B, o.H, o.W, o.Cin, status);
                e = hipGetLastError();
                break;
            }
            case SMAP_OP_HEADSUM: {
                if (!out) return fail(SMAP_E_ARG);
                HeadSrc s;
                s.n = o.n_aux;
                for (int k = 0; k < 3; ++k) {
                    s.p[k] = k < o.n_aux ? reinterpret_cast<const float*>(ar + o.aux_off[k]) : nullptr;
                    s.h[k] = o.aux_h[k];
                    s.w[k] = o.aux_w[k];
                }
                dim3 grid((o.Wo + HS_PX - 1) / HS_PX, o.Ho, o.B);
                float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + o.ext_off);
                int* status = o.status_off > 0 ? reinterpret_cast<int*>(reinterpret_cast<char*>(out) + o.status_off) : nullptr;
                if (o.flip_from > 0)
                    hipLaunchKernelGGL(headsum_kernel<true>, grid, dim3(256), 0, st, s, dst, o.Ho, o.Wo, o.Cout, o.Cin,
                                       o.flip_from, reinterpret_cast<const int*>(wb + o.w_off), o.in_c_off, status, o.scale_hms);
                else
                    hipLaunchKernelGGL(headsum_kernel<false>, grid, dim3(256), 0, st, s, dst, o.Ho, o.Wo, o.Cout, o.Cin,
                                       0, nullptr, o.in_c_off, status, o.scale_hms);
                e = hipGetLastError();
                break;
            }
            default:
                return fail(SMAP_E_ARG);
        }
        if (e != hipSuccess) return fail(hip_rc(e));
        if (lanes && plan->signalled[i])
            if (hipError_t e2 = hipEventRecord(plan->ev[i], st); e2 != hipSuccess) return fail(hip_rc(e2));
    }
    if (lanes) return join();                            // the caller's stream continues when every lane is done
    return 0;
}

// arena / output bytes the schedule touches, from the ops alone (a host that did not build the schedule sizes its buffers with it)
int smap_workspace_bytes(const smap_plan* plan, int64_t* arena_bytes, int64_t* out_bytes)
{
    if (!plan) return SMAP_E_ARG;
    int64_t ar = SMAP_ZERO_PAGE, ob = 0;
    auto up = [](int64_t& m, int64_t off, int64_t bytes) { if (off >= 0 && off + bytes > m) m = off + bytes; };
    for (const smap_op& o : plan->ops) {
        const int64_t M = (int64_t)o.B * o.Ho * o.Wo, pl = 1 + o.precision;
        switch (o.kind) {
            case SMAP_OP_CONV: {
                const int64_t c8 = o.tail_cout > 0 ? o.tail_cout : ((o.Cout + 7) & ~7);
                up(ar, o.in_off, (int64_t)o.B * o.H * o.W * o.in_stride_c * 2);
                if (o.in2_C > 0) up(ar, o.in2_off, (int64_t)o.B * o.in2_H * o.in2_W * o.in2_stride_c * 2);
                up(ar, o.out_off, M * o.out_stride_c * (o.out_fp32 ? 4 : 2));
                up(ar, o.res_off, M * c8 * 2 * pl); up(ar, o.add1_off, M * c8 * 2 * pl); up(ar, o.add2_off, M * c8 * 2 * pl);
                up(ar, o.aux_off[0], (int64_t)o.B * o.aux_h[0] * o.aux_w[0] * c8 * 2 * pl);
                for (int j = 0; j < 2; ++j)
                    if (o.seg_n[j] > 0) up(ar, o.seg_out_off[j], M * o.seg_out_stride_c[j] * 2);
                if (o.ksplit > 1) {
                    int bm = 0, bn = 1;
                    smap_conv_tile_dims(o.tile, &bm, &bn);
                    const int64_t tiles = ((M + bm - 1) / bm) * (o.cout_pad / bn);
                    up(ar, o.kpart_off, tiles * o.ksplit * bm * bn * 4);
                    up(ar, o.kcount_off, tiles * 4);
                }
                break;
            }
            case SMAP_OP_STEM: case SMAP_OP_STEMPOOL: up(ar, o.out_off, M * 64 * 2 * pl); break;
            case SMAP_OP_MAXPOOL:
                up(ar, o.in_off, (int64_t)o.B * o.H * o.W * o.Cin * 2 * pl); up(ar, o.out_off, M * o.Cout * 2 * pl); break;
            case SMAP_OP_UPADD:
                up(ar, o.in_off, M * o.Cout * 2); up(ar, o.out_off, M * o.Cout * 2);
                up(ar, o.aux_off[0], (int64_t)o.B * o.aux_h[0] * o.aux_w[0] * o.Cout * 2); break;
            case SMAP_OP_TAPSUM:
                up(ar, o.aux_off[0], M * o.Cin * 4);
                up(ob, o.ext_off, M * 4);
                if (o.status_off > 0) up(ob, o.status_off, 4 * (int64_t)SMAP_STATUS_WORDS(o.B));
                break;
            case SMAP_OP_HEADSUM: {
                const int64_t frames = o.flip_from > 0 ? 2 * (int64_t)o.B : o.B;        // the sources hold the mirrored half too
                for (int k = 0; k < o.n_aux; ++k) up(ar, o.aux_off[k], frames * o.aux_h[k] * o.aux_w[k] * o.Cin * 4);
                up(ob, o.ext_off, M * o.Cout * 4);
                if (o.status_off > 0) up(ob, o.status_off, 4 * (int64_t)SMAP_STATUS_WORDS(o.B));
                break;
            }
            default: break;
        }
    }
    if (arena_bytes) *arena_bytes = ar;
    if (out_bytes) *out_bytes = ob;
    return 0;
}

// ---- serialised plans: include/smap_hip.h "plan blob"
int smap_plan_create_from_blob(const void* blob, size_t blob_bytes, smap_plan** plan, smap_blob_info* info)
{
    if (!blob || !plan || blob_bytes < sizeof(smap_blob_header)) return SMAP_E_ARG;
    smap_blob_header h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, "SMAPPLN1", 8) || h.version != SMAP_BLOB_VERSION || h.sizeof_op != sizeof(smap_op) || h.header_bytes != sizeof(smap_blob_header))
        return SMAP_E_ARG;
    // every (offset, size) pair is checked as `off <= total - size` with total - size >= 0: no sum that could wrap
    const int64_t total = (int64_t)blob_bytes;
    auto inside = [](int64_t off, int64_t size, int64_t total_) { return off >= 0 && size >= 0 && size <= total_ && off <= total_ - size; };
    if (h.n_ops <= 0 || h.n_ops > 4096 || h.ops_offset < (int64_t)sizeof(h)) return SMAP_E_ARG;
    if (!inside(h.ops_offset, (int64_t)h.n_ops * (int64_t)sizeof(smap_op), total) || !inside(h.weights_offset, h.weights_bytes, total)) return SMAP_E_ARG;
    std::vector<smap_op> ops(h.n_ops);
    memcpy(ops.data(), static_cast<const char*>(blob) + h.ops_offset, (size_t)h.n_ops * sizeof(smap_op));
    for (const smap_op& o : ops) {      // the blob's weight section must hold EVERY byte the ops point at (packed sizes: include/smap_hip.h)
        const int64_t wb = h.weights_bytes, pl = o.precision == 1 ? 2 : 1;
        bool ok = true;
        if (o.kind == SMAP_OP_CONV) {
            if (o.cout_pad <= 0 || o.Cin <= 0 || o.ksize <= 0 || o.ksize > 7) return SMAP_E_ARG;
            if (o.in2_C < 0 || o.in2_C > (1 << 20)) return SMAP_E_ARG;
            ok = inside(o.w_off, (int64_t)o.cout_pad * (o.ksize * o.ksize * o.Cin + o.in2_C) * 2 * pl, wb) && inside(o.bias_off, (int64_t)o.cout_pad * 4, wb);
            if (o.tail_cout > 0)
                ok = ok && o.tail_cout_pad > 0 && o.Cout > 0 && inside(o.tail_w_off, (int64_t)o.tail_cout_pad * o.Cout * 2 * pl, wb) &&
                     inside(o.tail_bias_off, (int64_t)o.tail_cout_pad * 4, wb);
            if (o.head_cin > 0) {
                ok = ok && inside(o.head_w_off, (int64_t)o.head_cin * o.Cin * 2 * pl, wb) && inside(o.head_bias_off, (int64_t)o.Cin * 4, wb);
                if (o.short_acc_scale > 0.f) ok = ok && o.tail_cout > 0 && inside(o.short_w_off, (int64_t)o.tail_cout * o.head_cin * 2 * pl, wb);
            }
        } else if (o.kind == SMAP_OP_STEM || o.kind == SMAP_OP_STEMPOOL) {
            ok = inside(o.w_off, (int64_t)64 * ST_K * 2 * pl, wb) && inside(o.bias_off, 64 * 4, wb);
        } else if (o.kind == SMAP_OP_HEADSUM && o.flip_from > 0) {
            ok = o.Cout > 0 && inside(o.w_off, (int64_t)o.Cout * 4, wb);
        } else if (o.kind == SMAP_OP_TAPSUM) {
            ok = inside(o.bias_off, 4, wb);
        }
        if (o.kind == SMAP_OP_CONV && o.in2_C > 0 && o.in2_mode == 1) ok = ok && inside(o.in2_bias_off, (int64_t)o.cout_pad * 4, wb);
        if (o.kind == SMAP_OP_CONV && o.tap_n > 0) ok = ok && inside(o.tap_w_off, (int64_t)(o.cout_pad / 32) * 2 * 64 * 8 * 2, wb);
        if (!ok) return SMAP_E_ARG;
    }
    smap_plan* p = nullptr;
    if (int rc = smap_plan_create(ops.data(), h.n_ops, &p)) return rc;
    int64_t ar = 0, ob = 0;
    smap_workspace_bytes(p, &ar, &ob);
    // the sizes a host allocates from (header AND info) must cover what the ops touch
    if (ar > h.arena_bytes || ob > h.out_bytes || ar > h.info.arena_bytes || ob > h.info.out_bytes ||
        h.info.weights_offset != h.weights_offset || h.info.weights_bytes != h.weights_bytes) { smap_plan_destroy(p); return SMAP_E_ARG; }
    if (info) *info = h.info;
    *plan = p;
    return 0;
}

int smap_plan_run_range(const smap_plan* plan, int first, int count, const float* input, void* arena,
                        const void* weights, float* out, void* stream)
{
    return run_ops(plan, first, count, &input, input ? 1 : 0, arena, weights, out, stream);
}

int smap_plan_run(const smap_plan* plan, const float* input, void* arena, const void* weights, float* out,
                  void* stream)
{
    if (!plan) return SMAP_E_ARG;
    return run_ops(plan, 0, (int)plan->ops.size(), &input, input ? 1 : 0, arena, weights, out, stream);
}

int smap_plan_run_inputs(const smap_plan* plan, const float* const* inputs, int n_inputs, void* arena, const void* weights,
                         float* out, void* stream)
{
    if (!plan || n_inputs < 1) return SMAP_E_ARG;
    return run_ops(plan, 0, (int)plan->ops.size(), inputs, n_inputs, arena, weights, out, stream);
}

}  // extern "C"
