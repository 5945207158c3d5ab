// conv2.hip -- second-generation implicit-GEMM conv kernel for gfx950 (same contract as conv.hip:
// conv_bn_relu of model/smap.py:13-45 with folded BN, residual / ReLU / skip adds / bilinear add fused).
//
// What changed against conv.hip and why (measurements: DESIGN.md section 9):
//  * the kernel time of these layers follows  HBM bytes / ~5 TB/s  +  (L2 -> LDS bytes) / ~15 TB/s ,
//    and the L2 -> LDS stream (57 GB per B=8 forward, more than half of it weight tiles re-streamed
//    by every M tile) is the larger half.  Fewer L2 -> LDS bytes per FLOP need bigger block tiles,
//    bigger tiles need a smaller LDS footprint per workgroup to keep 3-5 workgroups per CU resident:
//      - BK is a template parameter (32 or 64): a 128x128 tile stages 2 x 16 KB instead of 2 x 32 KB;
//      - the epilogue no longer bounces the accumulators through a 64 KB fp32 LDS tile: the MFMA
//        operands are swapped (weights = A operand, activations = B operand) so that D[n][m] leaves
//        4 CONSECUTIVE CHANNELS of ONE PIXEL in each lane -> 8-byte NHWC stores (16 B for fp32 heads)
//        straight from registers, lanes l and l+32 writing adjacent 8 bytes of the same pixel.
//    The arithmetic is unchanged (same products, same K order) -- results are bit-identical to conv.hip.
//  * LDS image for BK = 32: 64-byte rows, LDS-DMA lane l -> row l/4, slot l%4, slot s of row r holds
//    K-granule s ^ ((r>>2)&3)  (conflict-free ds_read_b128 groups, same derivation as conv.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// FULL = false: epilogue with bias / residual / ReLU only (most layers; keeps the VGPR count low);
// FULL = true : + bilinear add and post-ReLU addends.
template <int BM, int BN, int BK, int STAGES, bool FULL>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvArgs a)
{
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(STAGES >= 2 && STAGES <= 4, "2..4 LDS stages");
    constexpr int ROWB = BK * 2;                        // bytes per LDS row
    constexpr int RPW = 1024 / ROWB;                    // rows one wave-wide LDS-DMA instruction covers (8 / 16)
    constexpr int RPR = 4 * RPW;                        // rows per round of the 4 waves (32 / 64)
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile must be a multiple of the DMA round");
    constexpr int LA = BM / RPR, LB = BN / RPR, LPT = LA + LB;
    constexpr int MI = BM / 2 / 32, NI = BN / 2 / 32;   // 2 x 2 waves
    static_assert(MI >= 1 && NI >= 1, "tile too small");
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert((STAGES - 2) * LPT <= 63, "vmcnt is 6 bits");
    static_assert(STAGES * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int logical;                                         // XCD-aware tile order (see conv.hip)
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m_tile = logical / a.n_tiles, n_tile = logical - m_tile * a.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    // ---- staging geometry: uniform 64-bit base + 32-bit per-lane byte offset (zero page = arena[0..))
    constexpr int SPR = ROWB / 16;                       // 16-byte slots per row (4 / 8)
    const int lrow = lane / SPR, lslot = lane % SPR;
    const int srow = wave * RPW + lrow;                  // row inside a round
    const int gch = BK == 64 ? (lslot ^ ((srow >> 1) & 7)) : (lslot ^ ((srow >> 2) & 3));
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ wt = reinterpret_cast<const char*>(a.w);
    const int HoWo = a.Ho * a.Wo;

    // Per staged A row: byte offset of tap (0,0) and the 9-bit mask of in-range taps.  The pixel
    // decomposition m -> (b, oy, ox) costs two integer divisions ONCE per thread; the thread's other rows
    // are RPR pixels further along the raster and are reached by carry propagation.  The tap mask is the
    // outer product of three row tests and three column tests (no loop over taps).
    unsigned a_off[LA], a_mask[LA];
    {
        int m = m0 + srow;
        int b = m / HoWo, rem = m - b * HoWo;
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            a_off[i] = 0;
            a_mask[i] = 0;
            if (m < a.M) {
                const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                const long long e = ((long long)(b * a.H + iy0) * a.W + ix0) * a.in_stride_c + a.in_c_off + gch * 8;
                a_off[i] = (unsigned)(a.in_off + e * 2);
                unsigned vx = 0, mk = 0;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (kw < a.ksize && (unsigned)(ix0 + kw) < (unsigned)a.W) vx |= 1u << kw;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
                    if (kh < a.ksize && (unsigned)(iy0 + kh) < (unsigned)a.H) mk |= vx << (kh * a.ksize);
                a_mask[i] = mk;
            }
            m += RPR;
            ox += RPR;
            while (ox >= a.Wo) { ox -= a.Wo; ++oy; }
            while (oy >= a.Ho) { oy -= a.Ho; ++b; }
        }
    }
    unsigned b_off[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) b_off[i] = (unsigned)(((n0 + i * RPR + srow) * a.K + gch * 8) * 2);

    const int cchunks = a.Cin / BK;
    const int n_iter = a.ksize * a.ksize * cchunks;

    int s_kh = 0, s_kw = 0, s_cc = 0;
    unsigned s_boff = 0;
    unsigned a_cur[LA];
    auto set_tap = [&]() {
        const unsigned tap_off = (unsigned)((s_kh * a.W + s_kw) * a.in_stride_c * 2);
        const unsigned bit = 1u << (s_kh * a.ksize + s_kw);
#pragma unroll
        for (int i = 0; i < LA; ++i) a_cur[i] = (a_mask[i] & bit) ? a_off[i] + tap_off : 0u;
    };
    set_tap();
    auto stage = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * ROWB;
        const char* gA = arena + (unsigned)(s_cc * ROWB);
        const char* gB = wt + s_boff;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_cur[i]), (lds_void*)(sA + (i * RPR + wave * RPW) * ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + b_off[i]), (lds_void*)(sB + (i * RPR + wave * RPW) * ROWB), 16, 0, 0);
        s_boff += ROWB;
        if (++s_cc == cchunks) {
            s_cc = 0;
            if (++s_kw == a.ksize) { s_kw = 0; ++s_kh; }
            set_tap();
        }
    };

    // ---- accumulators: D[n][m], acc[ni][mi]
    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rswz = BK == 64 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    const int a_row0 = wm * (BM / 2) + l31;              // + mi*32   (activation rows = pixels)
    const int b_row0 = wn * (BN / 2) + l31;              // + ni*32   (weight rows = channels)

#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < n_iter) stage(st);
    int buf = 0, nbuf = STAGES - 1;
    for (int it = 0; it < n_iter; ++it) {
        if (it + STAGES - 1 <= n_iter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + STAGES - 1 < n_iter) stage(nbuf);
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int slot = ((kk * 2 + lhi) ^ rswz) * 16;
            half8 xf[MI], wf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                xf[mi] = *reinterpret_cast<const half8*>(sA + (a_row0 + mi * 32) * ROWB + slot);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *reinterpret_cast<const half8*>(sB + (b_row0 + ni * 32) * ROWB + slot);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }

    // ---- epilogue straight from registers.  Lane = pixel m (MFMA column), register quad q of tile ni
    //      = channels n .. n+3 with n = n0 + wn*BN/2 + ni*32 + 8q + 4*lhi.
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * (BM / 2) + mi * 32 + l31;
        const bool okm = m < a.M;
        const int ms = okm ? m : 0;
        const int nb = n0 + wn * (BN / 2) + 4 * lhi;           // + ni*32 + 8q
        // batch the global reads of this pixel (residual, bilinear taps, post-ReLU addends), then the math
        half4 rr[NI][4], q1[NI][4], q2[NI][4];
        half4 t00[NI][4], t01[NI][4], t10[NI][4], t11[NI][4];
        float ly0 = 0.f, ly1 = 0.f, lx0 = 0.f, lx1 = 0.f;
        const long long dense = (long long)ms * a.Cout8;
        if (a.res) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + ni * 32 + 8 * q;
                    rr[ni][q] = *reinterpret_cast<const half4*>(a.res + dense + (n < a.Cout8 ? n : 0));
                }
        }
        if (FULL && a.up) {
            const int b = ms / HoWo, rem = ms - b * HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const Lerp ly = lerp_index(oy, a.up_h, a.Ho), lx = lerp_index(ox, a.up_w, a.Wo);
            ly0 = ly.l0; ly1 = ly.l1; lx0 = lx.l0; lx1 = lx.l1;
            const _Float16* tb = a.up + (long long)b * a.up_h * a.up_w * a.Cout8;
            const _Float16* p00 = tb + ((long long)ly.i0 * a.up_w + lx.i0) * a.Cout8;
            const _Float16* p01 = tb + ((long long)ly.i0 * a.up_w + lx.i1) * a.Cout8;
            const _Float16* p10 = tb + ((long long)ly.i1 * a.up_w + lx.i0) * a.Cout8;
            const _Float16* p11 = tb + ((long long)ly.i1 * a.up_w + lx.i1) * a.Cout8;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + ni * 32 + 8 * q;
                    const int ns = n < a.Cout8 ? n : 0;
                    t00[ni][q] = *reinterpret_cast<const half4*>(p00 + ns);
                    t01[ni][q] = *reinterpret_cast<const half4*>(p01 + ns);
                    t10[ni][q] = *reinterpret_cast<const half4*>(p10 + ns);
                    t11[ni][q] = *reinterpret_cast<const half4*>(p11 + ns);
                }
        }
        if (FULL && a.add1) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + ni * 32 + 8 * q;
                    q1[ni][q] = *reinterpret_cast<const half4*>(a.add1 + dense + (n < a.Cout8 ? n : 0));
                }
        }
        if (FULL && a.add2) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + ni * 32 + 8 * q;
                    q2[ni][q] = *reinterpret_cast<const half4*>(a.add2 + dense + (n < a.Cout8 ? n : 0));
                }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + ni * 32 + 8 * q;
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);      // bias is padded to cout_pad
                float v[4] = {acc[ni][mi][4 * q] + bv.x, acc[ni][mi][4 * q + 1] + bv.y,
                              acc[ni][mi][4 * q + 2] + bv.z, acc[ni][mi][4 * q + 3] + bv.w};
                if (a.res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rr[ni][q][e];
                }
                if (FULL && a.up) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] += ly0 * (lx0 * (float)t00[ni][q][e] + lx1 * (float)t01[ni][q][e]) +
                                ly1 * (lx0 * (float)t10[ni][q][e] + lx1 * (float)t11[ni][q][e]);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if (FULL && a.add1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)q1[ni][q][e];
                }
                if (FULL && a.add2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)q2[ni][q][e];
                }
                if (!okm || n >= a.Cout8) continue;
                const long long o = (long long)m * a.out_stride_c + a.out_c_off + n;
                if (a.out_fp32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
                    *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(a.out) + o) = h;
                }
            }
    }
}

template <int BM, int BN, int BK, int STAGES>
hipError_t launch2(const ConvArgs& a, hipStream_t st)
{
    if (a.up || a.add1 || a.add2)
        hipLaunchKernelGGL((conv_igemm2_kernel<BM, BN, BK, STAGES, true>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm2_kernel<BM, BN, BK, STAGES, false>), dim3(a.m_tiles * a.n_tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace

// tile ids 10.. : second-generation kernel.  Keep in sync with smap_amd/engine.py::TILES.
int smap_conv2_tile_dims(int tile, int* bm, int* bn)
{
    switch (tile) {
        case 10: case 14: case 17: *bm = 128; *bn = 128; return 0;
        case 11: case 15: *bm = 128; *bn = 64; return 0;
        case 12: case 16: *bm = 64; *bn = 64; return 0;
        case 13: case 18: *bm = 64; *bn = 128; return 0;
        default: return -1;
    }
}

hipError_t smap_launch_conv2(const ConvArgs& a, int tile, hipStream_t st)
{
    switch (tile) {
        case 10: return launch2<128, 128, 32, 2>(a, st);    // 32 KiB LDS
        case 11: return launch2<128, 64, 32, 2>(a, st);     // 24 KiB
        case 12: return launch2<64, 64, 32, 2>(a, st);      // 16 KiB
        case 13: return launch2<64, 128, 32, 2>(a, st);     // 24 KiB
        case 14: return launch2<128, 128, 64, 2>(a, st);    // 64 KiB
        case 15: return launch2<128, 64, 64, 2>(a, st);     // 48 KiB
        case 16: return launch2<64, 64, 64, 2>(a, st);      // 32 KiB
        case 17: return launch2<128, 128, 32, 3>(a, st);    // 48 KiB, 2 tiles in flight
        case 18: return launch2<64, 128, 64, 2>(a, st);     // 48 KiB
        default: return hipErrorInvalidValue;
    }
}
