// conv1.hip -- weight-stationary, persistent 1x1 stride-1 conv_bn_relu (model/smap.py:13-45: the Bottleneck
// c1/c3 convs, the Upsample_unit laterals and skips) for gfx950.
//
// Why: the wide 1x1 layers (K <= 256, N >= 256) are streaming problems -- per 64 rows a CU moves 64 x (K + 256
// [+ 256 residual]) x 2 bytes against 2048 MFMA cycles -- yet the tiled kernel of conv.hip re-streams the whole
// weight matrix L2 -> LDS once per tile and serialises load, multiply and store inside every short-lived
// workgroup.  Here the weights never move again after the first microsecond and the three streams run decoupled:
//
//   * a workgroup = 4 MFMA waves + NLOAD loader waves; it owns 256 output channels (MFMA wave w: 64 of them) and
//     walks a strided list of 64-row tiles of the activation matrix (persistent: one workgroup per CU);
//   * each MFMA wave keeps ITS weight slice [64 ch][K] in registers as ready-made fragments (32*K/64 VGPRs);
//   * the loader waves stream, through a ring of 8 KB LDS slots filled by LDS-DMA, the activation rows
//     ([64 rows][64 ch] fp16 per slot, the XOR-swizzled layout of conv.hip) AND the residual rows (one slot per
//     MFMA wave: [64 rows][its 64 ch]).  Only the loaders wait on vmcnt, and they issue nothing but loads, so
//     the counted wait is exact; the MFMA waves issue nothing but stores and never wait on vmcnt at all
//     (loads and stores share vmcnt on gfx9 and retire out of order with respect to each other: a residual
//     load waited for in the epilogue would also wait for every older store -- measured: 1 us per 32 rows);
//   * one raw s_barrier per step (a step = one activation slot, or the residual slots of a tile) hands the
//     step's slots to the MFMA waves and the previous step's slots back to the loaders;
//   * the epilogue is wave-private: D[channel][pixel] fragments (weights are the MFMA "A" operand) -> 16-byte
//     ds_writes into the wave's own fp32 patch -> bias / residual / ReLU -> 16-byte NHWC stores.
//
// The rarely used fused extras (bilinear add of the low-resolution up_conv, post-ReLU skip addends) are read
// directly in the epilogue (FULL variant); they pay the vmcnt stall described above.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

#ifndef SMAP_ABLATE
#define SMAP_ABLATE 0      // diagnostics builds only (tools/build_ablate.py): 1 no loads, 2 no ds_read/MFMA, 4 no stores, 8 no epilogue
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void wait_vm(int n)                  // n = a multiple of 4, at most 60
{
    switch (n >> 2) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k * 4) : "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

constexpr int WS_BM = 64, WS_BN = 256;

// RES : the residual tensor goes through the ring (4 extra slots per tile);  FULL : + direct-read fused extras
template <int KC, int STAGES, int NLOAD, bool RES, bool FULL>
__global__ __launch_bounds__(256 + 64 * NLOAD) void conv1x1_ws_kernel(const ConvArgs a, int groups)
{
    constexpr int BM = WS_BM;
    constexpr int SLOT = BM * 128;                              // 8 KB: 64 rows x 128 bytes
    constexpr int IPS = 8 / NLOAD;                              // DMA instructions per slot per loader wave
    constexpr int RS = RES ? 4 : 0;                             // residual slots per tile (one per MFMA wave)
    constexpr int TS = KC + RS;                                 // slots per tile
    static_assert(NLOAD == 1 || NLOAD == 2, "one or two loader waves");
    static_assert(STAGES > (RS > 1 ? RS : 1) && (STAGES - 1) * IPS <= 60, "ring depth vs the 6-bit vmcnt");
    constexpr int PP = 68;                                      // patch pitch in floats: 64 channels + 4 (conflict-free b128 rows)
    constexpr int PATCH = 32 * PP * 4;                          // per-wave fp32 epilogue patch: 32 pixels x 64 channels
    constexpr int LDS_BYTES = STAGES * SLOT + 4 * PATCH;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (n_tile, group): the n tiles of one group sit on the same XCD (they read the same rows)
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int n_tile = loc % a.n_tiles;
    const int g = (loc / a.n_tiles) * 8 + xcd;
    const int n_my = g < a.m_tiles ? (a.m_tiles - g + groups - 1) / groups : 0;
    if (n_my == 0) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader wave(s)
        // Slot stream of a tile: A(0) .. A(KC-1), then R(0) .. R(3).  Steps: every A slot is a step, the R slots
        // of a tile form one step.  At step s the loader waits until the step's slots have landed (the loads
        // still in flight are exactly the slots issued beyond the step), meets the MFMA waves at the barrier --
        // which also tells it that step s-1 has been consumed -- and tops the ring up to STAGES slots counted
        // from the first slot of step s.
        const int ld = wave - 4;
        const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
        const char* __restrict__ resb = reinterpret_cast<const char*>(a.res);
        const int lrow = lane >> 3, lslot = lane & 7;
        const int total = n_my * TS;
        int j = 0, k = 0;                                       // cursor of the next slot to request: tile j, slot k of TS
        unsigned a_off[IPS];                                    // per-lane byte offsets of the cursor tile's rows (0 = zero page)
        long long r_off[IPS];                                   // residual: byte offset of (row, channel granule lslot of wave 0), -1 = none
        auto set_tile = [&]() {
            const int m0 = (g + j * groups) * BM;
#pragma unroll
            for (int i = 0; i < IPS; ++i) {
                const int row = ld * (BM / NLOAD) + i * 8 + lrow;
                const int m = m0 + row;
                const int gch = lslot ^ ((row >> 1) & 7);
                a_off[i] = m < a.M ? (unsigned)(a.in_off + ((long long)m * a.in_stride_c + a.in_c_off + gch * 8) * 2) : 0u;
                r_off[i] = m < a.M ? ((long long)m * a.Cout8 + n_tile * WS_BN + lslot * 8) * 2 : -1;
            }
        };
        set_tile();
        int wslot = 0, issued = 0;
        auto issue = [&]() {
            char* s = smem + wslot * SLOT + ld * (SLOT / NLOAD);
            if (!(SMAP_ABLATE & 1)) {
                if (!RES || k < KC) {
                    const char* gA = arena + k * 128;
#pragma unroll
                    for (int i = 0; i < IPS; ++i)
                        __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_off[i]), (lds_void*)(s + i * 1024), 16, 0, 0);
                } else {
                    const int w = k - KC;                       // residual slot of MFMA wave w: channels n_tile*256 + w*64 ..
                    const bool n_ok = n_tile * WS_BN + w * 64 + lslot * 8 < a.Cout8;
#pragma unroll
                    for (int i = 0; i < IPS; ++i) {
                        const char* src = (r_off[i] >= 0 && n_ok) ? resb + r_off[i] + w * 128 : arena;
                        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(s + i * 1024), 16, 0, 0);
                    }
                }
            }
            wslot = wslot + 1 == STAGES ? 0 : wslot + 1;
            ++issued;
            if (++k == TS) { k = 0; ++j; set_tile(); }          // past the last tile: offsets nobody uses
        };
        while (issued < total && issued < STAGES) issue();
        int start = 0;                                          // first slot of the current step
        for (int t = 0; t < n_my; ++t) {
#pragma unroll
            for (int st = 0; st < KC + (RES ? 1 : 0); ++st) {
                const int n_s = st < KC ? 1 : RS;
                wait_vm((issued - (start + n_s)) * IPS);        // the step's slots have landed
                __builtin_amdgcn_s_barrier();
                while (issued < total && issued < start + STAGES) issue();
                start += n_s;
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nw0 = n_tile * WS_BN + wave * 64;                 // first output channel of this wave
    // Weights are the MFMA "A" operand, pixels the "B" operand: D[channel][pixel], i.e. lane l31 = pixel and each
    // group of four accumulator registers = four CONSECUTIVE channels -> the epilogue moves 16-byte pieces.
    half8 breg[KC * 4][2];
#pragma unroll
    for (int ks = 0; ks < KC * 4; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            breg[ks][ni] = *reinterpret_cast<const half8*>(a.w + (long long)(nw0 + ni * 32 + l31) * a.K + ks * 16 + lhi * 8);

    float* patch = reinterpret_cast<float*>(smem + STAGES * SLOT + wave * PATCH);
    const int rswz = (l31 >> 1) & 7;
    const int prow = lane >> 3, pcg = lane & 7;                 // epilogue: 8 lanes x 8 channels per pixel, 8 pixels per pass
    const int n = nw0 + pcg * 8;
    const bool n_ok = n < a.Cout8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = a.bias[n + e];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // weights and bias are in: no vmcnt wait below this line
    const int HoWo = a.Ho * a.Wo;
    int rslot = 0;
    for (int j = 0; j < n_my; ++j) {
        const int m0 = (g + j * groups) * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sA = smem + rslot * SLOT;
            rslot = rslot + 1 == STAGES ? 0 : rslot + 1;
#pragma unroll
            for (int kk = 0; kk < ((SMAP_ABLATE & 2) ? 0 : 4); ++kk) {
                const int gq = kk * 2 + lhi;
                half8 af[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    af[mi] = *reinterpret_cast<const half8*>(sA + (mi * 32 + l31) * 128 + ((gq ^ rswz) << 4));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[kc * 4 + kk][ni], af[mi], acc[mi][ni], 0, 0, 0);
            }
        }
        const char* sR = nullptr;                               // this wave's residual slot: [64 rows][64 ch] fp16
        if (RES) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int rs = rslot + wave;
            rs = rs >= STAGES ? rs - STAGES : rs;
            sR = smem + rs * SLOT;
            rslot += RS;
            rslot = rslot >= STAGES ? rslot - STAGES : rslot;
        }

        if (SMAP_ABLATE & 8) {
            if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(a.out)[tid] = acc[1][1][1];   // keep acc live
            continue;
        }
        // ---- wave-private epilogue, 32 pixels at a time
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // fragments -> patch[pixel][channel]: register group q of fragment ni holds channels ni*32 + 8q + 4*lhi .. +3
            // of pixel l31.  LDS operations of one wave execute in order, so the reads below need no barrier.
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(patch + l31 * PP + ni * 32 + q * 8 + lhi * 4) =
                        make_float4(acc[h][ni][q * 4], acc[h][ni][q * 4 + 1], acc[h][ni][q * 4 + 2], acc[h][ni][q * 4 + 3]);
            // this lane's 4 pixels of the half: rows p*8 + prow, channels n .. n+7
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = p * 8 + prow;
                const int m = m0 + h * 32 + row;
                const bool ok = m < a.M && n_ok;
                half8 ra1, ra2, t00, t01, t10, t11;
                float wy0 = 0.f, wy1 = 0.f, wx0 = 0.f, wx1 = 0.f;
                if (FULL) {
                    const int ms = ok ? m : 0, ns = ok ? n : 0;
                    const long long dense = (long long)ms * a.Cout8 + ns;
                    if (a.up) {
                        const int b = ms / HoWo, rem = ms - b * HoWo;
                        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                        const Lerp ly = lerp_index(oy, a.up_h, a.Ho), lx = lerp_index(ox, a.up_w, a.Wo);
                        const _Float16* tb = a.up + (long long)b * a.up_h * a.up_w * a.Cout8 + ns;
                        t00 = *reinterpret_cast<const half8*>(tb + ((long long)ly.i0 * a.up_w + lx.i0) * a.Cout8);
                        t01 = *reinterpret_cast<const half8*>(tb + ((long long)ly.i0 * a.up_w + lx.i1) * a.Cout8);
                        t10 = *reinterpret_cast<const half8*>(tb + ((long long)ly.i1 * a.up_w + lx.i0) * a.Cout8);
                        t11 = *reinterpret_cast<const half8*>(tb + ((long long)ly.i1 * a.up_w + lx.i1) * a.Cout8);
                        wy0 = ly.l0; wy1 = ly.l1; wx0 = lx.l0; wx1 = lx.l1;
                    }
                    if (a.add1) ra1 = *reinterpret_cast<const half8*>(a.add1 + dense);
                    if (a.add2) ra2 = *reinterpret_cast<const half8*>(a.add2 + dense);
                }
                const float4 lo = *reinterpret_cast<const float4*>(patch + row * PP + pcg * 8);
                const float4 hi = *reinterpret_cast<const float4*>(patch + row * PP + pcg * 8 + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias[e];
                if (RES) {
                    const half8 rr = *reinterpret_cast<const half8*>(sR + (h * 32 + row) * 128 + pcg * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
                }
                if (FULL && a.up) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[e] += wy0 * (wx0 * (float)t00[e] + wx1 * (float)t01[e]) +
                                wy1 * (wx0 * (float)t10[e] + wx1 * (float)t11[e]);
                }
                if (FULL) {
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    if (a.add1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)ra1[e];
                    }
                    if (a.add2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)ra2[e];
                    }
                }
                if (ok && !((SMAP_ABLATE & 4) && a.M != 7)) {
                    const long long o = (long long)m * a.out_stride_c + a.out_c_off + n;
                    if (a.out_fp32) {
                        if (!FULL && a.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                        float* op = reinterpret_cast<float*>(a.out) + o;
                        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        half8 hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)v[e];
                        if (!FULL && a.relu) {                  // ReLU commutes with the (monotonic) rounding: 4 packed max
                            half2v* h2 = reinterpret_cast<half2v*>(&hv);
                            const half2v z = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                            for (int e = 0; e < 4; ++e) h2[e] = __builtin_elementwise_max(h2[e], z);
                        }
                        *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(a.out) + o) = hv;
                    }
                }
            }
        }
    }
}

template <int KC, int STAGES, int NLOAD>
hipError_t launch_ws(const ConvArgs& a, hipStream_t st)
{
    // persistent grid: one workgroup per CU, a multiple of 8 * n_tiles, never more groups than row tiles,
    // and (almost) the same tile count for every group
    int groups = 256 / a.n_tiles;
    if (groups > a.m_tiles) groups = a.m_tiles;
    groups = (groups + 7) & ~7;
    const int per = (a.m_tiles + groups - 1) / groups;
    groups = (((a.m_tiles + per - 1) / per) + 7) & ~7;
    const dim3 grid(groups * a.n_tiles), block(256 + 64 * NLOAD);
    const bool full = a.up || a.add1 || a.add2;
    if (a.res) {
        if (full) hipLaunchKernelGGL((conv1x1_ws_kernel<KC, STAGES, NLOAD, true, true>), grid, block, 0, st, a, groups);
        else hipLaunchKernelGGL((conv1x1_ws_kernel<KC, STAGES, NLOAD, true, false>), grid, block, 0, st, a, groups);
    } else {
        if (full) hipLaunchKernelGGL((conv1x1_ws_kernel<KC, STAGES, NLOAD, false, true>), grid, block, 0, st, a, groups);
        else hipLaunchKernelGGL((conv1x1_ws_kernel<KC, STAGES, NLOAD, false, false>), grid, block, 0, st, a, groups);
    }
    return hipGetLastError();
}

}  // namespace

// tile ids 40..41: weight-stationary 1x1 (64-row tiles, 256 output channels per workgroup)
int smap_conv1_tile_dims(int tile, int* bm, int* bn)
{
    if (tile != 40 && tile != 41) return -1;
    *bm = WS_BM;
    *bn = WS_BN;
    return 0;
}

// 1x1 stride-1 convs with Cin in {64, 128, 256} only (plan validation rejects other ops for these tile ids).
hipError_t smap_launch_conv1(const ConvArgs& a, int tile, hipStream_t st)
{
    if (a.ksize != 1 || a.stride != 1 || a.pad != 0) return hipErrorInvalidValue;
    const bool two = tile == 41;      // 40: one loader wave, 8 slots (7 x 8 KB in flight); 41: two loader waves, 14 slots
    switch (a.Cin) {
        case 64: return two ? launch_ws<1, 14, 2>(a, st) : launch_ws<1, 8, 1>(a, st);
        case 128: return two ? launch_ws<2, 14, 2>(a, st) : launch_ws<2, 8, 1>(a, st);
        case 256: return two ? launch_ws<4, 14, 2>(a, st) : launch_ws<4, 8, 1>(a, st);
        default: return hipErrorInvalidValue;
    }
}
