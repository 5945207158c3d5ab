// convf.hip -- the TAIL of a Bottleneck in ONE launch (model/smap.py:48-77): conv_bn_relu2 (3x3, stride 1) followed by
// conv_bn_relu3 (1x1, planes -> 4*planes) + residual + ReLU (+ the two skip adds of smap.py:142-153).
//
// Why: the 3x3's output is the smallest tensor of the block, but as two launches it is written once and read once through
// the fabric, and every launch of this schedule is bound by the bytes it moves.  Here the 128-pixel x
// P-channel result of the 3x3 never leaves the CU: phase 1 is conv3.hip's halo-tiled main loop (weights as the FIRST MFMA
// operand, so a lane ends up with 4 consecutive channels of one pixel); its bias + ReLU output is split into hi/lo fp16
// and written into LDS in the row format the MFMA fragment reads expect; phase 2 multiplies that tile by the 1x1 weights
// in chunks of BN2 output channels (weight chunks double-buffered by LDS-DMA) and finishes each chunk with convp.hip's
// REGISTER epilogue (v_permlane32_swap -> 8 consecutive channels per lane: 16-byte NHWC loads of the residual / skip
// tensors and stores of both planes, no LDS transpose).
//
//   LDS phase 1 : 2 patches x PROWS x 128 B + NB weight tiles x P x 128 B           (conv3.hip)
//   LDS phase 2 : A2 [P/CH][128 px][128 B] + 2 x W2 chunk [P/CH][BN2][128 B], aliasing phase 1's buffers
//   weights     : 3x3 as conv3.hip's blocks ([chunk][tap][P rows][128 B]); 1x1 as [n2 chunk][k chunk][BN2 rows][128 B]
//                 with the same row format (smap_amd/engine.py::pack_halo_rows)
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

__device__ __forceinline__ void wait_vm(int n)
{
    switch (n) {
#define W_(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16)
        W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24)
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// (Phase-2 scheduling variants measured in situ and dropped, profiles/r3_ab_tail_variants.log: first weight chunk prefetched
// under the 3x3's last channel chunk: +-0; chunk barrier before the stores and the residual requested a chunk ahead: -0.5 %.)
template <int P, int TW, int NB, int BN2, bool X3>
__global__ __launch_bounds__(256) void conv3_tail_kernel(const ConvArgs a, int tiles_x, int tiles_y)
{
    constexpr int CH = X3 ? 32 : 64;                            // channels per 128-byte LDS row
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int BM = 128, TH = BM / TW, PW = TW + 2, PH = TH + 2;
    constexpr int PROWS = ((PH * PW + 31) / 32) * 32;
    constexpr int LA = PROWS / 32, LB = P / 32;
    constexpr int ROWB = 128;
    constexpr int A_BYTES = PROWS * ROWB, B_BYTES = P * ROWB;
    constexpr int D = NB - 1;
    static_assert(D >= 1 && D <= 8 && (D - 1) * LB + LA <= 63, "vmcnt is 6 bits");
    constexpr int PIPE = 2 * A_BYTES + NB * B_BYTES;
    constexpr int KC2 = P / CH;                                 // K chunks of the 1x1
    constexpr int A2_BYTES = KC2 * BM * ROWB;
    constexpr int W2_CHUNK = KC2 * BN2 * ROWB, L2R = W2_CHUNK / 4096;
    constexpr int PH2 = A2_BYTES + 2 * W2_CHUNK;
    constexpr int LDS_BYTES = PIPE > PH2 ? PIPE : PH2;
    static_assert(LDS_BYTES <= 160 * 1024 && W2_CHUNK % 4096 == 0, "LDS");
    constexpr int MI = 2, NI1 = P / 64, NI2 = BN2 / 64;         // 2 x 2 waves: 64 pixels x (P/2 | BN2/2) channels per wave
    static_assert(NI1 >= 1 && NI2 >= 1, "P, BN2 >= 64");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    SMAP_TL_BEGIN
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int logical;                                                // XCD-aware order (one N tile: P channels)
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int t = logical;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    // ================================================================= phase 1: the 3x3 (conv3.hip's pipeline)
    const int lrow = lane >> 3, lslot = lane & 7;
    const int srow = wave * 8 + lrow;
    const int gl = lslot ^ ((srow >> 1) & 7);
    const int gch = X3 ? (gl & 3) : gl;
    const int gpl = X3 ? (gl >> 2) : 0;
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ wt_tile = reinterpret_cast<const char*>(a.w);
    unsigned w_off[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) w_off[i] = (unsigned)((i * 32 + wave * 8) * ROWB + lane * 16);
    auto issue_b = [&](int buf, int blk) {                      // blk = cc * 9 + tap
        char* sB = smem + 2 * A_BYTES + buf * B_BYTES;
        const char* gB = wt_tile + (long long)blk * B_BYTES;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + w_off[i]), (lds_void*)(sB + (i * 32 + wave * 8) * ROWB), 16, 0, 0);
    };
    issue_b(0, 0);

    unsigned a_off[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int prow = i * 32 + srow;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        a_off[i] = 0;
        if (prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const long long e = ((long long)(b * a.H + iy) * a.W + ix) * a.in_stride_c + a.in_c_off + gch * 8 +
                                (X3 ? gpl * a.in_lo : 0);
            a_off[i] = (unsigned)(a.in_off + e * 2);
        }
    }
    auto issue_a = [&](int buf, int cc) {
        char* sA = smem + buf * A_BYTES;
        const char* gA = arena + (unsigned)(cc * CH * 2);
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_off[i]), (lds_void*)(sA + (i * 32 + wave * 8) * ROWB), 16, 0, 0);
    };
    issue_a(0, 0);

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[NI1][MI];                                        // rows = channels (weights first), columns = pixels
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    int prow0[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * 64 + mi * 32 + l31;
        prow0[mi] = (p / TW) * PW + (p % TW);
    }
    const int b_row0 = wn * (P / 2) + l31;
    const int bswz = (l31 >> 1) & 7;

    const int cchunks = a.Cin / CH;
    constexpr int a2_off = 0, w2_off0 = A2_BYTES, w2_off1 = A2_BYTES + W2_CHUNK;      // phase 2: [A2 | W2 chunk x 2]
    const char* __restrict__ w2 = reinterpret_cast<const char*>(a.w2);
    const unsigned w2_lane = (unsigned)(wave * 1024 + lane * 16);
    auto issue_w2 = [&](int nc) {
        const char* g = w2 + (long long)nc * W2_CHUNK;
        char* s = smem + ((nc & 1) ? w2_off1 : w2_off0) + wave * 1024;
#pragma unroll
        for (int i = 0; i < L2R; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + i * 4096 + w2_lane), (lds_void*)(s + i * 4096), 16, 0, 0);
    };
#pragma unroll
    for (int d = 1; d < D; ++d) issue_b(d, d);
    for (int cc = 0; cc < cchunks; ++cc) {
        const bool last = cc + 1 == cchunks;
        const char* sA = smem + (cc & 1) * A_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int it = cc * 9 + tap;
            if (last) {
                if (tap + D - 1 >= 9) wait_vm(0);
                else wait_vm((D - 1) * LB);
            } else {
                wait_vm((D - 1) * LB + ((tap >= 1 && tap <= D) ? LA : 0));
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {
                const int nt = tap + D;
                const int ncc = cc + nt / 9, ntap = nt % 9;
                if (ncc < cchunks) issue_b((it + D) % NB, ncc * 9 + ntap);
                if (tap == 0 && !last) issue_a((cc + 1) & 1, cc + 1);
            }
            const char* sB = smem + 2 * A_BYTES + (it % NB) * B_BYTES;
            const int shift = (tap / 3) * PW + (tap % 3);
#pragma unroll
            for (int kk = 0; kk < CH / 16; ++kk) {
                const int g = kk * 2 + lhi;
                half8 af[NPL][MI], bf[NPL][NI1];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int prow = prow0[mi] + shift;
                        af[pl][mi] = *reinterpret_cast<const half8*>(sA + prow * ROWB + (((g + 4 * pl) ^ ((prow >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int ni = 0; ni < NI1; ++ni)
                        bf[pl][ni] = *reinterpret_cast<const half8*>(sB + (b_row0 + ni * 32) * ROWB + (((g + 4 * pl) ^ bswz) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if (X3) {       // small cross terms first, then hi*hi (conv3.hip's order)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0][ni], af[NPL - 1][mi], acc[ni][mi], 0, 0, 0);
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[NPL - 1][ni], af[0][mi], acc[ni][mi], 0, 0, 0);
                        }
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0][ni], af[0][mi], acc[ni][mi], 0, 0, 0);
                    }
            }
        }
    }
    __syncthreads();                                            // every wave is done with the patches and the 3x3 weights

    // ================================================================= phase 2: the 1x1 on the tile in LDS
    char* sA2 = smem + a2_off;                                  // [KC2][128 pixels][128 B]
    issue_w2(0);

    // pixel -> output offsets of this lane's columns
    unsigned m_dense[MI], m_out[MI];
    bool m_ok[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * 64 + mi * 32 + l31;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        m_ok[mi] = oy < a.Ho && ox < a.Wo;
        const unsigned m = m_ok[mi] ? (unsigned)((b * a.Ho + oy) * a.Wo + ox) : 0u;
        m_dense[mi] = m * (unsigned)(NPL * a.tail_cout8);
        m_out[mi] = m * (unsigned)a.out_stride_c + (unsigned)a.out_c_off;
    }
    constexpr int NCH = NI2 * MI * 2;                           // (pixel, 8-channel) chunks per lane
    // residual of a chunk: requested before its MFMAs, used after them
    half8 rs[NCH][NPL];
    auto load_res = [&](int nc) {
        const int n_lane = nc * BN2 + wn * (BN2 / 2) + 8 * lhi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n_lane + ni * 32 + 16 * j;
                    const unsigned off = m_dense[mi] + (n < a.tail_cout8 ? (unsigned)n : 0u);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        rs[(mi * NI2 + ni) * 2 + j][pl] = *reinterpret_cast<const half8*>(a.res + off + pl * a.tail_cout8);
                }
    };

    // 3x3 epilogue: relu(acc * scale + bias) -> hi | lo fp16 -> A2.  acc[ni][mi][4*g + e] = channel
    // wn*(P/2) + ni*32 + 8*g + 4*lhi + e of pixel wm*64 + mi*32 + l31: 8 bytes per plane and group.
#pragma unroll
    for (int ni = 0; ni < NI1; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = wn * (P / 2) + ni * 32 + 8 * g + 4 * lhi;
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c);
            const int kc = c / CH, gran = (c % CH) >> 3;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int p = wm * 64 + mi * 32 + l31;
                float v[4] = {acc[ni][mi][4 * g + 0], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                half4 h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = X3 ? v[e] * a.acc_scale + bb[e] : v[e] + bb[e];
                    x = x < 0.f ? 0.f : x;                      // NaN stays NaN (torch's ReLU)
                    h[e] = (_Float16)x;
                    l[e] = (_Float16)(x - (float)h[e]);
                }
                char* row = sA2 + (kc * BM + p) * ROWB + lhi * 8;
                const int sw = (p >> 1) & 7;
                *reinterpret_cast<half4*>(row + ((gran ^ sw) << 4)) = h;
                if (X3) *reinterpret_cast<half4*>(row + (((gran + 4) ^ sw) << 4)) = l;
            }
        }
    wait_vm(0);
    __syncthreads();                                            // A2 written, W2 chunk 0 landed

    const int p_row0 = wm * 64 + l31;                           // + mi*32: pixel rows of A2
    const int c_row0 = wn * (BN2 / 2) + l31;                    // + ni*32: channel rows of the W2 chunk
    const int pswz = (l31 >> 1) & 7;                            // (row >> 1) & 7 of both (row = multiple of 32 + l31)
    _Float16* __restrict__ outp = reinterpret_cast<_Float16*>(a.out);

    for (int nc = 0; nc < a.tail_chunks; ++nc) {
        if (nc + 1 < a.tail_chunks) issue_w2(nc + 1);
        const int n_lane = nc * BN2 + wn * (BN2 / 2) + 8 * lhi;  // + ni*32 + 16*j
        if (a.res) load_res(nc);
        f32x16 acc2[NI2][MI];
#pragma unroll
        for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[ni][mi][r] = 0.f;
        const char* sW = smem + ((nc & 1) ? w2_off1 : w2_off0);
#pragma unroll
        for (int kc = 0; kc < KC2; ++kc)
#pragma unroll
            for (int kk = 0; kk < CH / 16; ++kk) {
                const int g = kk * 2 + lhi;
                half8 pf[NPL][MI], wf[NPL][NI2];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        pf[pl][mi] = *reinterpret_cast<const half8*>(sA2 + (kc * BM + p_row0 + mi * 32) * ROWB + (((g + 4 * pl) ^ pswz) << 4));
#pragma unroll
                    for (int ni = 0; ni < NI2; ++ni)
                        wf[pl][ni] = *reinterpret_cast<const half8*>(sW + (kc * BN2 + c_row0 + ni * 32) * ROWB + (((g + 4 * pl) ^ pswz) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if (X3) {
                            acc2[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ni], pf[NPL - 1][mi], acc2[ni][mi], 0, 0, 0);
                            acc2[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[NPL - 1][ni], pf[0][mi], acc2[ni][mi], 0, 0, 0);
                        }
                        acc2[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ni], pf[0][mi], acc2[ni][mi], 0, 0, 0);
                    }
            }
        // ---- register epilogue (convp.hip): half-wave swap -> acc2[ni][mi][8*j .. 8*j+7] = channels n_lane + ni*32 + 16*j .. +7
#pragma unroll
        for (int ni = 0; ni < NI2; ++ni) {
            float bias[2][8];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 b0 = *reinterpret_cast<const float4*>(a.bias2 + n_lane + ni * 32 + 16 * j);
                const float4 b1 = *reinterpret_cast<const float4*>(a.bias2 + n_lane + ni * 32 + 16 * j + 4);
                bias[j][0] = b0.x; bias[j][1] = b0.y; bias[j][2] = b0.z; bias[j][3] = b0.w;
                bias[j][4] = b1.x; bias[j][5] = b1.y; bias[j][6] = b1.z; bias[j][7] = b1.w;
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xf = acc2[ni][mi][8 * j + e], yf = acc2[ni][mi][8 * j + 4 + e];   // (bit_cast of a vector ELEMENT lvalue reads element 0)
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                        const unsigned s0 = sw[0], s1 = sw[1];
                        acc2[ni][mi][8 * j + e] = (X3 ? a.tail_acc_scale : 1.f) * __uint_as_float(s0) + bias[j][e];
                        acc2[ni][mi][8 * j + 4 + e] = (X3 ? a.tail_acc_scale : 1.f) * __uint_as_float(s1) + bias[j][4 + e];
                    }
        }
        auto add_tensor = [&](const _Float16* __restrict__ tsr) {
            half8 h[NCH][NPL];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = n_lane + ni * 32 + 16 * j;
                        const unsigned off = m_dense[mi] + (n < a.tail_cout8 ? (unsigned)n : 0u);
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            h[(mi * NI2 + ni) * 2 + j][pl] = *reinterpret_cast<const half8*>(tsr + off + pl * a.tail_cout8);
                    }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = (mi * NI2 + ni) * 2 + j;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            acc2[ni][mi][8 * j + e] += X3 ? (float)h[c][0][e] + (float)h[c][NPL - 1][e] : (float)h[c][0][e];
                    }
        };
        if (a.res) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int c = (mi * NI2 + ni) * 2 + j;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            acc2[ni][mi][8 * j + e] += X3 ? (float)rs[c][0][e] + (float)rs[c][NPL - 1][e] : (float)rs[c][0][e];
                    }
        }
        if (a.relu) {
#pragma unroll
            for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[ni][mi][r] = acc2[ni][mi][r] < 0.f ? 0.f : acc2[ni][mi][r];
        }
        if (a.add1) add_tensor(a.add1);
        if (a.add2) add_tensor(a.add2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI2; ++ni)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n_lane + ni * 32 + 16 * j;
                    if (!(m_ok[mi] && n < a.tail_cout8)) continue;
                    _Float16* op = outp + (m_out[mi] + (unsigned)n);
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (_Float16)acc2[ni][mi][8 * j + e];
                    *reinterpret_cast<half8*>(op) = h;
                    if (X3) {
                        half8 l;
#pragma unroll
                        for (int e = 0; e < 8; ++e) l[e] = (_Float16)(acc2[ni][mi][8 * j + e] - (float)h[e]);
                        *reinterpret_cast<half8*>(op + a.out_lo) = l;
                    }
                }
        wait_vm(0);                                             // the next weight chunk (and everything older) has landed
        __syncthreads();                                        // ... for every wave; this chunk's buffer may be refilled
    }
    SMAP_TL_END(a)
}

template <int P, int TW, int NB, int BN2>
hipError_t launchf(const ConvArgs& a, hipStream_t st)
{
    constexpr int TH = 128 / TW;
    const int B = a.M / (a.Ho * a.Wo);
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    if (a.x3)
        hipLaunchKernelGGL((conv3_tail_kernel<P, TW, NB, BN2, true>), dim3(tiles_x * tiles_y * B), dim3(256), 0, st, a, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((conv3_tail_kernel<P, TW, NB, BN2, false>), dim3(tiles_x * tiles_y * B), dim3(256), 0, st, a, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace

// tile ids 80..89: 3x3 (P = BN output channels, all of them in one tile) + 1x1 tail in chunks of *bn2 channels
int smap_convf_tile_dims(int tile, int* bm, int* bn, int* bn2)
{
    switch (tile) {
        case 80: *bm = 128; *bn = 64; *bn2 = 64; return 0;
        case 81: *bm = 128; *bn = 64; *bn2 = 128; return 0;
        case 82: *bm = 128; *bn = 128; *bn2 = 64; return 0;
        default: return -1;
    }
}

hipError_t smap_launch_convf(const ConvArgs& a, int tile, hipStream_t st)
{
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.up || a.out_fp32 || !a.w2) return hipErrorInvalidValue;
    switch (tile) {
        case 80: return launchf<64, 32, 3, 64>(a, st);       // 80 KiB both phases: two workgroups per CU
        case 81: return launchf<64, 32, 3, 128>(a, st);      // 96 KiB
        case 82: return launchf<128, 32, 2, 64>(a, st);      // 128 KiB
        default: return hipErrorInvalidValue;
    }
}
