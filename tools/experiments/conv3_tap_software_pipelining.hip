// EXPERIMENT (not built, not shipped): conv3.hip with 8-wave workgroups and cross-tap software pipelining of the
// fragment reads (MFMAs of tap t on registers read during tap t-1).  Parity-green but 10-35 % SLOWER than the
// interleaved loop of smap_amd/csrc/conv3.hip on every 3x3 shape (profiles/r1_v9_halo_experiments.log); kept as the
// starting point for a hand-scheduled (inline-asm ds_read + manual lgkmcnt) version.  See DESIGN.md section 9.
// conv3.hip -- 3x3 stride-1 pad-1 conv_bn_relu (model/smap.py:13-45, the Bottleneck 3x3s and the
// res_* heads) as a HALO-TILED implicit GEMM for gfx950.
//
// Why a second 3x3 path: the im2col view of conv.hip fetches every input element nine times, once
// per tap, through the L2 -> LDS stream that bounds these layers (loads alone are
// 70-90 % of their time).  Here a workgroup owns a 2-D tile of TH x TW = 128 output pixels; per
// 64-channel chunk it brings the (TH+2) x (TW+2) input patch into LDS ONCE and serves all nine taps
// from it (1.4-1.6x the tile instead of 9x); only the 8-16 KB weight tile changes per tap.
//
//   K order   : channel chunk cc (outer), tap (kh,kw) (inner) -- weights stay [cout][kh][kw][cin]
//   LDS       : A patch  2 x PROWS rows x 128 B (row = patch pixel, 16-B slot s holds granule
//               s ^ ((row>>1)&7), written by LDS-DMA exactly like conv.hip's tile rows)
//               B tile   NB x BN rows x 128 B
//   MFMA      : v_mfma_f32_32x32x16_f16; the A fragment of tap (kh,kw) for tile pixel (py,px) is
//               patch row (py+kh)*(TW+2) + px+kw -- a shifted view, no data movement
//   pipeline  : iteration = (cc, tap), software-pipelined: the MFMAs of tap t run on fragments read during tap
//               t-1 while the fragments of tap t+1 come out of LDS and NB-2 weight tiles (plus, from tap 0 of a
//               chunk, the next patch) are in flight (counted vmcnt + raw s_barrier)
//   epilogue  : fp32 LDS tile -> bias, ReLU -> 16-byte NHWC stores (fp16, or fp32 for the heads)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

#ifndef SMAP_ABLATE
#define SMAP_ABLATE 0      // diagnostics builds only (tools/build_ablate.py): 1 no LDS-DMA, 2 no MFMA, 8 no epilogue, 16 no ds_read
#endif

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// s_waitcnt vmcnt(n) for a value that is a constant after unrolling (the switch folds away)
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

// NW = 4 or 8 waves.  One wave issues a 32x32x16 MFMA only every ~80 cycles while the pipe takes one every ~38
// (tools/ubench/mfma_clock.hip): the 8-wave variants halve the MFMAs per wave so that the low-resolution layers,
// which put one or two workgroups on a CU, still keep two to four multiplying waves on every SIMD.
template <int BN, int TW, int NB, int NW>
__global__ __launch_bounds__(64 * NW) void conv3x3_halo_kernel(const ConvArgs a, int tiles_x, int tiles_y)
{
    constexpr int BM = 128, TH = BM / TW, PW = TW + 2, PH = TH + 2;
    constexpr int NT = 64 * NW, RPR = 8 * NW;                   // threads; rows per DMA round (8 rows per wave instruction)
    constexpr int PROWS = ((PH * PW + RPR - 1) / RPR) * RPR;    // patch rows rounded to a DMA round
    constexpr int LA = PROWS / RPR, LB = BN / RPR;
    static_assert(BN % RPR == 0, "weight tile vs DMA round");
    constexpr int ROWB = 128;
    constexpr int A_BYTES = PROWS * ROWB, B_BYTES = BN * ROWB;
    static_assert(NB >= 2 && NB <= 8 && (NB - 2) * LB + LA <= 24, "vmcnt is 6 bits; wait_vm handles 0..24");
    constexpr int PIPE = 2 * A_BYTES + NB * B_BYTES;
    constexpr int LDS_BYTES = PIPE > BM * BN * 4 ? PIPE : BM * BN * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    // wave grid: 4 waves = 2 x 2 (64 px x BN/2; 4 x 1 for BN = 32); 8 waves = 2 x 4 for BN = 128, 4 x 2 for BN = 64
    constexpr int WN = NW == 8 ? (BN >= 128 ? 4 : 2) : (BN >= 64 ? 2 : 1), WM = NW / WN;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    static_assert(NI >= 1 && MI >= 1, "BN >= 32");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SMAP_TRACE
    long long tr_t[6]; long long tr_vm = 0, tr_bar = 0;
    tr_t[0] = __builtin_amdgcn_s_memtime();
#define TR3(i) tr_t[i] = __builtin_amdgcn_s_memtime()
#else
#define TR3(i)
#endif

    int logical;                                                // XCD-aware order, n tile fastest
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int n_tile = logical % a.n_tiles;
    int t = logical / a.n_tiles;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = n_tile * BN;

    // ---- staging offsets (uniform base + 32-bit lane offset; 0 = zero page of the arena)
    const int lrow = lane >> 3, lslot = lane & 7;
    const int srow = wave * 8 + lrow;
    const int gch = lslot ^ ((srow >> 1) & 7);                  // (prow>>1)&7 == (srow>>1)&7: rounds are 32 or 64 rows
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ wt = reinterpret_cast<const char*>(a.w);

    unsigned b_off[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) b_off[i] = (unsigned)(((n0 + i * RPR + srow) * a.K + gch * 8) * 2);
    auto issue_b = [&](int buf, unsigned boff) {
        char* sB = smem + 2 * A_BYTES + buf * B_BYTES;
        const char* gB = wt + boff;
        if (SMAP_ABLATE & 1) return;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + b_off[i]), (lds_void*)(sB + (i * RPR + wave * 8) * ROWB), 16, 0, 0);
    };
    issue_b(0, 0);                                              // weights of (cc 0, tap 0): no pixel math needed

    unsigned a_off[LA];                                         // patch pixel -> byte offset of its channel granule gch
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int prow = i * RPR + srow;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        a_off[i] = 0;
        if (prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const long long e = ((long long)(b * a.H + iy) * a.W + ix) * a.in_stride_c + a.in_c_off + gch * 8;
            a_off[i] = (unsigned)(a.in_off + e * 2);
        }
    }
    auto issue_a = [&](int buf, int cc) {
        char* sA = smem + buf * A_BYTES;
        const char* gA = arena + (unsigned)(cc * ROWB);         // invalid pixels: zero page + cc*128
        if (SMAP_ABLATE & 1) return;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_off[i]), (lds_void*)(sA + (i * RPR + wave * 8) * ROWB), 16, 0, 0);
    };
    issue_a(0, 0);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    // tile pixel of this lane's A fragment rows: p = wm*64 + mi*32 + l31 -> (py, px)
    int prow0[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        prow0[mi] = (p / TW) * PW + (p % TW);                   // + kh*PW + kw per tap
    }
    const int b_row0 = wn * (BN / WN) + l31;
    const int bswz = (l31 >> 1) & 7;

    // ---- pipeline, software-pipelined across taps.  Flat iteration it = cc*9 + tap MULTIPLIES the fragments of
    //      tap `it` (already in registers) while the fragments of tap it+1 are read from LDS into the other
    //      register set and weight tiles it+2 .. it+NB-1 are still in flight: neither the LDS latency nor the
    //      L2 latency sits between two MFMA groups of a wave (ablations: MFMA alone 8.6 us of a 26.8 us layer3
    //      3x3; the serial ds_read phase cost 4.2 us, the exposed weight-tile waits 3.4 us).
    //      At the barrier of iteration it: tile it+1 has landed (counted vmcnt: loads retire in issue order) and
    //      every wave has finished READING tile it (that happened in iteration it-1), so its stage takes tile
    //      it+NB.  Issue order per iteration: B(it+NB), then at tap 0 the next patch A(cc+1) -- A(cc+1) is younger
    //      than B(it+1) exactly while 1 <= tap <= NB-1.
    const int cchunks = a.Cin / 64;
    const int n_iter = cchunks * 9;
    auto tile_off = [&](int itn) {                              // byte offset of weight tile itn inside a weight row
        const int c = itn / 9, t = itn - c * 9;
        return (unsigned)((t * a.Cin + c * 64) * 2);
    };
    auto read_frags = [&](half8 (&af)[4][MI], half8 (&bf)[4][NI], int itn) {
        const int c = itn / 9, t = itn - c * 9;
        const int kh = (t * 11) >> 5, kw = t - kh * 3;          // t / 3 for t < 9
        const int shift = kh * PW + kw;
        const char* sA = smem + (c & 1) * A_BYTES;
        const char* sB = smem + 2 * A_BYTES + (itn % NB) * B_BYTES;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int prow = prow0[mi] + shift;
            const char* rowp = sA + prow * ROWB;
            const int swz = (prow >> 1) & 7;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (SMAP_ABLATE & 16) { for (int e = 0; e < 8; ++e) af[kk][mi][e] = (_Float16)(float)(lane + kk); continue; }
                af[kk][mi] = *reinterpret_cast<const half8*>(rowp + (((kk * 2 + lhi) ^ swz) << 4));
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (SMAP_ABLATE & 16) { for (int e = 0; e < 8; ++e) bf[kk][ni][e] = (_Float16)(float)(t + kk); continue; }
                bf[kk][ni] = *reinterpret_cast<const half8*>(sB + (b_row0 + ni * 32) * ROWB + (((kk * 2 + lhi) ^ bswz) << 4));
            }
    };
    auto multiply = [&](half8 (&af)[4][MI], half8 (&bf)[4][NI]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (SMAP_ABLATE & 2) { acc[mi][ni][kk] += (float)af[kk][mi][0] + (float)bf[kk][ni][1]; continue; }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][mi], bf[kk][ni], acc[mi][ni], 0, 0, 0);
                }
    };
    // one iteration (it < n_iter - 1): wait + barrier, refill, read the NEXT tap's fragments into (afn, bfn), multiply
    // (afc, bfc).  No branch may enclose the ds_reads: at a control-flow join the compiler's waitcnt pass falls back to
    // lgkmcnt(0) and the multiply would wait for the fragments it does not need.
    auto step = [&](int it, half8 (&afc)[4][MI], half8 (&bfc)[4][NI], half8 (&afn)[4][MI], half8 (&bfn)[4][NI]) {
        const int cc = it / 9, tap = it - cc * 9;
#ifdef SMAP_TRACE
        const long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        const int hi = it + NB - 1 < n_iter - 1 ? it + NB - 1 : n_iter - 1;    // youngest weight tile issued so far
        int younger = (hi - (it + 1)) * LB;
        if (tap >= 1 && tap <= NB - 1 && cc + 1 < cchunks) younger += LA;
        wait_vm(younger);
#ifdef SMAP_TRACE
        const long long tw1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SMAP_TRACE
        tr_vm += tw1 - tw0;
        tr_bar += __builtin_amdgcn_s_memtime() - tw1;
#endif
        if (it + NB < n_iter) issue_b(it % NB, tile_off(it + NB));
        if (tap == 0 && cc + 1 < cchunks) issue_a((cc + 1) & 1, cc + 1);
        // MFMAs of K step kk first, then the next tap's reads of the same K step: the only lgkmcnt wait the compiler
        // needs is the one in front of the first MFMA (fragments requested an iteration ago); the new reads land
        // under the remaining MFMAs and the next barrier.
        const int itn = it + 1;
        const int c1 = itn / 9, t1 = itn - c1 * 9;
        const int kh = (t1 * 11) >> 5, kw = t1 - kh * 3;
        const int shift = kh * PW + kw;
        const char* sA = smem + (c1 & 1) * A_BYTES;
        const char* sB = smem + 2 * A_BYTES + (itn % NB) * B_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (SMAP_ABLATE & 2) { acc[mi][ni][kk] += (float)afc[kk][mi][0] + (float)bfc[kk][ni][1]; continue; }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[kk][mi], bfc[kk][ni], acc[mi][ni], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int prow = prow0[mi] + shift;
                if (SMAP_ABLATE & 16) { for (int e = 0; e < 8; ++e) afn[kk][mi][e] = (_Float16)(float)(lane + kk); continue; }
                afn[kk][mi] = *reinterpret_cast<const half8*>(sA + prow * ROWB + (((kk * 2 + lhi) ^ ((prow >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if (SMAP_ABLATE & 16) { for (int e = 0; e < 8; ++e) bfn[kk][ni][e] = (_Float16)(float)(t1 + kk); continue; }
                bfn[kk][ni] = *reinterpret_cast<const half8*>(sB + (b_row0 + ni * 32) * ROWB + (((kk * 2 + lhi) ^ bswz) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // prologue: tiles 1 .. NB-1 (tile 0 and patch 0 are already on their way), then the fragments of tap 0
#pragma unroll
    for (int d = 1; d < NB; ++d)
        if (d < n_iter) issue_b(d, tile_off(d));
    TR3(1);
    half8 af0[4][MI], bf0[4][NI], af1[4][MI], bf1[4][NI];
    wait_vm(((NB - 1 < n_iter - 1 ? NB - 1 : n_iter - 1)) * LB);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TR3(2);
    read_frags(af0, bf0, 0);
    int it = 0;
    for (; it + 2 <= n_iter - 1; it += 2) {                     // n_iter - 1 pipelined steps, two per trip
        step(it, af0, bf0, af1, bf1);
        step(it + 1, af1, bf1, af0, bf0);
    }
    if (it < n_iter - 1) {
        step(it, af0, bf0, af1, bf1);
        multiply(af1, bf1);
    } else {
        multiply(af0, bf0);
    }
    (void)n_iter;
    TR3(3);
    __syncthreads();
    if (SMAP_ABLATE & 8) {
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(a.out)[tid] = acc[MI - 1][NI - 1][1];   // keep acc live
        return;
    }

    // ---- epilogue: acc + bias -> fp32 [128][BN] LDS tile -> ReLU -> 16-byte NHWC stores
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = wn * (BN / WN) + ni * 32 + l31;
        const float bias = a.bias[n0 + col];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (MI * 32) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                Cs[row * BN + col] = acc[mi][ni][r] + bias;
            }
    }
    __syncthreads();
    TR3(4);
    constexpr int CG = BN / 8, PASSES = BM * CG / NT;
    static_assert(BM * CG % NT == 0, "epilogue passes");
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int idx = p * NT + tid;
        const int row = idx / CG, cg = idx - row * CG;
        const int oy = oy0 + row / TW, ox = ox0 + row % TW, n = n0 + cg * 8;
        if (oy >= a.Ho || ox >= a.Wo || n >= a.Cout8) continue;
        const float4 lo = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + row * BN + cg * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        const long long m = ((long long)b * a.Ho + oy) * a.Wo + ox;
        const long long o = m * a.out_stride_c + a.out_c_off + n;
        if (a.out_fp32) {
            float* op = reinterpret_cast<float*>(a.out) + o;
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(a.out) + o) = h;
        }
    }
#ifdef SMAP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR3(5);
    if (a.dbg && tid == 0) {
        long long* d = a.dbg + (long long)blockIdx.x * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        d[0] = tr_t[0]; d[1] = tr_t[1]; d[2] = tr_t[2]; d[3] = tr_t[3]; d[4] = tr_t[4]; d[5] = tr_t[5];
        d[6] = tr_vm | (tr_bar << 32); d[7] = hwid;
    }
#endif
}

template <int BN, int TW, int NB, int NW = 4>
hipError_t launch3(const ConvArgs& a, hipStream_t st)
{
    constexpr int TH = 128 / TW;
    const int B = a.M / (a.Ho * a.Wo);
    const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, TW, NB, NW>), dim3(tiles_x * tiles_y * B * a.n_tiles), dim3(64 * NW), 0, st, a, tiles_x,
                       tiles_y);
    return hipGetLastError();
}

}  // namespace

// tile ids 30..39 and 50..57: halo-tiled 3x3 (BM is always 128 output pixels)
int smap_conv3_tile_dims(int tile, int* bm, int* bn)
{
    switch (tile) {
        case 30: case 32: case 34: case 36: *bm = 128; *bn = 64; return 0;       // 8x16 / 4x32 pixel tiles
        case 31: case 33: case 35: case 37: *bm = 128; *bn = 128; return 0;
        case 38: case 39: *bm = 128; *bn = 32; return 0;
        case 50: case 52: case 54: case 56: *bm = 128; *bn = 64; return 0;       // 50..57: ids 30..37 with 8 waves
        case 51: case 53: case 55: case 57: *bm = 128; *bn = 128; return 0;
        default: return -1;
    }
}

// Only plain 3x3 stride-1 convs qualify (no residual / addends / bilinear add): the schedule's Bottleneck
// 3x3s and head convs.  Returns hipErrorInvalidValue otherwise (plan validation rejects such ops earlier).
hipError_t smap_launch_conv3(const ConvArgs& a, int tile, hipStream_t st)
{
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.res || a.add1 || a.add2 || a.up) return hipErrorInvalidValue;
    switch (tile) {
        // <BN, TW, NB[, waves]>: NB weight-tile stages = NB-2 tiles in flight beyond the one being read; LDS in KiB
        case 30: return launch3<64, 16, 3>(a, st);      //  72
        case 31: return launch3<128, 16, 3>(a, st);     //  96
        case 32: return launch3<64, 32, 3>(a, st);      //  80
        case 33: return launch3<128, 32, 3>(a, st);     // 104
        case 34: return launch3<64, 16, 4>(a, st);      //  80
        case 35: return launch3<128, 16, 4>(a, st);     // 112
        case 36: return launch3<64, 32, 2>(a, st);      //  72
        case 37: return launch3<128, 32, 4>(a, st);     // 120
        case 38: return launch3<32, 16, 5>(a, st);      //  68: Cout <= 32 heads, 4 x 1 waves
        case 39: return launch3<32, 32, 5>(a, st);      //  76
        case 50: return launch3<64, 16, 3, 8>(a, st);
        case 51: return launch3<128, 16, 3, 8>(a, st);
        case 52: return launch3<64, 32, 3, 8>(a, st);
        case 53: return launch3<128, 32, 3, 8>(a, st);
        case 54: return launch3<64, 16, 4, 8>(a, st);
        case 55: return launch3<128, 16, 4, 8>(a, st);
        case 56: return launch3<64, 32, 2, 8>(a, st);
        case 57: return launch3<128, 32, 4, 8>(a, st);
        default: return hipErrorInvalidValue;
    }
}
