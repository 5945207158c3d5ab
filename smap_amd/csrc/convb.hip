// convb.hip -- a whole stride-1 identity Bottleneck (model/smap.py:48-77) in ONE launch, split precision:
//     y1 = relu(W1 x + b1)            1x1, C = 4P -> P      (conv_bn_relu1)
//     y2 = relu(W2 * y1 + b2)         3x3 pad 1, P -> P     (conv_bn_relu2)
//     out = relu(W3 y2 + b3 + x)      1x1, P -> C, + the block's input as the residual (+ the two skip adds of smap.py:142-153)
//
// Why: as three launches the block moves 4096 bytes per pixel through the fabric (x read by c1, y1 written and read, y2 written
// and read, x read again as the residual, out written) and every launch of these layers sits on the HBM read+write roof
// (DESIGN.md section 6).  Here x is read ONCE and out written once: 2048 bytes per pixel.  y1 and y2 never leave the CU, and the
// residual never leaves it either: while the K chunks of x pass through the LDS staging buffers of the leading 1x1, every wave
// copies the centre pixels' values it will need in the last epilogue into registers (the register file holds the 64 KB / 128 KB
// an LDS that is full of y1 cannot).
//
// Work per workgroup (4 waves, 80 KiB of LDS, two workgroups per CU so that one streams x while the other multiplies):
// a TH x 16 tile of output pixels (TH = 4 | 8).
//   phase 1  c1 on the (TH+2) x 18 halo patch (rows of the patch = GEMM rows, rounded up to PROWS = 128 | 192): K = C in 16-channel
//            stages, each staged as [PROWS][hi16 | lo16] (x) + [P][hi16 | lo16] (W1) by LDS-DMA in a ring of 6 | 5 stages over the
//            WHOLE 80 KiB (y1 does not exist yet; x comes from HBM: depth is what this phase lives on); accumulators -> relu -> hi/lo -> y1 in conv3.hip's patch layout, rows outside the
//            image forced to 0 (the 3x3 pads y1, not x).  The halo is recomputed: +41 % (+69 %) of c1 = +10 % (+16 %) MFMA work.
//   phase 2  the 3x3 as nine shifted views of y1 (conv3.hip), one 16 KiB weight slot per tap (both 32-channel chunks) in a ring
//            behind y1; accumulators -> relu -> hi/lo -> y2, written over y1 once every wave is done with it.
//   phase 3  the tail 1x1 in four chunks of 64 output channels (weight slots in the same ring) with convp.hip's register
//            epilogue: permlane32_swap -> 8 consecutive channels per lane, + the residual held since phase 1, ReLU, skip adds,
//            16-byte NHWC stores of both planes.
// MFMA operand order is weights FIRST everywhere (D rows = channels, columns = pixels): a lane then owns 4 consecutive
// channels of one pixel, which is what the hi/lo writes into LDS and the register epilogue want.
//
// Weights (smap_amd/engine.py::Graph.conv_block, all three in pack_halo_rows' format: 128-byte rows = [hi32 | lo32] of one
// 32-channel chunk, slot s of row r = logical granule s ^ ((r >> 1) & 7)):
//   W2  [2 chunks][9 taps][64 rows]            W3  [4 n chunks][2 chunks][64 rows]
// and W1 in 16-channel stages (engine.pack_rows16): [16 stages][64 rows][64 B = hi16 | lo16], slot s of row r = granule s ^ ((r >> 2) & 3).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

#ifndef SMAP_CONVB_LDS_KB
#define SMAP_CONVB_LDS_KB 80     // LDS per workgroup (two per CU); experiments: 64
#endif
// 4 x 16 pixel tiles (tile ids 90, 92) need 148 VGPRs: a THIRD workgroup per CU (three waves per SIMD) fits the register file; it
// fits the LDS at 52 KiB per workgroup (y1 32 KiB + a ring of two weight slots; four 12 KiB stages in phase 1).  Experiment of
// round 5 (EXPERIMENTS R5.2): -DSMAP_CONVB_WGS4=3 -DSMAP_CONVB_LDS4_KB=52; the shipped build keeps 2 x 80 KiB.
#ifndef SMAP_CONVB_WGS4
#define SMAP_CONVB_WGS4 2
#endif
#ifndef SMAP_CONVB_LDS4_KB
#define SMAP_CONVB_LDS4_KB SMAP_CONVB_LDS_KB
#endif
// Experiment (EXPERIMENTS R5.2): the two workgroups that share a CU start together and walk through their phases (x streaming, 3x3
// multiplying, tail + stores) in step, so neither fills the other's gaps.  -DSMAP_CONVB_STAGGER_US=<n>: the second half of the FIRST
// wave of workgroups (ids 256 .. 511: the ones that take the CUs' second slots) starts n microseconds late; later workgroups inherit the
// offset, since a slot's next workgroup starts when its predecessor ends.  0 = off (shipped).
#ifndef SMAP_CONVB_STAGGER_US
#define SMAP_CONVB_STAGGER_US 0
#endif
// Experiment (EXPERIMENTS R5.2): -DSMAP_CONVB_SETPRIO=1 raises the wave's issue priority for the duration of every MFMA group (conv3.hip's
// staggered tiles do that), so that the co-resident workgroup's VALU / LDS / LDS-DMA issue cannot delay the matrix pipe.  0 = off (shipped).
#ifndef SMAP_CONVB_SETPRIO
#define SMAP_CONVB_SETPRIO 0
#endif
#if SMAP_CONVB_SETPRIO
#define CONVB_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define CONVB_PRIO(x)
#endif
__device__ __forceinline__ void convb_stagger()
{
#if SMAP_CONVB_STAGGER_US > 0
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
        const long long t0 = __builtin_amdgcn_s_memrealtime();             // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < 100LL * SMAP_CONVB_STAGGER_US) __builtin_amdgcn_s_sleep(64);
    }
#endif
}
#ifndef SMAP_CONVB_ABLATE
#define SMAP_CONVB_ABLATE 0      // diagnostics builds only (tools/build_ablate.py --convb N), identity kernel: 1 no x loads, 2 no MFMA,
#endif                           // 4 no global stores, 8 no weight loads (W1 stages and the slot ring)

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
#undef W_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// every LDS read of this wave has returned, then the workgroup barrier (raw: an LDS-DMA in flight must survive it)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// (diagnostics: an MFMA that can be compiled out, keeping its operands alive)
__device__ __forceinline__ f32x16 MFMA_(half8 x, half8 y, f32x16 c, int, int, int)
{
#if SMAP_CONVB_ABLATE & 2
    c[0] += (float)x[0] + (float)y[1];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
#endif
}

template <int TH>
__global__ __launch_bounds__(256, TH == 4 ? SMAP_CONVB_WGS4 : 2) void bottleneck_kernel(const ConvArgs a, int tiles_x, int tiles_y)
{
    constexpr int P = 64, C = 4 * P, TW = 16, CH = 32, ROWB = 128;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PROWS = ((PH * PW + 31) / 32) * 32;           // patch rows rounded to MFMA blocks: 128 | 192
    constexpr int MB1 = PROWS / 32;                             // M blocks of phase 1: 4 | 6
    constexpr int BM = TH * TW;                                 // output pixels of the tile: 64 | 128
    constexpr int MI = BM / 64;                                 // 32-pixel blocks per wave in phases 2, 3 (2 x 2 waves): 1 | 2
    static_assert(MB1 == 4 || MB1 == 6, "TH = 4 or 8");
    constexpr int KC2 = P / CH;                                 // 32-channel chunks of the 3x3's and the tail's K: 2
    constexpr int Y1_BYTES = KC2 * PROWS * ROWB;                // 32 | 48 KiB
    constexpr int Y2_BYTES = KC2 * BM * ROWB;                   // 16 | 32 KiB: y2 takes over the start of y1's region
    static_assert(Y2_BYTES + C * 4 <= Y1_BYTES && C == 256, "room for the tail-bias table; one bias value per thread");
    constexpr int LDS_BYTES = (TH == 4 ? SMAP_CONVB_LDS4_KB : SMAP_CONVB_LDS_KB) * 1024;
    // phase 1 stages 16 channels at a time (64-byte rows [hi16 | lo16], one MFMA K step per stage): a 12 | 16 KiB stage, so that
    // 5 | 4 of them are in flight behind the one being multiplied -- x comes from HBM, and with 32-channel stages (3 | 2 of them in
    // 80 KiB) a workgroup waited a full memory latency per stage (profiles/r4_v2_*: 172 us per block)
    constexpr int CH1 = 16, ROW1 = 64, KS1 = C / CH1;           // 16 stages
    constexpr int XS = PROWS * ROW1, WS1 = P * ROW1;            // phase-1 stage = x rows (8 | 12 KiB) + W1 rows (4 KiB)
    constexpr int ST1 = XS + WS1;
    constexpr int NS1 = LDS_BYTES / ST1;                        // stages: 6 | 5
    constexpr int LA = XS / 4096, LB1 = WS1 / 4096, LPT1 = LA + LB1;   // LDS-DMA instructions per thread and stage: 2 | 3, + 1
    constexpr int SLOT = P * ROWB;                              // 8 KiB weight slot of phases 2 and 3: 64 rows x one 32-channel chunk
    constexpr int NS = (LDS_BYTES - Y1_BYTES) / SLOT;           // ring slots behind y1: 6 | 4
    constexpr int LS = SLOT / 4096;                             // 2 per thread
    constexpr int NTAP = 9, NCH3 = C / 64;
    constexpr int NS2 = NTAP * KC2, NS3 = NCH3 * KC2, NSLOT = NS2 + NS3;   // 18 (tap, chunk) slots + 8 (tail chunk, k chunk) slots per tile
    static_assert(NS1 >= 2 && NS >= 2 && NS1 * ST1 <= LDS_BYTES && Y1_BYTES + NS * SLOT <= LDS_BYTES, "LDS plan");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];   // ONE array: a second __shared__ object makes hipcc drain vmcnt

    SMAP_TL_BEGIN
    convb_stagger();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SMAP_TRACE
    long long tr_t[8];
#define TRB(i) tr_t[i] = __builtin_amdgcn_s_memtime()
#else
#define TRB(i)
#endif
    TRB(0);
    int logical;                                                // XCD-aware order (conv.hip): neighbouring tiles share an L2
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
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                    // phases 2, 3: pixel half / channel half of the wave

    // ================================================================= phase 1: y1 = relu(W1 x + b1) on the halo patch
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ w1g = reinterpret_cast<const char*>(a.w0);
    // a wave-wide LDS-DMA covers 16 rows x 64 B: lane -> row lane / 4, 16-byte slot lane % 4; slot s of row r holds logical granule
    // s ^ ((r >> 2) & 3) (0, 1 = hi channels 0..7, 8..15 of the stage; 2, 3 = lo): conflict-free ds_read_b128 (conv.hip, BK = 32)
    const int srow = wave * 16 + (lane >> 2);
    const int gl = (lane & 3) ^ ((srow >> 2) & 3);
    unsigned a_off[LA];                                         // patch row -> byte offset of its granule (0 = zero page)
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int prow = i * 64 + srow;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        a_off[i] = 0;
        if (prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const long long e = ((long long)(b * a.H + iy) * a.W + ix) * a.in_stride_c + (gl & 1) * 8 + (gl >> 1) * a.in_lo;
            a_off[i] = (unsigned)(a.in_off + e * 2);
        }
    }
    auto issue1 = [&](int st, int ks) {                         // stage st <- channels 16 ks .. +15 of x and of W1
        char* sX = smem + st * ST1;
        const char* gA = arena + (unsigned)(ks * CH1 * 2);      // invalid rows: zero page + stage offset
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVB_ABLATE & 1) ? 0 : LA); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_off[i]), (lds_void*)(sX + (i * 64 + wave * 16) * ROW1), 16, 0, 0);
        const char* gW = w1g + (long long)ks * WS1 + (unsigned)(wave * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVB_ABLATE & 8) ? 0 : LB1); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gW + i * 4096), (lds_void*)(sX + XS + i * 4096 + wave * 1024), 16, 0, 0);
    };
    // centre pixels of this lane in phases 2 and 3: p = wm*(MI*32) + mi*32 + l31 -> patch row of the pixel itself
    int crow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        crow[mi] = (p / TW + 1) * PW + (p % TW) + 1;
    }
    // the residual of the last epilogue, collected while x passes through LDS: rs[nc][mi][j][plane] = channels
    // nc*64 + wn*32 + 16*j + 8*lhi .. +7 of pixel (mi, l31) -- the layout the register epilogue of phase 3 ends in
    half8 rs[NCH3][MI][2][2];

    constexpr int NB1 = MB1 == 6 ? 3 : 2;                       // 32 x 32 blocks of y1 per wave: (mb = wave, n = 0 | 1) [+ one of M blocks 4, 5]
    // Biases enter through the ACCUMULATORS (acc = b / 2^-s before the first MFMA; the power-of-two scale makes that exact): an
    // ordinary global load in the middle of the LDS-DMA pipeline would make hipcc drain the whole queue (vmcnt(0)) at its use.
    // b1 here (its loads are the oldest of the kernel), b2 during the last K chunk, b3 via a 1 KiB table in LDS.
    const int xmb = 4 + (wave >> 1), xnb = wave & 1;            // the extra block (TH = 8): patch rows 128..191, channel half wave & 1
    float4 b1raw[NB1][4];
#pragma unroll
    for (int j = 0; j < NB1; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) b1raw[j][q] = *reinterpret_cast<const float4*>(a.bias0 + (j < 2 ? j : xnb) * 32 + 8 * q + 4 * lhi);
    const float b3_mine = a.bias2[tid];                         // tail bias: one value per thread, parked until the table can be written
#pragma unroll
    for (int st = 0; st < NS1 - 1; ++st) issue1(st, st);        // the first stages go out behind the bias loads
    f32x16 acc1[NB1];
    {
        const float inv0 = 1.f / a.acc_scale0;
#pragma unroll
        for (int j = 0; j < NB1; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc1[j][4 * q + 0] = b1raw[j][q].x * inv0; acc1[j][4 * q + 1] = b1raw[j][q].y * inv0;
                acc1[j][4 * q + 2] = b1raw[j][q].z * inv0; acc1[j][4 * q + 3] = b1raw[j][q].w * inv0;
            }
    }
    const int fswz = (l31 >> 1) & 7;                            // (row >> 1) & 7 of every 128-byte fragment row = multiple of 32 + l31
    TRB(1);

    const int fs4 = (l31 >> 2) & 3;                             // (row >> 2) & 3 of every phase-1 fragment row = multiple of 32 + l31
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        if (ks + NS1 - 1 <= KS1) wait_vm((NS1 - 2) * LPT1);     // stage ks has landed; younger stages stay in flight
        else wait_vm((KS1 - 1 - ks) * LPT1);
        lds_barrier();
        if (ks + NS1 - 1 < KS1) issue1((ks + NS1 - 1) % NS1, ks + NS1 - 1);     // into the buffer stage ks-1 was read from
        const char* sX = smem + (ks % NS1) * ST1;
        const char* sW = sX + XS;
        if (((ks >> 1) & 1) == wn) {                            // this wave's residual channels: nc*64 + wn*32 + 16*j + 8*lhi .. = stage 4 nc + 2 wn + j
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    rs[ks >> 2][mi][ks & 1][pl] = *reinterpret_cast<const half8*>(
                        sX + crow[mi] * ROW1 + (((lhi + 2 * pl) ^ ((crow[mi] >> 2) & 3)) << 4));
        }
        half8 wf[2][2], xf[2], xe[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int slot = ((lhi + 2 * pl) ^ fs4) << 4;
            wf[pl][0] = *reinterpret_cast<const half8*>(sW + l31 * ROW1 + slot);
            wf[pl][1] = *reinterpret_cast<const half8*>(sW + (32 + l31) * ROW1 + slot);
            xf[pl] = *reinterpret_cast<const half8*>(sX + (wave * 32 + l31) * ROW1 + slot);
            if (MB1 == 6) xe[pl] = *reinterpret_cast<const half8*>(sX + (xmb * 32 + l31) * ROW1 + slot);
        }
        CONVB_PRIO(1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {                        // small cross terms first, then hi*hi (conv3.hip's order)
            acc1[nb] = MFMA_(wf[0][nb], xf[1], acc1[nb], 0, 0, 0);
            acc1[nb] = MFMA_(wf[1][nb], xf[0], acc1[nb], 0, 0, 0);
            acc1[nb] = MFMA_(wf[0][nb], xf[0], acc1[nb], 0, 0, 0);
        }
        if (MB1 == 6) {                                         // the extra block's channel half is wave-uniform: a scalar branch, no copy
            if (xnb) {
                acc1[NB1 - 1] = MFMA_(wf[0][1], xe[1], acc1[NB1 - 1], 0, 0, 0);
                acc1[NB1 - 1] = MFMA_(wf[1][1], xe[0], acc1[NB1 - 1], 0, 0, 0);
                acc1[NB1 - 1] = MFMA_(wf[0][1], xe[0], acc1[NB1 - 1], 0, 0, 0);
            } else {
                acc1[NB1 - 1] = MFMA_(wf[0][0], xe[1], acc1[NB1 - 1], 0, 0, 0);
                acc1[NB1 - 1] = MFMA_(wf[1][0], xe[0], acc1[NB1 - 1], 0, 0, 0);
                acc1[NB1 - 1] = MFMA_(wf[0][0], xe[0], acc1[NB1 - 1], 0, 0, 0);
            }
        }
        CONVB_PRIO(0);
    }
    lds_barrier();                                              // every wave is done with the staging buffers (all DMA has landed)
    TRB(2);
    // phase 2's accumulators start at b2 / scale: the loads go out now, ahead of the first weight slots, and are consumed after
    // y1 has been written (the wait hipcc puts there covers slot 0, which is needed then anyway)
    float4 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = *reinterpret_cast<const float4*>(a.bias + wn * 32 + 8 * q + 4 * lhi);

    // ---- weight slots of phases 2 and 3, 8 KiB each (64 rows of one 32-channel chunk): 18 (tap, chunk) slots, then 8 (tail chunk, k chunk)
    char* ring = smem + Y1_BYTES;
    const char* __restrict__ w2g = reinterpret_cast<const char*>(a.w);
    const char* __restrict__ w3g = reinterpret_cast<const char*>(a.w2);
    const unsigned wlane = (unsigned)(wave * 1024 + lane * 16);
    auto issue_slot = [&](int s) {
        char* dst = ring + (s % NS) * SLOT + wave * 1024;
        // slot s < 18: tap s / 2, chunk s % 2 of the 3x3 (conv3.hip's blocks are ordered [chunk][tap]); then [tail chunk][k chunk]
        const char* g = (s < NS2 ? w2g + (long long)((s % KC2) * NTAP + s / KC2) * SLOT : w3g + (long long)(s - NS2) * SLOT) + wlane;
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVB_ABLATE & 8) ? 0 : LS); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + i * 4096), (lds_void*)(dst + i * 4096), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_slot(s);
    // Ring protocol.  Slot s: [barrier: slot s is published, slot s-1's buffer is free] -> issue slot s + NS - 1 -> multiply ->
    // WAIT for the next slot(s); younger ones stay in flight.  In phase 3 a chunk's two slots are waited for together at the end
    // of the PREVIOUS chunk's multiplications, i.e. BEFORE that chunk's stores: a counted vmcnt also counts stores, and a wait
    // that covers stores issued a moment ago costs a store round trip (csrc/convf.hip pays that once per chunk).
    auto wait_slots = [&](int cur, int upto) {                  // in slot `cur` (its issue done): slots <= upto have landed
        if (upto >= NSLOT) upto = NSLOT - 1;
        const int issued = cur + NS - 1 < NSLOT ? cur + NS - 1 : NSLOT - 1;
        wait_vm((issued > upto ? issued - upto : 0) * LS);
    };
    // ---- accumulators -> y1 [KC2][PROWS][128 B] (rows = patch pixels, conv3.hip's format).  acc[4*q + e] = channel
    //      nb*32 + 8*q + 4*lhi + e of patch row mb*32 + l31; rows outside the image are the 3x3's zero padding.
    char* sY1 = smem;
    auto put_y1 = [&](const f32x16& acc, int nb, int mb) {
        const int prow = mb * 32 + l31;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool live = prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int sw = (prow >> 1) & 7;
        char* row = sY1 + (nb * PROWS + prow) * ROWB + lhi * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[4 * q + e] * a.acc_scale0;        // (bias inside: accumulator start value)
                x = x < 0.f ? 0.f : x;                          // NaN stays NaN (torch's ReLU)
                x = live ? x : 0.f;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<half4*>(row + ((q ^ sw) << 4)) = h;
            *reinterpret_cast<half4*>(row + (((q + 4) ^ sw) << 4)) = l;
        }
    };
    put_y1(acc1[0], 0, wave);
    put_y1(acc1[1], 1, wave);
    if (MB1 == 6) put_y1(acc1[NB1 - 1], xnb, xmb);
    f32x16 acc2[MI];
    {
        const float inv = 1.f / a.acc_scale;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc2[mi][4 * q + 0] = b2v[q].x * inv; acc2[mi][4 * q + 1] = b2v[q].y * inv;
                acc2[mi][4 * q + 2] = b2v[q].z * inv; acc2[mi][4 * q + 3] = b2v[q].w * inv;
            }
    }
    wait_vm((NS - 2) * LS);                                     // slot 0 (only slots 0 .. NS-2 are issued)
    TRB(3);

    // ================================================================= phase 2: the 3x3 on y1
    int prow0[MI];                                              // patch row of tap (0,0) of this lane's pixels
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) prow0[mi] = crow[mi] - PW - 1;
    const int b_row0 = wn * (P / 2) + l31;                      // this wave's 32 output channels of the 3x3

#pragma unroll
    for (int s = 0; s < NS2; ++s) {
        lds_barrier();                                          // slot s landed for every wave; y1 complete (s = 0); slot s-1's buffer is free
        if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
        const char* sB = ring + (s % NS) * SLOT;
        const int tap = s / KC2, cc = s % KC2;
        const int shift = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int kk = 0; kk < CH / 16; ++kk) {
            const int g = kk * 2 + lhi;
            half8 af[2][MI], bf[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int prow = prow0[mi] + shift;
                    af[pl][mi] = *reinterpret_cast<const half8*>(sY1 + (cc * PROWS + prow) * ROWB + (((g + 4 * pl) ^ ((prow >> 1) & 7)) << 4));
                }
                bf[pl] = *reinterpret_cast<const half8*>(sB + b_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
            }
            CONVB_PRIO(1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc2[mi] = MFMA_(bf[0], af[1][mi], acc2[mi], 0, 0, 0);
                acc2[mi] = MFMA_(bf[1], af[0][mi], acc2[mi], 0, 0, 0);
                acc2[mi] = MFMA_(bf[0], af[0][mi], acc2[mi], 0, 0, 0);
            }
            CONVB_PRIO(0);
        }
        wait_slots(s, s + 1 < NS2 ? s + 1 : s + 2);            // the last tap also waits for both slots of the first tail chunk
    }
    lds_barrier();                                              // every wave is done with y1: y2 may overwrite it
    TRB(4);

    // ---- accumulators -> y2 [KC2][BM][128 B] (rows = tile pixels).  acc2[mi][4*q + e] = channel wn*32 + 8*q + 4*lhi + e
    char* sY2 = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        const int sw = (p >> 1) & 7;
        char* row = sY2 + (wn * BM + p) * ROWB + lhi * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc2[mi][4 * q + e] * a.acc_scale;     // (bias inside)
                x = x < 0.f ? 0.f : x;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<half4*>(row + ((q ^ sw) << 4)) = h;
            *reinterpret_cast<half4*>(row + (((q + 4) ^ sw) << 4)) = l;
        }
    }

    float* sB3 = reinterpret_cast<float*>(smem + Y2_BYTES);    // [C] tail bias / scale: the part of y1's region y2 leaves free
    sB3[tid] = b3_mine * (1.f / a.tail_acc_scale);

    // ================================================================= phase 3: the tail 1x1 + residual + ReLU (+ skip adds)
    unsigned m_dense[MI], m_out[MI];
    bool m_ok[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        m_ok[mi] = oy < a.Ho && ox < a.Wo;
        const unsigned m = m_ok[mi] ? (unsigned)((b * a.Ho + oy) * a.Wo + ox) : 0u;
        m_dense[mi] = m * (unsigned)(2 * a.tail_cout8);
        m_out[mi] = m * (unsigned)a.out_stride_c + (unsigned)a.out_c_off;
    }
    const int p_row0 = wm * (MI * 32) + l31;                    // + mi*32: pixel rows of y2
    const int c_row0 = wn * 32 + l31;                           // channel rows of a tail chunk
    _Float16* __restrict__ outp = reinterpret_cast<_Float16*>(a.out);

#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc) {
        f32x16 acc3[MI];                                        // start value b3 / scale: rows = channels nc*64 + wn*32 + 8*q + 4*lhi + e
#pragma unroll
        for (int kc = 0; kc < KC2; ++kc) {
            const int s = NS2 + nc * KC2 + kc;
            lds_barrier();                                      // slot s landed for every wave; y2 + bias table complete (first slot); slot s-1's buffer is free
            if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
            const char* sW = ring + (s % NS) * SLOT;
            if (kc == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(sB3 + nc * 64 + wn * 32 + 8 * q + 4 * lhi);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc3[mi][4 * q + 0] = b4.x; acc3[mi][4 * q + 1] = b4.y; acc3[mi][4 * q + 2] = b4.z; acc3[mi][4 * q + 3] = b4.w;
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < CH / 16; ++kk) {
                const int g = kk * 2 + lhi;
                half8 pf[2][MI], wf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        pf[pl][mi] = *reinterpret_cast<const half8*>(sY2 + (kc * BM + p_row0 + mi * 32) * ROWB + (((g + 4 * pl) ^ fswz) << 4));
                    wf[pl] = *reinterpret_cast<const half8*>(sW + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
                }
                CONVB_PRIO(1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc3[mi] = MFMA_(wf[0], pf[1][mi], acc3[mi], 0, 0, 0);
                    acc3[mi] = MFMA_(wf[1], pf[0][mi], acc3[mi], 0, 0, 0);
                    acc3[mi] = MFMA_(wf[0], pf[0][mi], acc3[mi], 0, 0, 0);
                }
                CONVB_PRIO(0);
            }
            if (kc == KC2 - 1) wait_slots(s, s + 2);            // both slots of the next chunk, BEFORE this chunk's stores
        }
        // ---- register epilogue (convp.hip): half-wave swap -> acc3[mi][8*j .. 8*j+7] = channels n_lane + 16*j .. +7 of the pixel
        const int n_lane = nc * 64 + wn * 32 + 8 * lhi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xf = acc3[mi][8 * j + e], yf = acc3[mi][8 * j + 4 + e];   // (bit_cast of a vector ELEMENT lvalue reads element 0)
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    acc3[mi][8 * j + e] = a.tail_acc_scale * __uint_as_float(s0);       // (bias inside)
                    acc3[mi][8 * j + 4 + e] = a.tail_acc_scale * __uint_as_float(s1);
                }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)                          // + x, from the registers filled in phase 1
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc3[mi][8 * j + e] += (float)rs[nc][mi][j][0][e] + (float)rs[nc][mi][j][1][e];
        if (a.relu) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mi][r] = acc3[mi][r] < 0.f ? 0.f : acc3[mi][r];
        }
        auto add_tensor = [&](const _Float16* __restrict__ tsr) {       // post-ReLU skip adds of the last block of a layer
            half8 h[MI][2][2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        h[mi][j][pl] = *reinterpret_cast<const half8*>(tsr + m_dense[mi] + (unsigned)(n_lane + 16 * j) + pl * a.tail_cout8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc3[mi][8 * j + e] += (float)h[mi][j][0][e] + (float)h[mi][j][1][e];
        };
        if (a.add1) add_tensor(a.add1);
        if (a.add2) add_tensor(a.add2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!m_ok[mi]) continue;
                _Float16* op = outp + (m_out[mi] + (unsigned)(n_lane + 16 * j));
                half8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = (_Float16)acc3[mi][8 * j + e];
                    l[e] = (_Float16)(acc3[mi][8 * j + e] - (float)h[e]);
                }
                if (SMAP_CONVB_ABLATE & 4) { if (h[0] == (_Float16)123.25f && l[1] == (_Float16)77.5f) *reinterpret_cast<half8*>(op) = h; continue; }   // keep the values live
                *reinterpret_cast<half8*>(op) = h;
                *reinterpret_cast<half8*>(op + a.out_lo) = l;
            }
    }
#ifdef SMAP_TRACE
    TRB(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TRB(6);
    if (a.dbg && tid == 0) {                                    // stamps of wave 0: start, set-up, phase 1, y1 written, phase 2, last store issued, stores retired
        long long* d = a.dbg + (long long)blockIdx.x * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        d[0] = tr_t[0]; d[1] = tr_t[1]; d[2] = tr_t[2]; d[3] = tr_t[3]; d[4] = tr_t[4]; d[5] = tr_t[5]; d[6] = tr_t[6]; d[7] = hwid;
    }
#endif
    SMAP_TL_END(a)
}

template <int TH>
__global__ __launch_bounds__(256, TH == 4 ? SMAP_CONVB_WGS4 : 2) void bottleneck_first_kernel(const ConvArgs a, int tiles_x, int tiles_y)
{
    constexpr int P = 64, C = 4 * P, TW = 16, CH = 32, ROWB = 128;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PROWS = ((PH * PW + 31) / 32) * 32;           // patch rows rounded to MFMA blocks: 128 | 192
    constexpr int MB1 = PROWS / 32;                             // M blocks of phase 1: 4 | 6
    constexpr int BM = TH * TW;                                 // output pixels of the tile: 64 | 128
    constexpr int MI = BM / 64;                                 // 32-pixel blocks per wave in phases 2, 3 (2 x 2 waves): 1 | 2
    static_assert(MB1 == 4 || MB1 == 6, "TH = 4 or 8");
    constexpr int KC2 = P / CH;                                 // 32-channel chunks of the 3x3's and the tail's K: 2
    constexpr int Y1_BYTES = KC2 * PROWS * ROWB;                // 32 | 48 KiB
    constexpr int Y2_BYTES = KC2 * BM * ROWB;                   // 16 | 32 KiB: y2 takes over the start of y1's region
    static_assert(Y2_BYTES + C * 4 + P * 4 <= Y1_BYTES && C == 256, "room for the bias tables; one tail-bias value per thread");
    constexpr int LDS_BYTES = (TH == 4 ? SMAP_CONVB_LDS4_KB : SMAP_CONVB_LDS_KB) * 1024;
    constexpr int SLOT = P * ROWB;                              // 8 KiB weight slot of phases 2 and 3: 64 rows x one 32-channel chunk
    constexpr int NS = (LDS_BYTES - Y1_BYTES) / SLOT;           // ring slots behind y1: 6 | 4
    constexpr int LS = SLOT / 4096;                             // 2 per thread
    constexpr int NTAP = 9, NCH3 = C / 64;
    constexpr int NS2 = NTAP * KC2, NS3 = NCH3 * KC2;          // 18 (tap, chunk) slots, 8 (tail chunk, k chunk) slots
    constexpr int S1 = KC2, SD = NCH3 * KC2, SB = S1 + SD;      // in front of them: 2 chunks of W1, 8 (chunk, k chunk) slots of the shortcut conv
    constexpr int NSLOT = SB + NS2 + NS3;                       // 36 weight slots per tile
    constexpr int LX = PROWS / 32;                              // LDS-DMA instructions per thread and 32-channel chunk of the x patch
    static_assert(NS >= 2 && Y1_BYTES + NS * SLOT <= LDS_BYTES, "LDS plan");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];   // ONE array: a second __shared__ object makes hipcc drain vmcnt

    SMAP_TL_BEGIN
    convb_stagger();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SMAP_TRACE
    long long tr_t[8];
#define TRB(i) tr_t[i] = __builtin_amdgcn_s_memtime()
#else
#define TRB(i)
#endif
    TRB(0);
    int logical;                                                // XCD-aware order (conv.hip): neighbouring tiles share an L2
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
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                    // phases 2, 3: pixel half / channel half of the wave

    // ================================================================= the x patch: C_in = 64 channels = two 32-channel chunks, loaded ONCE
    // into the region y1 will take over: [KC2][PROWS][128 B], conv3.hip's patch format (slot s of row r = granule s ^ ((r >> 1) & 7))
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ w1g = reinterpret_cast<const char*>(a.w0);
    const char* __restrict__ wdg = reinterpret_cast<const char*>(a.wd);
    char* sX = smem;
    {
        const int lrow = lane >> 3, lslot = lane & 7;
        const int srow = wave * 8 + lrow;
        const int gl = lslot ^ ((srow >> 1) & 7);               // logical granule this lane fetches: 0..3 hi, 4..7 lo
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int prow = i * 32 + srow;
            const int py = prow / PW, px = prow - py * PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            unsigned off = 0;                                   // 0 = zero page
            if (prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const long long e = ((long long)(b * a.H + iy) * a.W + ix) * a.in_stride_c + (gl & 3) * 8 + (gl >> 2) * a.in_lo;
                off = (unsigned)(a.in_off + e * 2);
            }
#pragma unroll
            for (int cc = 0; cc < KC2; ++cc)
                __builtin_amdgcn_global_load_lds((gbl_void*)(arena + (unsigned)(cc * CH * 2) + off),
                                                 (lds_void*)(sX + (cc * PROWS + i * 32 + wave * 8) * ROWB), 16, 0, 0);
        }
    }
    // centre pixels of this lane in phases 1b, 2 and 3: p = wm*(MI*32) + mi*32 + l31 -> patch row of the pixel itself
    int crow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        crow[mi] = (p / TW + 1) * PW + (p % TW) + 1;
    }
    constexpr int NB1 = MB1 == 6 ? 3 : 2;                       // 32 x 32 blocks of y1 per wave: (mb = wave, n = 0 | 1) [+ one of M blocks 4, 5]
    const int xmb = 4 + (wave >> 1), xnb = wave & 1;
    // biases: b1 through the accumulators (its loads are the oldest of the kernel); b2 and the tail bias (b3 + the shortcut's)
    // one value per thread, parked in a register until a table in LDS can take them (after phase 2)
    float4 b1raw[NB1][4];
#pragma unroll
    for (int j = 0; j < NB1; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) b1raw[j][q] = *reinterpret_cast<const float4*>(a.bias0 + (j < 2 ? j : xnb) * 32 + 8 * q + 4 * lhi);
    const float b3_mine = a.bias2[tid];
    const float b2_mine = a.bias[tid & (P - 1)];
    // ---- ONE ring of 8 KiB weight slots (64 rows of one 32-channel chunk) through the whole kernel, started before anything else
    char* ring = smem + Y1_BYTES;
    const char* __restrict__ w2g = reinterpret_cast<const char*>(a.w);
    const char* __restrict__ w3g = reinterpret_cast<const char*>(a.w2);
    const unsigned wlane = (unsigned)(wave * 1024 + lane * 16);
    auto issue_slot = [&](int s) {
        char* dst = ring + (s % NS) * SLOT + wave * 1024;
        // slots: W1 chunk s | shortcut [n chunk][k chunk] | tap t / 2, chunk t % 2 of the 3x3 (blocks ordered [chunk][tap]) | tail [n chunk][k chunk]
        const int t2 = s - SB;
        const char* g = (s < S1 ? w1g + (long long)s * SLOT : s < SB ? wdg + (long long)(s - S1) * SLOT :
                         t2 < NS2 ? w2g + (long long)((t2 % KC2) * NTAP + t2 / KC2) * SLOT : w3g + (long long)(t2 - NS2) * SLOT) + wlane;
#pragma unroll
        for (int i = 0; i < LS; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + i * 4096), (lds_void*)(dst + i * 4096), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_slot(s);
    // Ring protocol.  Slot s: [barrier: slot s is published, slot s-1's buffer is free] -> issue slot s + NS - 1 -> multiply ->
    // WAIT for the next slot(s); younger ones stay in flight.  In phase 3 a chunk's two slots are waited for together at the end
    // of the PREVIOUS chunk's multiplications, i.e. BEFORE that chunk's stores: a counted vmcnt also counts stores, and a wait
    // that covers stores issued a moment ago costs a store round trip (csrc/convf.hip pays that once per chunk).
    auto wait_slots = [&](int cur, int upto) {                  // in slot `cur` (its issue done): slots <= upto have landed
        if (upto >= NSLOT) upto = NSLOT - 1;
        const int issued = cur + NS - 1 < NSLOT ? cur + NS - 1 : NSLOT - 1;
        wait_vm((issued > upto ? issued - upto : 0) * LS);
    };
    // ---- accumulators -> y1 [KC2][PROWS][128 B] (rows = patch pixels, conv3.hip's format).  acc[4*q + e] = channel
    //      nb*32 + 8*q + 4*lhi + e of patch row mb*32 + l31; rows outside the image are the 3x3's zero padding.
    char* sY1 = smem;
    auto put_y1 = [&](const f32x16& acc, int nb, int mb) {
        const int prow = mb * 32 + l31;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool live = prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int sw = (prow >> 1) & 7;
        char* row = sY1 + (nb * PROWS + prow) * ROWB + lhi * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[4 * q + e] * a.acc_scale0;        // (bias inside: accumulator start value)
                x = x < 0.f ? 0.f : x;                          // NaN stays NaN (torch's ReLU)
                x = live ? x : 0.f;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<half4*>(row + ((q ^ sw) << 4)) = h;
            *reinterpret_cast<half4*>(row + (((q + 4) ^ sw) << 4)) = l;
        }
    };
    const int fswz = (l31 >> 1) & 7;                            // (row >> 1) & 7 of every fragment row = multiple of 32 + l31
    f32x16 acc1[NB1];
    {
        const float inv0 = 1.f / a.acc_scale0;
#pragma unroll
        for (int j = 0; j < NB1; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc1[j][4 * q + 0] = b1raw[j][q].x * inv0; acc1[j][4 * q + 1] = b1raw[j][q].y * inv0;
                acc1[j][4 * q + 2] = b1raw[j][q].z * inv0; acc1[j][4 * q + 3] = b1raw[j][q].w * inv0;
            }
    }
    wait_vm((NS - 2) * LS);                                     // the x patch and slot 0 (older than slots 1 .. NS-2)
    TRB(1);
    // ================================================================= phase 1a: y1 = relu(W1 x + b1) on the halo patch (K = 64: two slots)
#pragma unroll
    for (int s = 0; s < S1; ++s) {
        lds_barrier();
        if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
        const char* sW = ring + (s % NS) * SLOT;
#pragma unroll
        for (int kk = 0; kk < CH / 16; ++kk) {
            const int slot0 = ((kk * 2 + lhi) ^ fswz) << 4, slot1 = ((kk * 2 + lhi + 4) ^ fswz) << 4;
            half8 wf[2][2], xf[2], xe[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int slot = pl ? slot1 : slot0;
                wf[pl][0] = *reinterpret_cast<const half8*>(sW + l31 * ROWB + slot);
                wf[pl][1] = *reinterpret_cast<const half8*>(sW + (32 + l31) * ROWB + slot);
                xf[pl] = *reinterpret_cast<const half8*>(sX + (s * PROWS + wave * 32 + l31) * ROWB + slot);
                if (MB1 == 6) xe[pl] = *reinterpret_cast<const half8*>(sX + (s * PROWS + xmb * 32 + l31) * ROWB + slot);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nb], xf[1], acc1[nb], 0, 0, 0);
                acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][nb], xf[0], acc1[nb], 0, 0, 0);
                acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][nb], xf[0], acc1[nb], 0, 0, 0);
            }
            if (MB1 == 6) {
                if (xnb) {
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][1], xe[1], acc1[NB1 - 1], 0, 0, 0);
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][1], xe[0], acc1[NB1 - 1], 0, 0, 0);
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][1], xe[0], acc1[NB1 - 1], 0, 0, 0);
                } else {
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][0], xe[1], acc1[NB1 - 1], 0, 0, 0);
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][0], xe[0], acc1[NB1 - 1], 0, 0, 0);
                    acc1[NB1 - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][0], xe[0], acc1[NB1 - 1], 0, 0, 0);
                }
            }
        }
        wait_slots(s, s + 1);
    }
    // ================================================================= phase 1b: the shortcut conv Wd x on the tile's own pixels (1x1, 64 -> 256,
    // no ReLU): kept in the accumulators until the last epilogue -- where an identity block keeps x itself (its residual)
    const int c_row0 = wn * 32 + l31;                           // channel rows of a 64-channel chunk (shortcut and tail)
    f32x16 accd[NCH3][MI];
#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) accd[nc][mi][r] = 0.f;
#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc)
#pragma unroll
        for (int kc = 0; kc < KC2; ++kc) {
            const int s = S1 + nc * KC2 + kc;
            lds_barrier();
            if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
            const char* sW = ring + (s % NS) * SLOT;
#pragma unroll
            for (int kk = 0; kk < CH / 16; ++kk) {
                const int g = kk * 2 + lhi;
                half8 pf[2][MI], wf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        pf[pl][mi] = *reinterpret_cast<const half8*>(sX + (kc * PROWS + crow[mi]) * ROWB + (((g + 4 * pl) ^ ((crow[mi] >> 1) & 7)) << 4));
                    wf[pl] = *reinterpret_cast<const half8*>(sW + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    accd[nc][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], pf[1][mi], accd[nc][mi], 0, 0, 0);
                    accd[nc][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1], pf[0][mi], accd[nc][mi], 0, 0, 0);
                    accd[nc][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], pf[0][mi], accd[nc][mi], 0, 0, 0);
                }
            }
            wait_slots(s, s + 1);
        }
    lds_barrier();                                              // every wave is done with the x patch: y1 takes its place
    TRB(2);
    put_y1(acc1[0], 0, wave);
    put_y1(acc1[1], 1, wave);
    if (MB1 == 6) put_y1(acc1[NB1 - 1], xnb, xmb);
    f32x16 acc2[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[mi][r] = 0.f;
    TRB(3);

    // ================================================================= phase 2: the 3x3 on y1
    int prow0[MI];                                              // patch row of tap (0,0) of this lane's pixels
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) prow0[mi] = crow[mi] - PW - 1;
    const int b_row0 = wn * (P / 2) + l31;                      // this wave's 32 output channels of the 3x3

#pragma unroll
    for (int t2 = 0; t2 < NS2; ++t2) {
        const int s = SB + t2;
        lds_barrier();                                          // slot s landed for every wave; y1 complete (first tap); slot s-1's buffer is free
        if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
        const char* sB = ring + (s % NS) * SLOT;
        const int tap = t2 / KC2, cc = t2 % KC2;
        const int shift = (tap / 3) * PW + (tap % 3);
#pragma unroll
        for (int kk = 0; kk < CH / 16; ++kk) {
            const int g = kk * 2 + lhi;
            half8 af[2][MI], bf[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int prow = prow0[mi] + shift;
                    af[pl][mi] = *reinterpret_cast<const half8*>(sY1 + (cc * PROWS + prow) * ROWB + (((g + 4 * pl) ^ ((prow >> 1) & 7)) << 4));
                }
                bf[pl] = *reinterpret_cast<const half8*>(sB + b_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc2[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[1][mi], acc2[mi], 0, 0, 0);
                acc2[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[1], af[0][mi], acc2[mi], 0, 0, 0);
                acc2[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[0][mi], acc2[mi], 0, 0, 0);
            }
        }
        wait_slots(s, t2 + 1 < NS2 ? s + 1 : s + 2);           // the last tap also waits for both slots of the first tail chunk
    }
    lds_barrier();                                              // every wave is done with y1: y2 may overwrite it
    TRB(4);

    // ---- bias tables in the part of y1's region that y2 leaves free: [C] (b3 + shortcut bias) / tail scale, [P] b2
    float* sB3 = reinterpret_cast<float*>(smem + Y2_BYTES);
    float* sB2 = sB3 + C;
    sB3[tid] = b3_mine * (1.f / a.tail_acc_scale);
    if (tid < P) sB2[tid] = b2_mine;
    lds_barrier();
    // ---- accumulators -> y2 [KC2][BM][128 B] (rows = tile pixels).  acc2[mi][4*q + e] = channel wn*32 + 8*q + 4*lhi + e
    char* sY2 = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        const int sw = (p >> 1) & 7;
        char* row = sY2 + (wn * BM + p) * ROWB + lhi * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(sB2 + wn * 32 + 8 * q + 4 * lhi);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc2[mi][4 * q + e] * a.acc_scale + bb[e];
                x = x < 0.f ? 0.f : x;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<half4*>(row + ((q ^ sw) << 4)) = h;
            *reinterpret_cast<half4*>(row + (((q + 4) ^ sw) << 4)) = l;
        }
    }

    // ================================================================= phase 3: the tail 1x1 + residual + ReLU (+ skip adds)
    unsigned m_dense[MI], m_out[MI];
    bool m_ok[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * (MI * 32) + mi * 32 + l31;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        m_ok[mi] = oy < a.Ho && ox < a.Wo;
        const unsigned m = m_ok[mi] ? (unsigned)((b * a.Ho + oy) * a.Wo + ox) : 0u;
        m_dense[mi] = m * (unsigned)(2 * a.tail_cout8);
        m_out[mi] = m * (unsigned)a.out_stride_c + (unsigned)a.out_c_off;
    }
    const int p_row0 = wm * (MI * 32) + l31;                    // + mi*32: pixel rows of y2
    _Float16* __restrict__ outp = reinterpret_cast<_Float16*>(a.out);

#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc) {
        f32x16 acc3[MI];                                        // start value b3 / scale: rows = channels nc*64 + wn*32 + 8*q + 4*lhi + e
#pragma unroll
        for (int kc = 0; kc < KC2; ++kc) {
            const int s = SB + NS2 + nc * KC2 + kc;
            lds_barrier();                                      // slot s landed for every wave; y2 + bias table complete (first slot); slot s-1's buffer is free
            if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
            const char* sW = ring + (s % NS) * SLOT;
            if (kc == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(sB3 + nc * 64 + wn * 32 + 8 * q + 4 * lhi);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc3[mi][4 * q + 0] = b4.x; acc3[mi][4 * q + 1] = b4.y; acc3[mi][4 * q + 2] = b4.z; acc3[mi][4 * q + 3] = b4.w;
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < CH / 16; ++kk) {
                const int g = kk * 2 + lhi;
                half8 pf[2][MI], wf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        pf[pl][mi] = *reinterpret_cast<const half8*>(sY2 + (kc * BM + p_row0 + mi * 32) * ROWB + (((g + 4 * pl) ^ fswz) << 4));
                    wf[pl] = *reinterpret_cast<const half8*>(sW + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc3[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], pf[1][mi], acc3[mi], 0, 0, 0);
                    acc3[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1], pf[0][mi], acc3[mi], 0, 0, 0);
                    acc3[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], pf[0][mi], acc3[mi], 0, 0, 0);
                }
            }
            if (kc == KC2 - 1) wait_slots(s, s + 2);            // both slots of the next chunk, BEFORE this chunk's stores
        }
        {                                                       // + the shortcut conv's accumulators (same lane layout), in units of the tail's scale
            const float rsd = a.acc_scale_d / a.tail_acc_scale;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mi][r] += rsd * accd[nc][mi][r];
        }
        // ---- register epilogue (convp.hip): half-wave swap -> acc3[mi][8*j .. 8*j+7] = channels n_lane + 16*j .. +7 of the pixel
        const int n_lane = nc * 64 + wn * 32 + 8 * lhi;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xf = acc3[mi][8 * j + e], yf = acc3[mi][8 * j + 4 + e];   // (bit_cast of a vector ELEMENT lvalue reads element 0)
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                    const unsigned s0 = sw[0], s1 = sw[1];
                    acc3[mi][8 * j + e] = a.tail_acc_scale * __uint_as_float(s0);       // (bias inside)
                    acc3[mi][8 * j + 4 + e] = a.tail_acc_scale * __uint_as_float(s1);
                }
        if (a.relu) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mi][r] = acc3[mi][r] < 0.f ? 0.f : acc3[mi][r];
        }
        auto add_tensor = [&](const _Float16* __restrict__ tsr) {       // post-ReLU skip adds of the last block of a layer
            half8 h[MI][2][2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        h[mi][j][pl] = *reinterpret_cast<const half8*>(tsr + m_dense[mi] + (unsigned)(n_lane + 16 * j) + pl * a.tail_cout8);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc3[mi][8 * j + e] += (float)h[mi][j][0][e] + (float)h[mi][j][1][e];
        };
        if (a.add1) add_tensor(a.add1);
        if (a.add2) add_tensor(a.add2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!m_ok[mi]) continue;
                _Float16* op = outp + (m_out[mi] + (unsigned)(n_lane + 16 * j));
                half8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = (_Float16)acc3[mi][8 * j + e];
                    l[e] = (_Float16)(acc3[mi][8 * j + e] - (float)h[e]);
                }
                *reinterpret_cast<half8*>(op) = h;
                *reinterpret_cast<half8*>(op + a.out_lo) = l;
            }
    }
#ifdef SMAP_TRACE
    TRB(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TRB(6);
    if (a.dbg && tid == 0) {                                    // stamps of wave 0: start, set-up, phase 1, y1 written, phase 2, last store issued, stores retired
        long long* d = a.dbg + (long long)blockIdx.x * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        d[0] = tr_t[0]; d[1] = tr_t[1]; d[2] = tr_t[2]; d[3] = tr_t[3]; d[4] = tr_t[4]; d[5] = tr_t[5]; d[6] = tr_t[6]; d[7] = hwid;
    }
#endif
    SMAP_TL_END(a)
}

template <int TH, bool FIRST>
hipError_t launchb(const ConvArgs& a, hipStream_t st)
{
    const int B = a.M / (a.Ho * a.Wo);
    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + TH - 1) / TH;
    if (FIRST) hipLaunchKernelGGL((bottleneck_first_kernel<TH>), dim3(tiles_x * tiles_y * B), dim3(256), 0, st, a, tiles_x, tiles_y);
    else hipLaunchKernelGGL((bottleneck_kernel<TH>), dim3(tiles_x * tiles_y * B), dim3(256), 0, st, a, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace

// tile ids 90..99: the fused identity Bottleneck (P = 64 planes; *bm = output pixels per workgroup, *bn = P, *bn2 = tail chunk)
int smap_convb_tile_dims(int tile, int* bm, int* bn, int* bn2)
{
    switch (tile) {
        case 90: *bm = 64; *bn = 64; *bn2 = 64; return 0;       // 4 x 16 pixel tiles
        case 91: *bm = 128; *bn = 64; *bn2 = 64; return 0;      // 8 x 16
        case 92: *bm = 64; *bn = 64; *bn2 = 64; return 0;       // FIRST block of a layer (64 input channels, 1x1 shortcut conv): 4 x 16
        case 93: *bm = 128; *bn = 64; *bn2 = 64; return 0;      //   8 x 16
        case 94: *bm = 128; *bn = 128; *bn2 = 128; return 0;    // csrc/convc.hip: identity blocks of 128 planes / 512 channels, 8 x 16
        default: return -1;
    }
}

// tools only (not part of include/smap_hip.h; exported from diagnostics builds with -DSMAP_DEBUG_EXPORTS, tools/build_ablate.py): resident
// workgroups per CU the runtime reports for a tile id's kernel
extern "C" __attribute__((visibility("default"))) int smap_debug_convb_occupancy(int tile);
#ifdef SMAP_DEBUG_EXPORTS
extern "C" int smap_debug_convb_occupancy(int tile)
{
    int n = -1;
    hipError_t e = hipErrorInvalidValue;
    switch (tile) {
        case 90: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bottleneck_kernel<4>, 256, 0); break;
        case 91: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bottleneck_kernel<8>, 256, 0); break;
        case 92: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bottleneck_first_kernel<4>, 256, 0); break;
        case 93: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bottleneck_first_kernel<8>, 256, 0); break;
        default: break;
    }
    return e == hipSuccess ? n : -1000 - (int)e;
}
#endif

hipError_t smap_launch_convb(const ConvArgs& a, int tile, hipStream_t st)
{
    if (tile == 94) return smap_launch_convc(a, st);
    if (!a.x3 || a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.up || a.out_fp32 || !a.w0 || !a.w2 || a.Cin != 64 ||
        a.tail_cout8 != 256 || a.H != a.Ho || a.W != a.Wo)
        return hipErrorInvalidValue;
    const bool first = tile == 92 || tile == 93;
    if (first ? (a.head_cin != 64 || !a.wd || a.res || a.add1 || a.add2) : (a.head_cin != 256 || a.wd)) return hipErrorInvalidValue;
    switch (tile) {
        case 90: return launchb<4, false>(a, st);
        case 91: return launchb<8, false>(a, st);
        case 92: return launchb<4, true>(a, st);
        case 93: return launchb<8, true>(a, st);
        default: return hipErrorInvalidValue;
    }
}
