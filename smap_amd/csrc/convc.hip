// convc.hip -- a whole stride-1 identity Bottleneck of 128 planes / 512 channels (layer2, model/smap.py:48-77) in ONE launch, split
// precision: convb.hip's scheme re-planned for twice the width.
//     y1 = relu(W1 x + b1)   1x1, 512 -> 128       y2 = relu(W2 * y1 + b2)   3x3 pad 1, 128 -> 128       out = relu(W3 y2 + b3 + x) (+ skip adds)
// As three launches the block moves 8192 bytes per pixel through the fabric; here x is read once (a quarter of it twice, see below) and out
// written once: 4608.
//
// One workgroup of EIGHT waves per CU (y1 on the halo patch alone is 96 KiB), 160 KiB of LDS, an 8 x 16 tile of output pixels:
//   phase 1  c1 on the 10 x 18 halo patch (192 GEMM rows): K = 512 in 32-channel stages, each staged as conv3.hip rows
//            [192][hi32 | lo32] (x, 24 KiB) + [128][hi32 | lo32] (W1, 16 KiB) by LDS-DMA in a ring of 4 stages over the whole LDS.
//            A wave owns 3 of the 24 32 x 32 blocks of y1: channel block wave & 3, patch-row blocks 3 (wave >> 2) + {0, 1, 2}.
//            While x streams past, every wave copies the centre pixels' values its last epilogue will add into REGISTERS: three of
//            the four 128-channel chunks (96 registers per lane; the fourth would need scratch next to the 48 accumulators of this
//            phase, so the last chunk's residual is loaded again while that chunk is multiplied).
//   phase 2  the 3x3 as nine shifted views of y1 ([4 chunks][192 rows][128 B]), 16 KiB weight slots (128 rows of one 32-channel
//            chunk of one tap) in a ring of 4 behind y1; 2 x 4 waves: pixel half wave >> 2, channel block wave & 3; y2 over y1.
//   phase 3  the tail 1x1 in four chunks of 128 output channels (four 16 KiB k-chunk slots each, same ring), register epilogue:
//            permlane32_swap -> 8 consecutive channels per lane, + residual registers, ReLU, skip adds, 16-byte stores of both planes.
// MFMA operand order is weights FIRST everywhere (D rows = channels, columns = pixels), as in convb.hip.
//
// Weights (smap_amd/engine.py::Graph.conv_block, all three in pack_halo_rows' format: 128-byte rows = [hi32 | lo32] of one 32-channel
// chunk, slot s of row r = logical granule s ^ ((r >> 1) & 7)):
//   W1  [16 k chunks][128 rows]      W2  [4 chunks][9 taps][128 rows]      W3  [4 n chunks][4 k chunks][128 rows]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Residual chunks (of four) whose values are copied into registers while x streams through LDS in phase 1; the others are read again from
// L2 / Infinity Cache while their chunk of the tail is multiplied.  Round 6: 2 (64 registers) instead of 3 (96) -- the registers pay for a
// SECOND fragment set in every phase (the software pipeline below).
#ifndef SMAP_CONVC_NRS
#define SMAP_CONVC_NRS 2
#endif
#ifndef SMAP_CONVC_PIPE
#define SMAP_CONVC_PIPE 1          // 0 = round 5's loops (fragments single-buffered, read -> wait -> multiply inside every K step)
#endif

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
// all LDS reads of this wave have landed (the builtin: hipcc's wait-count pass sees it and adds no lgkmcnt(0) of its own further down)
__device__ __forceinline__ void lds_reads_landed() { __builtin_amdgcn_s_waitcnt(0xC07F); }

__device__ __forceinline__ void lds_barrier()
{
#if SMAP_CONVC_PIPE
    // the BUILTIN, not inline asm: hipcc's wait-count pass then knows that the fragment set read before the barrier has landed and puts no
    // lgkmcnt(0) in front of the MFMAs that use it behind the barrier (gfx9 encoding: vmcnt = 63 and expcnt = 7 "don't wait", lgkmcnt = 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifndef SMAP_CONVC_ABLATE
#define SMAP_CONVC_ABLATE 0        // diagnostics builds only (tools/build_ablate.py --one convc.hip SMAP_CONVC_ABLATE=N): 1 no x requests, 2 no MFMA,
#endif                             // 4 no global stores, 8 no weight requests
#if SMAP_CONVC_ABLATE & 2
__device__ __forceinline__ f32x16 MFMA_(half8 x, half8 y, f32x16 c) { c[0] += (float)x[0] + (float)y[1]; return c; }
#else
#define MFMA_(x, y, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0)
#endif
// (Measured and removed, EXPERIMENTS R4.1b: conv3.hip's staggered schedule for phases 2 and 3 -- the two pixel halves half a slot apart,
//  bit-identical, 138 vs 137-142 us; LDS-DMA requests issued from inside the MFMA bursts -- 144-152 vs 145-146 us.)

__global__ __launch_bounds__(512, 2) void bottleneck128_kernel(const ConvArgs a, int tiles_x, int tiles_y)
{
    constexpr int P = 128, C = 512, TH = 8, TW = 16, CH = 32, ROWB = 128, NW = 8;
    constexpr int PW = TW + 2, PH = TH + 2;
    constexpr int PROWS = 192;                                  // 180 patch rows rounded to MFMA blocks
    constexpr int BM = TH * TW;                                 // 128 output pixels
    constexpr int MI = 2;                                       // 32-pixel blocks per wave in phases 2, 3 (2 x 4 waves)
    constexpr int KC = P / CH;                                  // 4 chunks of the 3x3's and the tail's K
    constexpr int Y1_BYTES = KC * PROWS * ROWB;                 // 96 KiB
    constexpr int Y2_BYTES = KC * BM * ROWB;                    // 64 KiB: y2 takes over the start of y1's region
    constexpr int LDS_BYTES = 160 * 1024;
    static_assert(PH * PW <= PROWS && Y2_BYTES + C * 4 <= Y1_BYTES && C == NW * 64, "tail-bias table behind y2; one bias value per thread");
    constexpr int XS = PROWS * ROWB, WS1 = P * ROWB, ST1 = XS + WS1;     // phase-1 stage: 24 KiB of x rows + 16 KiB of W1 rows
    constexpr int NS1 = LDS_BYTES / ST1;                        // 4 stages
    constexpr int KS1 = C / CH;                                 // 16
    constexpr int LA = XS / (NW * 1024), LB1 = WS1 / (NW * 1024), LPT1 = LA + LB1;   // LDS-DMA instructions per thread and stage: 3 + 2
    constexpr int SLOT = P * ROWB;                              // 16 KiB weight slot of phases 2 and 3
    constexpr int NS = (LDS_BYTES - Y1_BYTES) / SLOT;           // 4 ring slots behind y1
    constexpr int LS = SLOT / (NW * 1024);                      // 2 per thread
    constexpr int NTAP = 9, NCH3 = C / P;                       // 4 tail chunks of 128 channels
    constexpr int NS2 = NTAP * KC, NS3 = NCH3 * KC, NSLOT = NS2 + NS3;   // 36 (tap, chunk) slots + 16 (tail chunk, k chunk) slots
    static_assert(NS1 >= 3 && NS >= 3 && NS1 * ST1 <= LDS_BYTES && Y1_BYTES + NS * SLOT <= LDS_BYTES && (NS1 - 2) * LPT1 <= 16, "LDS plan");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];   // ONE array: a second __shared__ object makes hipcc drain vmcnt

    SMAP_TL_BEGIN
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int wm = wave >> 2, wn = wave & 3;                    // phases 2, 3: pixel half / 32-channel block; phase 1: row-block third / channel block

    // ================================================================= phase 1: y1 = relu(W1 x + b1) on the halo patch
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ w1g = reinterpret_cast<const char*>(a.w0);
    // a wave-wide LDS-DMA covers 8 rows x 128 B: lane -> row lane / 8, 16-byte slot lane % 8; slot s of row r holds logical granule
    // s ^ ((r >> 1) & 7) (0..3 = hi channels 0..31 of the stage in eights, 4..7 = lo): conv3.hip's patch rows
    const int srow = wave * 8 + (lane >> 3);
    const int gl = (lane & 7) ^ ((srow >> 1) & 7);              // rounds are 64 rows: (row >> 1) & 7 == (srow >> 1) & 7
    unsigned a_off[LA];                                         // patch row -> byte offset of its granule (0 = zero page)
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int prow = i * 64 + srow;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        a_off[i] = 0;
        if (prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const long long e = ((long long)(b * a.H + iy) * a.W + ix) * a.in_stride_c + (gl & 3) * 8 + (gl >> 2) * a.in_lo;
            a_off[i] = (unsigned)(a.in_off + e * 2);
        }
    }
    const unsigned wlane = (unsigned)(wave * 1024 + lane * 16);
    auto issue1 = [&](int st, int ks) {                         // stage st <- channels 32 ks .. +31 of x and of W1
        char* sX = smem + st * ST1;
        const char* gA = arena + (unsigned)(ks * CH * 2);       // invalid rows: zero page + stage offset
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVC_ABLATE & 1) ? 0 : LA); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_off[i]), (lds_void*)(sX + (i * 64 + wave * 8) * ROWB), 16, 0, 0);
        const char* gW = w1g + (long long)ks * WS1 + wlane;
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVC_ABLATE & 8) ? 0 : LB1); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gW + i * (NW * 1024)), (lds_void*)(sX + XS + i * (NW * 1024) + wave * 1024), 16, 0, 0);
    };
    // centre pixels of this lane in phases 2 and 3: p = wm*64 + mi*32 + l31 -> patch row of the pixel itself
    int crow[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * 64 + mi * 32 + l31;
        crow[mi] = (p / TW + 1) * PW + (p % TW) + 1;
    }
    // the residual of the last epilogue, collected while x passes through LDS: rs[nc][mi][j][plane] = channels
    // nc*128 + wn*32 + 16*j + 8*lhi .. +7 of pixel (mi, l31) -- the layout the register epilogue of phase 3 ends in
    // Chunks 0 .. NRS-1 only: 128 + the 48 accumulators of phase 1 + fragments and addresses do not fit 256 registers without scratch,
    // the last chunk's residual is read again (from L2 / Infinity Cache) while that chunk is multiplied: 3584 instead of 4096 B/px saved.
    constexpr int NRS = SMAP_CONVC_NRS;
    static_assert(NRS >= 0 && NRS < NCH3, "residual chunks held in registers");
    half8 rs[NRS > 0 ? NRS : 1][MI][2][2];

    // Biases enter through the ACCUMULATORS (acc = b / 2^-s before the first MFMA; the power-of-two scale makes that exact): an
    // ordinary global load in the middle of the LDS-DMA pipeline would make hipcc drain the whole queue (vmcnt(0)) at its use.
    float4 b1raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b1raw[q] = *reinterpret_cast<const float4*>(a.bias0 + wn * 32 + 8 * q + 4 * lhi);
    const float b3_mine = a.bias2[tid];                         // tail bias: one value per thread, parked until the table can be written
#pragma unroll
    for (int st = 0; st < NS1 - 1; ++st) issue1(st, st);        // the first stages go out behind the bias loads
    constexpr int NB1 = 3;                                      // 32 x 32 blocks of y1 per wave: channel block wn, patch-row blocks 3 wm + j
    f32x16 acc1[NB1];
    {
        const float inv0 = 1.f / a.acc_scale0;
#pragma unroll
        for (int j = 0; j < NB1; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc1[j][4 * q + 0] = b1raw[q].x * inv0; acc1[j][4 * q + 1] = b1raw[q].y * inv0;
                acc1[j][4 * q + 2] = b1raw[q].z * inv0; acc1[j][4 * q + 3] = b1raw[q].w * inv0;
            }
    }
    const int fswz = (l31 >> 1) & 7;                            // (row >> 1) & 7 of every fragment row = multiple of 32 + l31

#if SMAP_CONVC_PIPE
    // SOFTWARE PIPELINE (round 6).  Round 5's loops read a K step's fragments, waited for them and multiplied, step after step: hipcc
    // turns that into `ds_read x2-4, s_waitcnt lgkmcnt(0), mfma x1-4` groups -- an LDS round trip exposed five or six times per barrier
    // interval, in BOTH waves of a SIMD at once (they walk in step), which is where the tile's 105k non-MFMA cycles of 163k came from
    // (EXPERIMENTS R4.1b, R6.1).  Now every phase keeps TWO fragment sets: the reads of K step u + 1 go out, then step u is multiplied
    // from registers that landed half a barrier interval ago.  The first reads behind a barrier hide under the previous interval's last
    // MFMAs.  Same MFMAs in the same order: bit-identical results.  The second set costs 32 / 24 registers; they come from the residual
    // (SMAP_CONVC_NRS = 2 chunks in registers instead of 3).
    half8 w1a[2], x1a[NB1][2], w1b[2], x1b[NB1][2];
    auto rd1 = [&](int ks, int kk, half8 (&wf)[2], half8 (&xf)[NB1][2]) {
        const char* sX = smem + (ks % NS1) * ST1;
        const char* sW = sX + XS;
        const int g = kk * 2 + lhi;
        const int slot0 = (g ^ fswz) << 4, slot1 = ((g + 4) ^ fswz) << 4;
        wf[0] = *reinterpret_cast<const half8*>(sW + (wn * 32 + l31) * ROWB + slot0);
        wf[1] = *reinterpret_cast<const half8*>(sW + (wn * 32 + l31) * ROWB + slot1);
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            const char* xr = sX + ((wm * NB1 + j) * 32 + l31) * ROWB;
            xf[j][0] = *reinterpret_cast<const half8*>(xr + slot0);
            xf[j][1] = *reinterpret_cast<const half8*>(xr + slot1);
        }
    };
    auto mm1 = [&](const half8 (&wf)[2], const half8 (&xf)[NB1][2]) {      // small cross terms first, then hi*hi (conv3.hip's order)
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            acc1[j] = MFMA_(wf[0], xf[j][1], acc1[j]);
            acc1[j] = MFMA_(wf[1], xf[j][0], acc1[j]);
            acc1[j] = MFMA_(wf[0], xf[j][0], acc1[j]);
        }
    };
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        if (ks + NS1 - 1 <= KS1) wait_vm((NS1 - 2) * LPT1);     // stage ks has landed; younger stages stay in flight
        else wait_vm((KS1 - 1 - ks) * LPT1);
        lds_barrier();
        if (ks + NS1 - 1 < KS1) issue1((ks + NS1 - 1) % NS1, ks + NS1 - 1);     // into the buffer stage ks-1 was read from
        rd1(ks, 0, w1a, x1a);
        if ((ks >> 2) < NRS && (ks & 3) == wn) {                // this wave's residual channels: nc*128 + wn*32 + .. = stage 4 nc + wn
            const char* sX = smem + (ks % NS1) * ST1;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int cr = crow[mi];
                asm volatile("" : "+v"(cr));                    // addresses computed HERE, four times per wave (registers)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        rs[ks >> 2][mi][j][pl] = *reinterpret_cast<const half8*>(sX + cr * ROWB + (((2 * j + lhi + 4 * pl) ^ ((cr >> 1) & 7)) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks > 0) mm1(w1b, x1b);                              // K step (ks - 1, 1)
        __builtin_amdgcn_sched_barrier(0);
        lds_reads_landed();                                     // set a, under the MFMAs above (hipcc would wait for a AND b after the next reads)
        rd1(ks, 1, w1b, x1b);
        __builtin_amdgcn_sched_barrier(0);
        mm1(w1a, x1a);                                          // K step (ks, 0)
        __builtin_amdgcn_sched_barrier(0);
    }
    mm1(w1b, x1b);
#else
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        if (ks + NS1 - 1 <= KS1) wait_vm((NS1 - 2) * LPT1);     // stage ks has landed; younger stages stay in flight
        else wait_vm((KS1 - 1 - ks) * LPT1);
        lds_barrier();
        if (ks + NS1 - 1 < KS1) issue1((ks + NS1 - 1) % NS1, ks + NS1 - 1);     // into the buffer stage ks-1 was read from
        const char* sX = smem + (ks % NS1) * ST1;
        const char* sW = sX + XS;
        if ((ks >> 2) < NRS && (ks & 3) == wn) {                // this wave's residual channels: nc*128 + wn*32 + .. = stage 4 nc + wn
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int cr = crow[mi];
                asm volatile("" : "+v"(cr));                    // addresses computed HERE, four times per wave: precomputed for all stages they
#pragma unroll                                                  // would cost the 8-16 registers the residual does not leave
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        rs[ks >> 2][mi][j][pl] = *reinterpret_cast<const half8*>(sX + cr * ROWB + (((2 * j + lhi + 4 * pl) ^ ((cr >> 1) & 7)) << 4));
            }
        }
#pragma unroll
        for (int kk = 0; kk < CH / 16; ++kk) {
            const int g = kk * 2 + lhi;
            half8 wf[2];
            const int slot0 = (g ^ fswz) << 4, slot1 = ((g + 4) ^ fswz) << 4;
            wf[0] = *reinterpret_cast<const half8*>(sW + (wn * 32 + l31) * ROWB + slot0);
            wf[1] = *reinterpret_cast<const half8*>(sW + (wn * 32 + l31) * ROWB + slot1);
#pragma unroll
            for (int j = 0; j < NB1; ++j) {                     // one block's fragments at a time (the residual already holds 128 registers);
                const char* xr = sX + ((wm * NB1 + j) * 32 + l31) * ROWB;      // small cross terms first, then hi*hi (conv3.hip's order)
                const half8 xh = *reinterpret_cast<const half8*>(xr + slot0);
                const half8 xl = *reinterpret_cast<const half8*>(xr + slot1);
                acc1[j] = MFMA_(wf[0], xl, acc1[j]);
                acc1[j] = MFMA_(wf[1], xh, acc1[j]);
                acc1[j] = MFMA_(wf[0], xh, acc1[j]);
            }
        }
    }
#endif
    lds_barrier();                                              // every wave is done with the staging buffers (all DMA has landed)
    // phase 2's accumulators start at b2 / scale: the loads go out now, ahead of the first weight slots
    float4 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = *reinterpret_cast<const float4*>(a.bias + wn * 32 + 8 * q + 4 * lhi);

    // ---- weight slots of phases 2 and 3, 16 KiB each (128 rows of one 32-channel chunk): 36 (tap, chunk) slots, then 16 (tail chunk, k chunk)
    char* ring = smem + Y1_BYTES;
    const char* __restrict__ w2g = reinterpret_cast<const char*>(a.w);
    const char* __restrict__ w3g = reinterpret_cast<const char*>(a.w2);
    auto issue_slot = [&](int s) {
        char* dst = ring + (s % NS) * SLOT + wave * 1024;
        // slot s < 36: tap s / 4, chunk s % 4 of the 3x3 (conv3.hip's blocks are ordered [chunk][tap]); then [tail chunk][k chunk]
        const char* g = (s < NS2 ? w2g + (long long)((s % KC) * NTAP + s / KC) * SLOT : w3g + (long long)(s - NS2) * SLOT) + wlane;
#pragma unroll
        for (int i = 0; i < ((SMAP_CONVC_ABLATE & 8) ? 0 : LS); ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(g + i * (NW * 1024)), (lds_void*)(dst + i * (NW * 1024)), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_slot(s);
    // Ring protocol.  Slot s: [barrier: slot s is published, slot s-1's buffer is free] -> issue slot s + NS - 1 -> multiply ->
    // wait for slot s + 1; younger ones stay in flight.
    auto wait_next = [&](int cur, int extra = 0) {              // in slot `cur` (its issue done): slot cur + 1 has landed; `extra` younger
        if (cur + 1 >= NSLOT) return;                           // plain loads (the last chunk's residual) may stay in flight too
        const int issued = cur + NS - 1 < NSLOT ? cur + NS - 1 : NSLOT - 1;
        const int upto = cur + 1 < NSLOT ? cur + 1 : NSLOT - 1;
        wait_vm((issued > upto ? issued - upto : 0) * LS + extra);
    };
    // ---- accumulators -> y1 [KC][PROWS][128 B] (rows = patch pixels, conv3.hip's format).  acc[4*q + e] = channel
    //      nb*32 + 8*q + 4*lhi + e of patch row mb*32 + l31; rows outside the image are the 3x3's zero padding.
    char* sY1 = smem;
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int prow = (wm * NB1 + j) * 32 + l31;
        const int py = prow / PW, px = prow - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool live = prow < PH * PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int sw = (prow >> 1) & 7;
        char* row = sY1 + (wn * PROWS + prow) * ROWB + lhi * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc1[j][4 * q + e] * a.acc_scale0;    // (bias inside: accumulator start value)
                x = x < 0.f ? 0.f : x;                          // NaN stays NaN (torch's ReLU)
                x = live ? x : 0.f;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<half4*>(row + ((q ^ sw) << 4)) = h;
            *reinterpret_cast<half4*>(row + (((q + 4) ^ sw) << 4)) = l;
        }
    }
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

    // ================================================================= phase 2: the 3x3 on y1
    int prow0[MI];                                              // patch row of tap (0,0) of this lane's pixels
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) prow0[mi] = crow[mi] - PW - 1;
    const int c_row0 = wn * 32 + l31;                           // this wave's 32 output channels: rows of a weight slot

#if SMAP_CONVC_PIPE
    half8 a2a[2][MI], b2a[2], a2b[2][MI], b2b[2];
    auto rd2 = [&](int s, int kk, half8 (&af)[2][MI], half8 (&bf)[2]) {
        const char* sB = ring + (s % NS) * SLOT;
        const int tap = s / KC, cc = s % KC;
        const int shift = (tap / 3) * PW + (tap % 3);
        const int g = kk * 2 + lhi;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int prow = prow0[mi] + shift;
                af[pl][mi] = *reinterpret_cast<const half8*>(sY1 + (cc * PROWS + prow) * ROWB + (((g + 4 * pl) ^ ((prow >> 1) & 7)) << 4));
            }
            bf[pl] = *reinterpret_cast<const half8*>(sB + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
        }
    };
    auto mm2 = [&](const half8 (&af)[2][MI], const half8 (&bf)[2]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            acc2[mi] = MFMA_(bf[0], af[1][mi], acc2[mi]);
            acc2[mi] = MFMA_(bf[1], af[0][mi], acc2[mi]);
            acc2[mi] = MFMA_(bf[0], af[0][mi], acc2[mi]);
        }
    };
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
        lds_barrier();                                          // slot s landed for every wave; y1 complete (s = 0); slot s-1's buffer is free
        if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
        rd2(s, 0, a2a, b2a);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) mm2(a2b, b2b);                               // K step (s - 1, 1)
        __builtin_amdgcn_sched_barrier(0);
        lds_reads_landed();
        rd2(s, 1, a2b, b2b);
        __builtin_amdgcn_sched_barrier(0);
        mm2(a2a, b2a);                                          // K step (s, 0)
        __builtin_amdgcn_sched_barrier(0);
        wait_next(s);
    }
    mm2(a2b, b2b);
#else
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
        lds_barrier();                                          // slot s landed for every wave; y1 complete (s = 0); slot s-1's buffer is free
        if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
        const char* sB = ring + (s % NS) * SLOT;
        const int tap = s / KC, cc = s % KC;
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
                bf[pl] = *reinterpret_cast<const half8*>(sB + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc2[mi] = MFMA_(bf[0], af[1][mi], acc2[mi]);
                acc2[mi] = MFMA_(bf[1], af[0][mi], acc2[mi]);
                acc2[mi] = MFMA_(bf[0], af[0][mi], acc2[mi]);
            }
        }
        wait_next(s);
    }
#endif
    lds_barrier();                                              // every wave is done with y1: y2 may overwrite it

    // ---- accumulators -> y2 [KC][BM][128 B] (rows = tile pixels).  acc2[mi][4*q + e] = channel wn*32 + 8*q + 4*lhi + e
    char* sY2 = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int p = wm * 64 + mi * 32 + l31;
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
        const int p = wm * 64 + mi * 32 + l31;
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        m_ok[mi] = oy < a.Ho && ox < a.Wo;
        const unsigned m = m_ok[mi] ? (unsigned)((b * a.Ho + oy) * a.Wo + ox) : 0u;
        m_dense[mi] = m * (unsigned)(2 * a.tail_cout8);
        m_out[mi] = m * (unsigned)a.out_stride_c + (unsigned)a.out_c_off;
    }
    const int p_row0 = wm * 64 + l31;                           // + mi*32: pixel rows of y2
    _Float16* __restrict__ outp = reinterpret_cast<_Float16*>(a.out);

#if SMAP_CONVC_PIPE
    // Same pipeline; the epilogue of chunk nc runs half a slot late (behind the first reads of chunk nc + 1's first slot), the residual
    // loads of a chunk that is not held in registers go out behind the previous chunk's epilogue (one register set).
    f32x16 acc3[MI];                                            // start value b3 / scale: rows = channels nc*128 + wn*32 + 8*q + 4*lhi + e
    half8 rl[MI][2][2];                                         // residual of a chunk that is not held in registers (nc >= NRS)
    half8 p3a[2][MI], w3a[2], p3b[2][MI], w3b[2];
    auto rd3 = [&](int s, int kc, int kk, half8 (&pf)[2][MI], half8 (&wf)[2]) {
        const char* sW = ring + (s % NS) * SLOT;
        const int g = kk * 2 + lhi;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                pf[pl][mi] = *reinterpret_cast<const half8*>(sY2 + (kc * BM + p_row0 + mi * 32) * ROWB + (((g + 4 * pl) ^ fswz) << 4));
            wf[pl] = *reinterpret_cast<const half8*>(sW + c_row0 * ROWB + (((g + 4 * pl) ^ fswz) << 4));
        }
    };
    auto mm3 = [&](const half8 (&pf)[2][MI], const half8 (&wf)[2]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            acc3[mi] = MFMA_(wf[0], pf[1][mi], acc3[mi]);
            acc3[mi] = MFMA_(wf[1], pf[0][mi], acc3[mi]);
            acc3[mi] = MFMA_(wf[0], pf[0][mi], acc3[mi]);
        }
    };
    // The bias table is read with RAW ds_read_b128: behind an ordinary LDS load hipcc puts s_waitcnt vmcnt(0) when LDS-DMA is in flight (it
    // cannot tell the table from the ring the DMA writes), which drained the three weight slots in flight -- and the residual loads just
    // issued -- once per chunk.  The raw reads are followed by their own lgkmcnt(0) and a scheduling fence (the compiler does not know
    // that the outputs of the asm need the wait).
    const unsigned sB3_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)(smem + Y2_BYTES) + (unsigned)((wn * 32 + 4 * lhi) * 4);
    auto init_acc = [&](int nc) {
        f32x4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            asm volatile("ds_read_b128 %0, %1" : "=v"(b4[q]) : "v"(sB3_lds + (unsigned)((nc * P + 8 * q) * 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                acc3[mi][4 * q + 0] = b4[q][0]; acc3[mi][4 * q + 1] = b4[q][1]; acc3[mi][4 * q + 2] = b4[q][2]; acc3[mi][4 * q + 3] = b4[q][3];
            }
    };
    auto load_rl = [&](int nc) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    rl[mi][j][pl] = *reinterpret_cast<const half8*>(a.res + m_dense[mi] + (unsigned)(nc * P + wn * 32 + 8 * lhi + 16 * j) + pl * a.tail_cout8);
    };
    auto epilogue = [&](int nc) {
        // ---- register epilogue (convp.hip): half-wave swap -> acc3[mi][8*j .. 8*j+7] = channels n_lane + 16*j .. +7 of the pixel
        const int n_lane = nc * P + wn * 32 + 8 * lhi;
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
                for (int e = 0; e < 8; ++e)
                    acc3[mi][8 * j + e] += nc < NRS ? (float)rs[nc < NRS ? nc : 0][mi][j][0][e] + (float)rs[nc < NRS ? nc : 0][mi][j][1][e]
                                                    : (float)rl[mi][j][0][e] + (float)rl[mi][j][1][e];
        if (a.relu) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mi][r] = acc3[mi][r] < 0.f ? 0.f : acc3[mi][r];
        }
        auto add_tensor = [&](const _Float16* __restrict__ tsr) {       // post-ReLU skip adds of the last block of a layer; one pixel block at
#pragma unroll                                                          // a time: the residual registers leave room for 16 more, not 32
            for (int mi = 0; mi < MI; ++mi) {
                half8 h[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        h[j][pl] = *reinterpret_cast<const half8*>(tsr + m_dense[mi] + (unsigned)(n_lane + 16 * j) + pl * a.tail_cout8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc3[mi][8 * j + e] += (float)h[j][0][e] + (float)h[j][1][e];
            }
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
                if (SMAP_CONVC_ABLATE & 4) { if (h[0] == (_Float16)123.25f && l[1] == (_Float16)77.5f) *reinterpret_cast<half8*>(op) = h; continue; }   // keep the values live
                *reinterpret_cast<half8*>(op) = h;
                *reinterpret_cast<half8*>(op + a.out_lo) = l;
            }
    };
    // Counted waits name LOADS only (LDS-DMA and the residual loads; in-order among themselves), never stores: "slot s + 1 has landed" =
    // at most the loads issued after its request are outstanding.  Those are the requests of slots s + 2 and s + 3 plus the residual loads
    // of this chunk when they went out in between (they are issued in the chunk's first iteration, behind that iteration's wait).
    auto wait_next3 = [&](int s, int nc, int kc) {
        if (s + 1 >= NSLOT) return;
        const int younger = (s + 2 < NSLOT ? LS : 0) + (s + 3 < NSLOT ? LS : 0) + ((nc >= NRS && (kc == 1 || kc == 2)) ? MI * 4 : 0);
        wait_vm(younger);
    };
#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int s = NS2 + nc * KC + kc;
            lds_barrier();                                      // slot s landed for every wave; y2 + bias table complete (first slot); slot s-1's buffer is free
            if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
            rd3(s, kc, 0, p3a, w3a);
            __builtin_amdgcn_sched_barrier(0);
            if (s > NS2) mm3(p3b, w3b);                         // K step (s - 1, 1): the last one of chunk nc - 1 when kc == 0
            __builtin_amdgcn_sched_barrier(0);
            if (kc == 0) {
                wait_next3(s, nc, kc);                          // BEFORE the epilogue's stores and this chunk's residual loads
                if (nc > 0) epilogue(nc - 1);
                if (nc >= NRS) load_rl(nc);
                init_acc(nc);
            }
            lds_reads_landed();
            rd3(s, kc, 1, p3b, w3b);
            __builtin_amdgcn_sched_barrier(0);
            mm3(p3a, w3a);                                      // K step (s, 0)
            __builtin_amdgcn_sched_barrier(0);
            if (kc != 0) wait_next3(s, nc, kc);
        }
    }
    mm3(p3b, w3b);
    epilogue(NCH3 - 1);
#else
#pragma unroll
    for (int nc = 0; nc < NCH3; ++nc) {
        f32x16 acc3[MI];                                        // start value b3 / scale: rows = channels nc*128 + wn*32 + 8*q + 4*lhi + e
        half8 rl[MI][2][2];                                     // residual of a chunk that is not held in registers (nc >= NRS)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int s = NS2 + nc * KC + kc;
            const char* sW = ring + (s % NS) * SLOT;
            auto requests = [&]() {
                if (s + NS - 1 < NSLOT) issue_slot(s + NS - 1);
                if (nc >= NRS && kc == 0) {                     // (after the slot request: younger than every LDS-DMA the counted waits below name)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl)
                                rl[mi][j][pl] = *reinterpret_cast<const half8*>(a.res + m_dense[mi] + (unsigned)(nc * P + wn * 32 + 8 * lhi + 16 * j) + pl * a.tail_cout8);
                }
            };
            auto init_acc = [&]() {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4*>(sB3 + nc * P + wn * 32 + 8 * q + 4 * lhi);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc3[mi][4 * q + 0] = b4.x; acc3[mi][4 * q + 1] = b4.y; acc3[mi][4 * q + 2] = b4.z; acc3[mi][4 * q + 3] = b4.w;
                    }
                }
            };
            lds_barrier();                                      // slot s landed for every wave; y2 + bias table complete (first slot); slot s-1's buffer is free
            requests();
            if (kc == 0) init_acc();
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
                    acc3[mi] = MFMA_(wf[0], pf[1][mi], acc3[mi]);
                    acc3[mi] = MFMA_(wf[1], pf[0][mi], acc3[mi]);
                    acc3[mi] = MFMA_(wf[0], pf[0][mi], acc3[mi]);
                }
            }
            wait_next(s, nc >= NRS ? MI * 4 : 0);               // (last k chunk: BEFORE this chunk's stores -- a counted vmcnt also counts stores)
        }
        // ---- register epilogue (convp.hip): half-wave swap -> acc3[mi][8*j .. 8*j+7] = channels n_lane + 16*j .. +7 of the pixel
        const int n_lane = nc * P + wn * 32 + 8 * lhi;
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
                for (int e = 0; e < 8; ++e)
                    acc3[mi][8 * j + e] += nc < NRS ? (float)rs[nc < NRS ? nc : 0][mi][j][0][e] + (float)rs[nc < NRS ? nc : 0][mi][j][1][e]
                                                    : (float)rl[mi][j][0][e] + (float)rl[mi][j][1][e];
        if (a.relu) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mi][r] = acc3[mi][r] < 0.f ? 0.f : acc3[mi][r];
        }
        auto add_tensor = [&](const _Float16* __restrict__ tsr) {       // post-ReLU skip adds of the last block of a layer; one pixel block at
#pragma unroll                                                          // a time: the residual registers leave room for 16 more, not 32
            for (int mi = 0; mi < MI; ++mi) {
                half8 h[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        h[j][pl] = *reinterpret_cast<const half8*>(tsr + m_dense[mi] + (unsigned)(n_lane + 16 * j) + pl * a.tail_cout8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc3[mi][8 * j + e] += (float)h[j][0][e] + (float)h[j][1][e];
            }
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
                if (SMAP_CONVC_ABLATE & 4) { if (h[0] == (_Float16)123.25f && l[1] == (_Float16)77.5f) *reinterpret_cast<half8*>(op) = h; continue; }   // keep the values live
                *reinterpret_cast<half8*>(op) = h;
                *reinterpret_cast<half8*>(op + a.out_lo) = l;
            }
    }
#endif
    SMAP_TL_END(a)
}

}  // namespace

// tile id 94 (smap_convb_tile_dims / smap_launch_convb in convb.hip route it here): 8 x 16 pixels, 128 planes, 512 channels
hipError_t smap_launch_convc(const ConvArgs& a, hipStream_t st)
{
    if (!a.x3 || a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.up || a.out_fp32 || !a.w0 || !a.w2 || a.wd || a.Cin != 128 || a.head_cin != 512 ||
        a.tail_cout8 != 512 || a.H != a.Ho || a.W != a.Wo)
        return hipErrorInvalidValue;
    const int B = a.M / (a.Ho * a.Wo);
    const int tiles_x = (a.Wo + 15) / 16, tiles_y = (a.Ho + 7) / 8;
    hipLaunchKernelGGL(bottleneck128_kernel, dim3(tiles_x * tiles_y * B), dim3(512), 0, st, a, tiles_x, tiles_y);
    return hipGetLastError();
}

// tools only (diagnostics builds with -DSMAP_DEBUG_EXPORTS): resident workgroups per CU the runtime reports
#ifdef SMAP_DEBUG_EXPORTS
extern "C" __attribute__((visibility("default"))) int smap_debug_convc_occupancy(void)
{
    int n = -1;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bottleneck128_kernel, 512, 0);
    return e == hipSuccess ? n : -1000 - (int)e;
}
#endif
