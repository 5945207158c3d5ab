// conv.hip -- conv_bn_relu (model/smap.py:13-45) as an implicit GEMM on the gfx950
// matrix cores, with folded BN, and bias + residual + ReLU + skip adds fused into
// the epilogue.  Hand-written for CDNA4: wave64, v_mfma_f32_32x32x16_f16,
// global_load_lds (LDS-DMA) staging with a source-side XOR swizzle, XCD-aware
// block order.
//
// GEMM view     D[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = (b, oy, ox) output pixel, M = B*Ho*Wo          (NHWC fp16 activations)
//   n = output channel, N = cout_pad                   (weights [cout_pad][KH][KW][Cin] fp16)
//   k = (kh, kw, cin), K = KH*KW*Cin, walked in BK=64 chunks; a chunk never straddles a
//       (kh,kw) position because Cin % 64 == 0, so the A rows of a chunk are 128
//       contiguous bytes of one input pixel -- or 16 bytes of zeros from a zero page
//       when the tap falls into the padding / past M.
//
// LDS image of a staged tile: [rows][64 halves] = 128-B rows, written by LDS-DMA
// (lane-linear: wave w, round i covers rows i*32+w*8 .. +8, lane l -> row l/8, 16-B
// slot l%8).  Slot s of row r holds K-granule s ^ ((r>>1)&7): the permutation is
// applied to the per-lane GLOBAL address (the DMA cannot scatter) and again on the
// ds_read_b128 side, which makes the 16-lane groups of ds_read_b128 conflict free.
//
// Epilogue: accumulators (+bias) go to LDS as an fp32 [BM][BN] tile, then every
// thread owns 8 consecutive channels of one pixel: residual (16-B load) -> ReLU ->
// post-ReLU skip adds -> one 16-B fp16 store (or two 16-B fp32 stores), i.e. full
// coalesced NHWC lines.  One rounding to fp16 per output value.
//
// Round 5: N SEGMENTS (smap_op.seg_*: the 1x1 convs of an Upsample_unit that read the same tensor, model/smap.py:210-241, as one
// launch -- the epilogue picks output tensor / ReLU / channel count / accumulator scale per N tile) and SPLIT K (smap_op.ksplit, template
// argument SPLITK: the long-K launches of batch-1 schedules, S workgroups per output tile, deterministic reduction by the last arriver).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "smap_hip.h"
#include "plan.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#ifndef SMAP_ABLATE
#define SMAP_ABLATE 0            // experiments only (tools/build_ablate.py): 1 no loads, 2 no MFMA, 4 no stores, 8 no epilogue, 32 no bilinear tap loads, 64 no bilinear index math
#endif
// FULL = false: epilogue with bias / residual / ReLU only (most layers: ~45 fewer VGPRs, more workgroups per CU);
// FULL = true : + fused bilinear add and post-ReLU addends.
// X3 = true  : fp32-equivalent arithmetic on the fp16 matrix cores (smap_op.precision = 1).  Every activation and weight
//              is stored as TWO fp16 planes hi = fp16(v), lo = fp16(v - hi) (22 significant bits; a pixel is [hi(C) | lo(C)],
//              the weight matrix [cout_pad][K] hi followed by [cout_pad][K] lo, pre-scaled by a power of two so that lo
//              stays in fp16's normal range) and a K step issues three MFMAs into the same fp32 accumulator:
//              hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-24 relative).  A K tile stages four LDS images
//              (A hi, A lo, W hi, W lo); the epilogue re-splits the fp32 result.  3x the MFMA work and 2x the bytes of
//              the fp16 mode, ~1e-6 relative error instead of ~1e-3: the mode whose output meets the reference's fp32.
// WM x WN = 4 or 8 waves.  Eight waves (two per SIMD from ONE workgroup) are for the low-resolution layers whose grids
// do not fill the chip: with <= 256 workgroups of 4 waves there is no second wave on the SIMD to multiply while the first
// one waits for its fragments, its LDS-DMA requests or the barrier (a lone wave CAN issue an MFMA every 32 cycles:
// tools/ubench/mfma_issue.hip; what it cannot do is hide its own stalls).
// SPLITK = true: the instance that also takes a.ksplit > 1 (its own template argument: the extra paths cost the plain instances
// registers and 4 bytes of LDS that tip 80 KiB tiles from two workgroups per CU to one).
// REGEPI = true (round 6, tile 56 = 256 x 256): REGISTER epilogue.  The fp32 [BM][BN] staging tile of the LDS epilogue would be 256 KiB, and
// what this tile is for -- halving the L2 -> LDS bytes per MFMA of the N >= 256 launches (341 B at 128 x 128, 170 B at 256 x 256) --
// needs all the LDS for its two 64 KiB stages.  The MFMA operands are swapped (weights first: accumulator rows = channels, columns =
// pixels), a half-wave swap (v_permlane32_swap) leaves every lane with 8 consecutive channels of one pixel, and bias (as the
// accumulators' start value, b / 2^-s: exact), residual, ReLU and the hi | lo split happen in registers: 16-byte loads and stores
// straight from / to the NHWC tensors (convp.hip's and convc.hip's epilogue).  Split precision, fp16 outputs, no fused bilinear add
// or post-ReLU addends, no split K; N segments as everywhere.
// DUAL = true (round 6, smap_op.in2_*): a SECOND input concatenated along K -- the K loop walks the first input's Cin / BK tiles, then the
// second input's Cin2 / BK tiles (other tensor, other pixel stride: the staging offsets are switched once), into the SAME accumulators; the
// packed weight matrix is the two convs' matrices side by side.  A stride-2 Bottleneck's last 1x1 and its shortcut 1x1 as one launch
// (model/smap.py:60-77).  1x1 only, plain epilogue, no split K.
// TAPDOT = true (round 6, smap_op.tap_n; tile 54 = 128 x 256, ONE N tile): the launch's activation y = act(W x + b) [pixel][256] never reaches
// memory.  Its only consumer is a 3x3 conv with ONE output channel (model/smap.py:227-229: res_rd_conv2 on res_rd_conv1's output), and
//     conv3x3(y)[p] = sum over taps of  < w[tap], y[p + offset(tap)] >
// so the epilogue stores t[p][tap] = < w[tap], y[p] > -- nine fp32 numbers per pixel instead of 256 hi | lo pairs -- and a nine-term stencil
// over t (plan.hip::tapsum_kernel) finishes the conv.  Dot products in fp32 on the fp32 accumulators (nothing is rounded to fp16 in between).
// RELUSUM = true (with DUAL; smap_op.in2_mode = 1): out = relu(W1 x + b1) + relu(W2 x2 + b2).  When the K loop reaches the second input's first
// tile the accumulators become relu(acc * s1 + b1) and are parked in registers (as many again as the accumulators: 32 per lane on the
// eight-wave 128 x 128 tiles), the accumulators restart at zero; the LDS epilogue tile receives relu(acc * s2 + b2) + parked.
template <int BM, int BN, int WM, int WN, int STAGES, int BK, bool FULL, bool X3, bool SPLITK = false, bool REGEPI = false, bool DUAL = false, bool TAPDOT = false,
          bool RELUSUM = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_kernel(const ConvArgs a)
{
    static_assert(!TAPDOT || (!FULL && !SPLITK && !REGEPI && !DUAL), "tap-dot epilogue: plain instance only");
    static_assert(!RELUSUM || DUAL, "relu-sum: a second input");
    static_assert(!REGEPI || (X3 && !FULL && !SPLITK), "register epilogue: split precision, plain epilogue, no split K");
    static_assert(!DUAL || (!FULL && !SPLITK && !REGEPI), "second input: plain epilogue, no split K");
    constexpr int NPL = X3 ? 2 : 1;                    // fp16 planes per operand
    constexpr int NW = WM * WN, NT = NW * 64;          // waves, threads
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(STAGES >= 2 && STAGES <= 4, "2..4 LDS stages");
    static_assert(BK == 32 || BK == 64, "BK = halves per K chunk");
    constexpr int ROWB = BK * 2;                       // bytes per LDS row
    constexpr int SPR = ROWB / 16;                     // 16-byte slots per row (8 / 4)
    constexpr int RPW = 64 / SPR;                      // rows one wave-wide LDS-DMA instruction covers (8 / 16)
    constexpr int RPR = NW * RPW;                      // rows per round of all waves (32 / 64 with 4 waves)
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile must be a multiple of the DMA round");
    constexpr int LA = BM / RPR, LB = BN / RPR;
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    static_assert(MI >= 1 && NI >= 1, "tile too small for the wave grid");
    constexpr int STAGE = NPL * (BM + BN) * ROWB;      // [A planes][B planes]
    constexpr int LPT = NPL * (LA + LB);               // LDS-DMA loads per thread per K tile
    static_assert((STAGES - 2) * LPT <= 63, "vmcnt is 6 bits");
    constexpr int CSS = TAPDOT ? BN + 8 : BN;                 // row stride (floats) of the fp32 epilogue tile (tap-dot: padded, its rows are read as MFMA fragments)
    constexpr int LDS_BYTES = REGEPI ? STAGES * STAGE : (STAGES * STAGE > BM * CSS * 4 ? STAGES * STAGE : BM * CSS * 4) + (SPLITK ? 16 : 0);   // pipeline | fp32 epilogue tile (+ the split-K flag)
    static_assert(LDS_BYTES <= 160 * 1024, "LDS is 160 KiB per CU");
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    SMAP_TL_BEGIN
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
#ifdef SMAP_TRACE
    long long tr_t[8]; long long tr_wait = 0;
    tr_t[0] = __builtin_amdgcn_s_memtime();
#define TR(i) tr_t[i] = __builtin_amdgcn_s_memtime()
#else
#define TR(i)
#endif

    // ---- XCD-aware block order: blocks b, b+8, b+16.. run on one XCD; give each XCD a
    //      contiguous range of logical tiles so that the N-tiles of one M-tile (which
    //      re-read the same activation rows) share an L2.
    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // SPLIT K (a.ksplit = S > 1: the launches of small schedules whose m_tiles x n_tiles do not fill the chip): S consecutive workgroups
    // -- neighbours in the logical order, hence mostly on one XCD / one L2 -- share an output tile and take the K tiles
    // [ks n / S, (ks + 1) n / S) each; their raw accumulators meet in a scratch buffer and the LAST one to arrive (a ticket per tile) sums
    // them in the fixed order 0 .. S-1 and runs the epilogue: deterministic, no second launch (epilogue 1 below).
    const int S = SPLITK ? a.ksplit : 1;
    const int tile_id = S > 1 ? logical / S : logical, ks = logical - tile_id * S;
    const int m_tile = tile_id / a.n_tiles, n_tile = tile_id - m_tile * a.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    // ---- N segments (smap_op.seg_*: several 1x1 convs on one input as one launch): the output tensor, channel count, ReLU flag and
    //      accumulator scale of this workgroup's rows of the weight matrix.  Wave-uniform (from blockIdx): SGPR selects.  Residual,
    //      post-ReLU addends and the fused bilinear add belong to segment 0.
    const int seg = (n0 >= a.seg_n1 ? 1 : 0) + (n0 >= a.seg_n2 ? 1 : 0);
    const int nb = n0 - (seg == 0 ? 0 : seg == 1 ? a.seg_n1 : a.seg_n2);           // first channel of this tile inside its output tensor
    void* const o_out = seg == 0 ? a.out : seg == 1 ? a.seg_out1 : a.seg_out2;
    const int o_cout8 = seg == 0 ? a.Cout8 : seg == 1 ? a.seg_cout8_1 : a.seg_cout8_2;
    const int o_stride = seg == 0 ? a.out_stride_c : seg == 1 ? a.seg_stride1 : a.seg_stride2;
    const int o_c_off = seg == 0 ? a.out_c_off : 0;
    const int o_relu = seg == 0 ? a.relu : seg == 1 ? a.seg_relu1 : a.seg_relu2;
    const float o_scale = seg == 0 ? a.acc_scale : seg == 1 ? a.seg_scale1 : a.seg_scale2;
    const int o_lo = o_stride >> 1;
    const _Float16* const o_res = seg == 0 ? a.res : nullptr;
    const _Float16* const o_up = seg == 0 ? a.up : nullptr;
    const _Float16* const o_add1 = seg == 0 ? a.add1 : nullptr;
    const _Float16* const o_add2 = seg == 0 ? a.add2 : nullptr;

    // ---- per-thread staging geometry.  Every global address of the K loop is
    //      (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset in one VGPR): the A base is
    //      the ARENA (whose first 256 bytes are zeros = the padding "zero page"), the B base the
    //      weight matrix.  Per K tile only the uniform part moves, so the loop body carries no
    //      per-lane address arithmetic; the per-lane offsets change once per (kh,kw) tap.
    const int lrow = lane / SPR, lslot = lane % SPR;
    const int srow = wave * RPW + lrow;                          // row inside a DMA round
    const int gch = BK == 64 ? (lslot ^ ((srow >> 1) & 7)) : (lslot ^ ((srow >> 2) & 3));   // K-granule this lane fetches
    const char* __restrict__ arena = reinterpret_cast<const char*>(a.arena);
    const char* __restrict__ wt = reinterpret_cast<const char*>(a.w);
    const int HoWo = a.Ho * a.Wo;

    // Weight tiles: the host packs every (n tile, K tile) weight tile as one contiguous block in LDS order, swizzle included
    // (smap_amd/engine.py::pack_conv_weights), so a wave-wide LDS-DMA reads ONE contiguous KiB -- strided 64-byte row
    // segments of a [cout][K] matrix stream from L2 at half that rate (tools/ubench/lds_dma_rows.hip).  The weight half of
    // K tile 0 goes out at once: it does not depend on the pixel arithmetic below, its L2 latency overlaps the set-up.
    // (uniform 64-bit base in SGPRs) + (32-bit per-lane offset in one VGPR), like the activation side: no per-lane 64-bit math.
    // BK = 32 tiles are packed in PAIRS of K tiles (128-byte rows = [tile 2p | tile 2p+1]): the line fetched for one K tile
    // already holds the next one's bytes -- what the latency-bound small launches (batch 1) live on.
    const bool wpair = BK == 32 && a.w_pairs;                   // wave-uniform
    const int wrow = wpair ? 128 : ROWB;                        // bytes of a packed weight row
    const int wblk = NPL * BN * wrow;                           // bytes of one packed block (pairs: two K tiles)
    unsigned w_off[NPL * LB];
#pragma unroll
    for (int j = 0; j < NPL * LB; ++j) w_off[j] = (unsigned)((j * RPR + wave * RPW + lrow) * wrow + lslot * 16);
    const char* __restrict__ wt_tile = wt + (long long)n_tile * (wpair ? a.K / 64 : a.K / BK) * wblk;
    auto issue_b = [&](int buf, int it) {
        char* sB = smem + buf * STAGE + NPL * BM * ROWB;
        const char* gB = wpair ? wt_tile + (long long)(it >> 1) * wblk + (it & 1) * 64 : wt_tile + (long long)it * wblk;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NPL * LB; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + w_off[j]), (lds_void*)(sB + (j * RPR + wave * RPW) * ROWB), 16, 0, 0);
    };
    // (the K range of a split-K workgroup is known only further down: its first weight tile goes out there)
    if (!(SMAP_ABLATE & 1) && !SPLITK) issue_b(0, 0);

    // Per staged A row: byte offset (from the arena base) of tap (0,0), channel granule gch, and the mask
    // of in-range taps (bit kh*ksize+kw).  m -> (b, oy, ox) costs two integer divisions ONCE per thread;
    // the thread's other rows are 32 pixels further along the raster (carry propagation), and the tap
    // mask is the outer product of three row tests and three column tests (no loop over taps): the
    // set-up phase was 20-30 % of a short-K workgroup's lifetime (tools/trace_conv.py).
    unsigned a_off[LA];
    unsigned a_mask[LA];
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
                a_off[i] = (unsigned)(a.in_off + e * 2);       // wraps for taps above/left of the image; only
                unsigned vx = 0, mk = 0;                       // used (mod 2^32) when the tap itself is valid
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
    unsigned a_off2[DUAL ? LA : 1];                            // second input: byte offset of the sampled pixel's granule (0 = zero page: rows past M)
    if (DUAL) {
        int m = m0 + srow;
        int b = m / HoWo, rem = m - b * HoWo;
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            a_off2[i] = 0;
            if (m < a.M) {
                const long long e = ((long long)(b * a.H2 + oy * a.stride2) * a.W2 + ox * a.stride2) * a.in2_stride_c + gch * 8;
                a_off2[i] = (unsigned)(a.in2_off + e * 2);
            }
            m += RPR;
            ox += RPR;
            while (ox >= a.Wo) { ox -= a.Wo; ++oy; }
            while (oy >= a.Ho) { oy -= a.Ho; ++b; }
        }
    }
    const int cchunks = a.Cin / BK;
    const int cchunks2 = DUAL ? a.Cin2 / BK : 0;
    const int n_all = a.ksize * a.ksize * cchunks + cchunks2;  // K tiles of the op
    const int it_lo = S > 1 ? ks * n_all / S : 0;              // this workgroup's share (all of them without split K)
    const int n_iter = (S > 1 ? (ks + 1) * n_all / S : n_all) - it_lo;

    // staging cursor (all wave-uniform -> SGPRs): tap (s_kh, s_kw), channel chunk s_cc
    int s_kh = 0, s_kw = 0, s_cc = 0;
    int s_in2 = 0;                                             // DUAL: the cursor is in the second input's K tiles
    int s_it = it_lo;                                          // K tile the cursor stands on
    if (S > 1) {
        const int tap = it_lo / cchunks;
        s_cc = it_lo - tap * cchunks;
        s_kh = tap / a.ksize;
        s_kw = tap - s_kh * a.ksize;
    }
    unsigned a_cur[LA];                                        // per-lane offsets of the current tap (0 = zero page)
    auto set_tap = [&]() {
        const unsigned tap_off = (unsigned)((s_kh * a.W + s_kw) * a.in_stride_c * 2);
        const unsigned bit = 1u << (s_kh * a.ksize + s_kw);
#pragma unroll
        for (int i = 0; i < LA; ++i) a_cur[i] = (a_mask[i] & bit) ? a_off[i] + tap_off : 0u;
    };
    set_tap();
    auto issue_a = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            char* sA = smem + buf * STAGE + pl * BM * ROWB;
            // invalid taps: a_cur = 0 -> zero page + s_cc*ROWB (+ the lo-plane offset: the zero page covers both)
            const char* gA = arena + (unsigned)(s_cc * ROWB + (X3 ? pl * ((DUAL && s_in2) ? a.in2_lo : a.in_lo) * 2 : 0));
#pragma unroll
            for (int i = 0; i < LA; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(gA + a_cur[i]), (lds_void*)(sA + (i * RPR + wave * RPW) * ROWB), 16, 0, 0);
        }
    };
    auto advance = [&]() {
        ++s_it;
        if (DUAL && s_in2) { ++s_cc; return; }                 // (past the last tile nothing is issued)
        if (++s_cc == cchunks) {
            s_cc = 0;
            if (DUAL) {                // 1x1: the first input is through -> the second input's pixels from here on
                s_in2 = 1;
#pragma unroll
                for (int i = 0; i < LA; ++i) a_cur[i] = a_off2[i];
                return;
            }
            if (++s_kw == a.ksize) { s_kw = 0; ++s_kh; }
            set_tap();                 // past the last tap the mask bit is 0 -> offsets 0, never issued anyway
        }
    };
    auto stage = [&](int buf) {        // all loads of K tile t are issued before any load of tile t+1 (counted vmcnt)
        issue_a(buf);
        issue_b(buf, s_it);
        advance();
    };
    if (!(SMAP_ABLATE & 1)) {                                     // activation half of K tile 0 (+ its weight half in a split-K workgroup)
        if (SPLITK) issue_b(0, it_lo);
        issue_a(0);
        advance();
    }

    // ---- residual prefetch: the epilogue's residual tile (8 channels x PASSES pixels per thread) is
    //      requested before the K loop so that its HBM latency hides under the whole main loop
    //      (a pass-by-pass load in the epilogue exposes one full memory round trip per pass).
    constexpr int CG = BN / 8;                    // channel groups per row
    constexpr int PASSES = REGEPI ? 1 : BM * CG / NT;       // (register epilogue: the residual is loaded block by block down there)
    static_assert(BM * CG % NT == 0, "tile/thread mismatch");
    half8 rres[PASSES][NPL];
    auto load_res = [&]() {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int idx = p * NT + tid;
            const int row = idx / CG, cg = idx - row * CG;
            const int m = m0 + row, n = nb + cg * 8;
            const unsigned dense = (m < a.M && n < o_cout8) ? (unsigned)m * (unsigned)(NPL * o_cout8) + (unsigned)n : 0u;   // < 2^31 elements per tensor
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                rres[p][pl] = *reinterpret_cast<const half8*>(o_res + dense + pl * o_cout8);
        }
    };
    if (o_res && !SPLITK && !REGEPI) load_res();               // (split K: only the workgroup that runs the epilogue needs it)

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (REGEPI) {       // accumulator rows are CHANNELS here: row r of block ni = channel n0 + wn*(BN/WN) + ni*32 + (r&3) + 8*(r>>2) + 4*lhi; start value b / 2^-s
        const float inv = 1.f / o_scale;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + n0 + wn * (BN / WN) + ni * 32 + 8 * q + 4 * lhi);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    acc[mi][ni][4 * q + 0] = b4.x * inv; acc[mi][ni][4 * q + 1] = b4.y * inv;
                    acc[mi][ni][4 * q + 2] = b4.z * inv; acc[mi][ni][4 * q + 3] = b4.w * inv;
                }
            }
    }
    const int rswz = BK == 64 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    const int a_row0 = wm * (BM / WM) + l31;     // + mi*32
    const int b_row0 = wn * (BN / WN) + l31;     // + ni*32
    f32x16 park[RELUSUM ? MI : 1][RELUSUM ? NI : 1];          // relu(W1 x + b1) of this lane's accumulator elements
    float bias1[RELUSUM ? NI : 1];                            // (loaded HERE: a plain load inside the LDS-DMA pipeline would drain vmcnt)
    if (RELUSUM) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bias1[ni] = a.bias[n0 + wn * (BN / WN) + ni * 32 + l31];
    }

    // ---- STAGES-deep LDS-DMA pipeline.  Iteration `it` needs K tile `it`; tiles it+1 .. it+STAGES-2
    //      stay in flight across the barrier (counted vmcnt + raw s_barrier: a __syncthreads() here
    //      would drain the DMA queue).  RAW: every wave waits for its own slice of tile `it`, then the
    //      barrier publishes all slices.  WAR: the buffer refilled after the barrier held tile it-1,
    //      whose ds_reads were consumed by MFMAs that precede the barrier in program order.
    TR(1);
#pragma unroll
    for (int st = 1; st < STAGES - 1; ++st)                       // K tile 0 went out during the set-up
        if (st < n_iter && !(SMAP_ABLATE & 1)) stage(st);
    int buf = 0, nbuf = STAGES - 1;
    for (int it = 0; it < n_iter; ++it) {
#ifdef SMAP_TRACE
        const long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        if (it + STAGES - 1 <= n_iter) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SMAP_TRACE
        tr_wait += __builtin_amdgcn_s_memtime() - tw0;
        if (it == 0) tr_t[2] = __builtin_amdgcn_s_memtime();
#endif
        if (it + STAGES - 1 < n_iter && !(SMAP_ABLATE & 1)) stage(nbuf);
        if (RELUSUM && it == cchunks) {        // the first conv is complete: activate, park, restart
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = X3 ? acc[mi][ni][r] * o_scale + bias1[ni] : acc[mi][ni][r] + bias1[ni];
                        park[mi][ni][r] = v < 0.f ? 0.f : v;          // NaN stays NaN (torch's ReLU)
                        acc[mi][ni][r] = 0.f;
                    }
        }
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + NPL * BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < ((SMAP_ABLATE & 2) ? 0 : BK / 16); ++kk) {
            const int slot = ((kk * 2 + lhi) ^ rswz) * 16;
            half8 af[NPL][MI], bf[NPL][NI];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[pl][mi] = *reinterpret_cast<const half8*>(sA + (pl * BM + a_row0 + mi * 32) * ROWB + slot);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    bf[pl][ni] = *reinterpret_cast<const half8*>(sB + (pl * BN + b_row0 + ni * 32) * ROWB + slot);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if (REGEPI) {   // weights first: D rows = channels, columns = pixels (same products, same order)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0][ni], af[NPL - 1][mi], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[NPL - 1][ni], af[0][mi], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0][ni], af[0][mi], acc[mi][ni], 0, 0, 0);
                        continue;
                    }
                    if (X3) {       // small cross terms first, then hi*hi
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[NPL - 1][mi], bf[0][ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][mi], bf[NPL - 1][ni], acc[mi][ni], 0, 0, 0);
                    }
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][mi], bf[0][ni], acc[mi][ni], 0, 0, 0);
                }
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    TR(3);
    __syncthreads();   // everyone is done reading the staging buffers
    if (SMAP_ABLATE & 8) {
        if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(a.out)[tid] = acc[0][0][1];   // keep acc live
        return;
    }

    if constexpr (REGEPI) {
        // ---- register epilogue: per (pixel block mi, channel block ni) the half-wave swap turns acc[8j .. 8j+7] into channels
        //      n_lane + 16j .. +7 of pixel m0 + wm*(BM/WM) + mi*32 + l31; the residual of block (mi, ni) is requested one block ahead.
        _Float16* const outp = reinterpret_cast<_Float16*>(o_out);
        unsigned m_dense[MI], m_out[MI];
        bool m_ok[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (BM / WM) + mi * 32 + l31;
            m_ok[mi] = m < a.M;
            const unsigned ms = m_ok[mi] ? (unsigned)m : 0u;
            m_dense[mi] = ms * (unsigned)(NPL * o_cout8);
            m_out[mi] = ms * (unsigned)o_stride + (unsigned)o_c_off;
        }
        auto n_of = [&](int ni) { return nb + wn * (BN / WN) + ni * 32 + 8 * lhi; };        // + 16 j
        auto load_rr = [&](int mi, int ni, half8 (&rr)[2][NPL]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n_of(ni) + 16 * j;
                const unsigned d = n < o_cout8 ? m_dense[mi] + (unsigned)n : 0u;          // (clamped: the load stays in bounds)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) rr[j][pl] = *reinterpret_cast<const half8*>(o_res + d + pl * o_cout8);
            }
        };
        half8 rr[2][2][NPL];
        if (o_res) load_rr(0, 0, rr[0]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                constexpr int NB = MI * NI;
                const int blk = mi * NI + ni;
                if (o_res && blk + 1 < NB) load_rr((blk + 1) / NI, (blk + 1) % NI, rr[(blk + 1) & 1]);
                f32x16& c = acc[mi][ni];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xf = c[8 * j + e], yf = c[8 * j + 4 + e];
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                        const unsigned s0 = sw[0], s1 = sw[1];
                        c[8 * j + e] = o_scale * __uint_as_float(s0);                       // (bias inside)
                        c[8 * j + 4 + e] = o_scale * __uint_as_float(s1);
                    }
                if (o_res) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) c[8 * j + e] += (float)rr[blk & 1][j][0][e] + (float)rr[blk & 1][j][NPL - 1][e];
                }
                if (o_relu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = c[r] < 0.f ? 0.f : c[r];            // NaN stays NaN (torch's ReLU)
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n_of(ni) + 16 * j;
                    if (!m_ok[mi] || n >= o_cout8 || ((SMAP_ABLATE & 4) && a.M != 7)) continue;
                    _Float16* op = outp + (m_out[mi] + (unsigned)n);
                    half8 h, l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h[e] = (_Float16)c[8 * j + e];
                        l[e] = (_Float16)(c[8 * j + e] - (float)h[e]);
                    }
                    *reinterpret_cast<half8*>(op) = h;
                    *reinterpret_cast<half8*>(op + o_lo) = l;
                }
            }
        SMAP_TL_END(a)
        return;
    }

    // ---- epilogue 1: acc + bias -> fp32 [BM][BN] tile in LDS
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = wn * (BN / WN) + ni * 32 + l31;
        const float bias = RELUSUM ? a.bias_b[n0 + col] : a.bias[n0 + col];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / WM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (RELUSUM) {
                    const float v = X3 ? acc[mi][ni][r] * a.acc_scale_b + bias : acc[mi][ni][r] + bias;
                    Cs[row * CSS + col] = (v < 0.f ? 0.f : v) + park[mi][ni][r];
                    continue;
                }
                Cs[row * CSS + col] = S > 1 ? acc[mi][ni][r] : (X3 ? acc[mi][ni][r] * o_scale + bias : acc[mi][ni][r] + bias);
            }
    }
    __syncthreads();
    if (SPLITK && S == 1 && o_res) load_res();
    if (SPLITK && S > 1) {
        // raw partial tile -> scratch [tile][k part][BM * BN]; ticket.  Everyone but the last arriver is done.
        // Coherence: the parts of a tile may run on different XCDs, whose L2s are not coherent with each other.  The partials are written
        // and read with agent-scope relaxed atomic accesses (sc1: written through / read around the non-coherent caches), the writers wait
        // for their stores to be acknowledged (vmcnt counts stores on gfx9) before the barrier that precedes the ticket, and the ticket is
        // an agent-scope atomic: when the last arriver sees S - 1, every part is in memory.
        // SHIPPED FORM (SMAP_SPLITK_ACQREL=1, the default): the ticket is a RELEASE read-modify-write of every writer and the last
        // arriver passes an agent-scope ACQUIRE fence before it reads the parts -- the hand-off the HIP / LLVM memory model asks for
        // (release sequence over the ticket's RMWs; the workgroup barriers on either side carry it to the other threads).  It costs a
        // write-back of the XCD's L2 per workgroup: a batch-1 forward 2.89 -> 2.98 ms (EXPERIMENTS R6.5; batch-8 schedules have no split K).
        // A full __threadfence() per thread was 3.7 -> 10+ ms (profiles/r5_v2_ab_b1.log).
        // -DSMAP_SPLITK_ACQREL=0 is the fence-free form rounds 5 shipped: OUTSIDE the memory model (relaxed accesses only), resting on
        // gfx942 / gfx950 facts -- an agent-scope atomic store is an sc1 write-THROUGH to memory, an agent-scope atomic load an sc1 read
        // that misses the XCD's L2, vmcnt counts a store until memory has acknowledged it -- and on hipcc lowering relaxed agent-scope
        // atomics to exactly those accesses; it compiles for those two architectures only.  Either form is covered by
        // tests/test_backbone_gpu.py::test_split_k_batch_1_full_size_many_runs_bit_for_bit (two streams, 40 runs, bit for bit).
#ifndef SMAP_SPLITK_ACQREL
#define SMAP_SPLITK_ACQREL 1
#endif
#if !SMAP_SPLITK_ACQREL && !defined(__gfx942__) && !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "split K: the fence-free partial-tile hand-off is verified for gfx942 / gfx950 only; build with -DSMAP_SPLITK_ACQREL=1 elsewhere"
#endif
        float* part = a.kpart + (size_t)tile_id * S * (BM * BN);
        for (int i = tid; i < BM * BN; i += NT) __hip_atomic_store(part + (size_t)ks * (BM * BN) + i, Cs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        volatile int* s_last = reinterpret_cast<volatile int*>(smem + LDS_BYTES - 16);     // behind the epilogue tile
        if (tid == 0) {
#if SMAP_SPLITK_ACQREL
            const unsigned t = __hip_atomic_fetch_add(a.kcount + tile_id, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (unsigned)(S - 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
            const unsigned t = atomicAdd(a.kcount + tile_id, 1u);
#endif
            *s_last = t == (unsigned)(S - 1);
            if (t == (unsigned)(S - 1)) atomicExch(a.kcount + tile_id, 0u);      // ready for the next launch
        }
        __syncthreads();
        if (!*s_last) return;
        if (o_res) load_res();
        for (int i = tid; i < BM * BN; i += NT) {                // fixed order 0 .. S-1 whoever arrives last (own part re-read too)
            float v = __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 1; q < S; ++q) v += __hip_atomic_load(part + (size_t)q * (BM * BN) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float bias = a.bias[n0 + i % BN];
            Cs[i] = X3 ? v * o_scale + bias : v + bias;
        }
        __syncthreads();
    }

    if constexpr (TAPDOT) {
        // ---- tap-dot epilogue ON THE MATRIX CORES: T[128 pixels][16 taps] = act(Y)[128][256] . tapw^T, one v_mfma_f32_16x16x32_f16 tile of 16
        //      pixels per wave (8 waves = the 128 rows of the tile), K = 256 channels in 8 steps; act(Y) is split into fp16 hi | lo in
        //      registers (three MFMAs per step, like every GEMM of the split-precision path), the tap weights arrive pre-split and in fragment
        //      order.  Lane l: A = 8 channels 32 ks + 8 (l / 16) .. of pixel row l % 16 (one 32-byte LDS read; the row stride is padded by 8
        //      floats, so the 16 rows of a lane group fall on different banks); D[i] = pixel 4 (l / 16) + i, tap l % 16.
        static_assert(BM == 128 && BN == 256 && NW == 8, "one 16-row MFMA tile per wave");
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int prow = wave * 16 + (lane & 15), kq = lane >> 4;
        const half8* __restrict__ wfrag = reinterpret_cast<const half8*>(a.tap_w) + lane;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < BN / 32; ++ks) {
            const half8 wh = wfrag[(ks * 2 + 0) * 64], wl = wfrag[(ks * 2 + 1) * 64];
            const float4 y0 = *reinterpret_cast<const float4*>(Cs + prow * CSS + ks * 32 + kq * 8);
            const float4 y1 = *reinterpret_cast<const float4*>(Cs + prow * CSS + ks * 32 + kq * 8 + 4);
            float y[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
            half8 yh, yl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = y[e];
                if (o_relu) v = v < 0.f ? 0.f : v;              // NaN stays NaN (torch's ReLU)
                yh[e] = (_Float16)v;
                yl[e] = (_Float16)(v - (float)yh[e]);
            }
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl, wh, d, 0, 0, 0);     // small cross terms first, then hi * hi
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh, wh, d, 0, 0, 0);
        }
        float* const tout = reinterpret_cast<float*>(o_out);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wave * 16 + kq * 4 + i;
            if (m < a.M) tout[(size_t)m * 16 + (lane & 15)] = d[i] * a.tap_scale;        // (taps 9..15: zero weights -> zeros)
        }
        SMAP_TL_END(a)
        return;
    }

    // ---- epilogue 2: 8 consecutive channels of one pixel per thread.  Software-pipelined over the
    //      passes: the global loads of pass p+1 (bilinear taps, post-ReLU addends) are issued before the
    //      arithmetic + store of pass p, so one memory round trip is exposed per tile, not per pass.
    struct Extra { half8 t00[NPL], t01[NPL], t10[NPL], t11[NPL], a1[NPL], a2[NPL]; float ly0, ly1, lx0, lx1; bool ok; unsigned o; int row, cg; };
    constexpr int TS = NPL;                                        // pixel stride multiplier of the dense split tensors
    auto val = [](const half8 (&h)[NPL], int e) -> float {       // hi (+ lo) -> fp32, exact
        return X3 ? (float)h[0][e] + (float)h[NPL - 1][e] : (float)h[0][e];
    };
    auto load_pass = [&](int p) -> Extra {
        Extra x;
        const int idx = p * NT + tid;
        x.row = idx / CG;
        x.cg = idx - x.row * CG;
        const int m = m0 + x.row, n = nb + x.cg * 8;
        x.ok = m < a.M && n < o_cout8 && !((SMAP_ABLATE & 4) && a.M != 7);
        const int ms = x.ok ? m : 0, ns = x.ok ? n : 0;            // clamped: loads stay in bounds
        // 32-bit element offsets from the (uniform, 64-bit) tensor bases: a tensor has < 2^31 elements (plan.hip::validate)
        const unsigned dense = (unsigned)ms * (unsigned)(TS * o_cout8) + (unsigned)ns;     // res/add tensors are dense [M][Cout8] (x planes)
        x.o = (unsigned)ms * (unsigned)o_stride + (unsigned)(o_c_off + ns);
        if (FULL && o_up) {
            const int b = (SMAP_ABLATE & 64) ? 0 : ms / HoWo, rem = ms - b * HoWo;
            const int oy = (SMAP_ABLATE & 64) ? (ms & 63) : rem / a.Wo, ox = (SMAP_ABLATE & 64) ? (ms & 127) : rem - oy * a.Wo;
            Lerp ly = lerp_index(oy, a.up_h, a.Ho), lx = lerp_index(ox, a.up_w, a.Wo);
            if (SMAP_ABLATE & 64) { ly.i0 = oy >> 1; ly.i1 = ly.i0; ly.l0 = 0.5f; ly.l1 = 0.5f; lx.i0 = ox >> 1; lx.i1 = lx.i0; lx.l0 = 0.5f; lx.l1 = 0.5f; }
            const int us = TS * o_cout8;
            const unsigned tb0 = (unsigned)b * (unsigned)(a.up_h * a.up_w) * (unsigned)us + (unsigned)ns;
            const _Float16* __restrict__ tb = o_up;
#pragma unroll
            for (int pl = 0; pl < ((SMAP_ABLATE & 32) ? 0 : NPL); ++pl) {
                x.t00[pl] = *reinterpret_cast<const half8*>(tb + (tb0 + (unsigned)(ly.i0 * a.up_w + lx.i0) * (unsigned)us + (unsigned)(pl * o_cout8)));
                x.t01[pl] = *reinterpret_cast<const half8*>(tb + (tb0 + (unsigned)(ly.i0 * a.up_w + lx.i1) * (unsigned)us + (unsigned)(pl * o_cout8)));
                x.t10[pl] = *reinterpret_cast<const half8*>(tb + (tb0 + (unsigned)(ly.i1 * a.up_w + lx.i0) * (unsigned)us + (unsigned)(pl * o_cout8)));
                x.t11[pl] = *reinterpret_cast<const half8*>(tb + (tb0 + (unsigned)(ly.i1 * a.up_w + lx.i1) * (unsigned)us + (unsigned)(pl * o_cout8)));
            }
            x.ly0 = ly.l0; x.ly1 = ly.l1; x.lx0 = lx.l0; x.lx1 = lx.l1;
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            if (FULL && o_add1) x.a1[pl] = *reinterpret_cast<const half8*>(o_add1 + dense + pl * o_cout8);
            if (FULL && o_add2) x.a2[pl] = *reinterpret_cast<const half8*>(o_add2 + dense + pl * o_cout8);
        }
        return x;
    };
    auto finish_pass = [&](int p, const Extra& x) {
        float v[8];
        {
            const float4 lo = *reinterpret_cast<const float4*>(Cs + x.row * BN + x.cg * 8);
            const float4 hi = *reinterpret_cast<const float4*>(Cs + x.row * BN + x.cg * 8 + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
            v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        if (o_res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += val(rres[p], e);
        }
        if (FULL && o_up) {        // Upsample_unit: out = relu(u_skip(x) + up_conv(bilinear_up(prev))) (smap.py:213-217); the
                           // 1x1 up_conv was applied at low resolution, this is its bilinear resampling
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] += x.ly0 * (x.lx0 * val(x.t00, e) + x.lx1 * val(x.t01, e)) +
                        x.ly1 * (x.lx0 * val(x.t10, e) + x.lx1 * val(x.t11, e));
        }
        if (o_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] < 0.f ? 0.f : v[e];       // NaN stays NaN, as torch's ReLU: an overflow must reach the head-sum guard
        }
        if (FULL && o_add1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += val(x.a1, e);
        }
        if (FULL && o_add2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += val(x.a2, e);
        }
        if (!x.ok) return;
        if (a.out_fp32) {
            float* op = reinterpret_cast<float*>(o_out) + x.o;
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(o_out) + x.o) = h;
            if (X3) {           // lo plane: the part of v that fp16 dropped (v - hi is exact in fp32)
                half8 l;
#pragma unroll
                for (int e = 0; e < 8; ++e) l[e] = (_Float16)(v[e] - (float)h[e]);
                *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(o_out) + x.o + o_lo) = l;
            }
        }
    };
    TR(4);
#ifndef SMAP_EPI_DEPTH
#define SMAP_EPI_DEPTH 1          // passes whose loads are in flight ahead of the one being finished (FULL epilogue: 2)
#endif
    constexpr int ED = (FULL && PASSES > 2) ? SMAP_EPI_DEPTH : 1;
    Extra ex[ED + 1];
#pragma unroll
    for (int p = 0; p < ED && p < PASSES; ++p) ex[p] = load_pass(p);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (p + ED < PASSES) ex[(p + ED) % (ED + 1)] = load_pass(p + ED);
        finish_pass(p, ex[p % (ED + 1)]);
    }
#ifdef SMAP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR(5);
    if (a.dbg && tid == 0) {
        long long* d = a.dbg + (long long)blockIdx.x * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        d[0] = tr_t[0]; d[1] = tr_t[1]; d[2] = tr_t[2]; d[3] = tr_t[3]; d[4] = tr_t[4]; d[5] = tr_t[5]; d[6] = tr_wait; d[7] = hwid;
    }
#endif
    SMAP_TL_END(a)
}

// the split-K instances (tile ids smap_conv_tile_has_splitk: 2, 20, 22 -- what the small schedules run)
template <int BM, int BN, int WM, int WN, int STAGES, int BK, bool X3>
hipError_t launch_splitk(const ConvArgs& a, hipStream_t st)
{
    const dim3 grid(a.m_tiles * a.n_tiles * a.ksplit);
    if (a.up || a.add1 || a.add2)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, true, X3, true>), grid, dim3(WM * WN * 64), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, X3, true>), grid, dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int STAGES, int BK = 64>
hipError_t launch(const ConvArgs& a, hipStream_t st)
{
    if (a.x3) return hipErrorInvalidValue;                 // split-precision ops use the tiles of launch_x3 only
    if (a.up || a.add1 || a.add2)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, true, false>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, false>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

// split-precision (X3) instances: the same tile ids select them when the op says precision = 1
template <int BM, int BN, int WM, int WN, int STAGES, int BK>
hipError_t launch_x3(const ConvArgs& a, hipStream_t st)
{
    if (a.up || a.add1 || a.add2)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, true, true>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, true>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

}  // namespace

// the register-epilogue instance (tile 56): split precision, plain epilogue only (plan.hip::validate keeps everything else away)
template <int BM, int BN, int WM, int WN, int STAGES, int BK>
hipError_t launch_regepi(const ConvArgs& a, hipStream_t st)
{
    if (!a.x3 || a.up || a.add1 || a.add2 || a.out_fp32 || a.ksplit > 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, true, false, true>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

// the second-input instances (smap_op.in2_C > 0; tiles 20, 50, 51: plan.hip::validate keeps the rest away)
template <int BM, int BN, int WM, int WN, int STAGES, int BK, bool X3>
hipError_t launch_dual(const ConvArgs& a, hipStream_t st)
{
    if (a.up || a.add1 || a.add2 || a.ksplit > 1 || a.ksize != 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, X3, false, false, true>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int STAGES, int BK, bool X3>
hipError_t launch_relusum(const ConvArgs& a, hipStream_t st)
{
    if (a.up || a.add1 || a.add2 || a.res || a.ksplit > 1 || a.ksize != 1 || a.relu || !a.bias_b) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, STAGES, BK, false, X3, false, false, true, false, true>), dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, st, a);
    return hipGetLastError();
}

int smap_conv_tile_has_dual(int tile) { return tile == 20 || tile == 50 || tile == 51 || tile == 53 || tile == 54; }
int smap_conv_tile_has_relusum(int tile) { return tile == 50 || tile == 51 || tile == 53 || tile == 54; }

// the tap-dot instance (smap_op.tap_n = 9; tile 54 only: one N tile of 256 channels)
template <bool X3>
hipError_t launch_tapdot(const ConvArgs& a, hipStream_t st)
{
    if (a.up || a.add1 || a.add2 || a.res || a.ksplit > 1 || a.Cin2 > 0 || a.n_tiles != 1 || a.tap_n != 9 || !a.tap_w) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_igemm_kernel<128, 256, 2, 4, 2, 32, false, X3, false, false, false, true>), dim3(a.m_tiles), dim3(512), 0, st, a);
    return hipGetLastError();
}

// halves per staged K tile (= the packing unit of the weight blob, include/smap_hip.h); mirrors the BK template arguments
// of smap_launch_conv below and smap_amd/engine.py::tile_bk (tests/test_host_cpu.py compares the two)
extern "C" int smap_conv_tile_bk(int tile, int precision)
{
    int bm, bn;
    if (smap_conv_tile_dims(tile, &bm, &bn)) return 0;
    if ((tile >= 30 && tile < 50) || (tile >= 80 && tile < 100)) return precision ? 32 : 64;
    if (tile >= 60 && tile < 80) return 32;
    if (precision) return (tile <= 4 || tile == 7 || tile == 52) ? 64 : 32;
    return ((tile >= 20 && tile <= 27) || tile == 50 || tile == 51 || tile == 53 || tile == 54 || tile == 55) ? 32 : 64;
}

// tiles 80..89 (3x3 + fused 1x1 tail): channels per chunk of the TAIL's output (its weights are padded to a multiple), else 0
extern "C" int smap_conv_tile_tail_bn(int tile)
{
    int bm, bn, bn2;
    if (tile >= 90 && tile < 100) return smap_convb_tile_dims(tile, &bm, &bn, &bn2) ? 0 : bn2;
    return (tile >= 80 && tile < 90 && !smap_convf_tile_dims(tile, &bm, &bn, &bn2)) ? bn2 : 0;
}

// tile selector -> (BM, BN).  Keep in sync with smap_amd/engine.py::TILES.
//   0..4 : 2-stage (double-buffered) variants; 5..9 : the same tiles with deeper LDS-DMA pipelines
extern "C" int smap_conv_tile_dims(int tile, int* bm, int* bn)
{
    if (tile >= 30 && tile < 50) return smap_conv3_tile_dims(tile, bm, bn);
    if (tile >= 60 && tile < 80) return smap_convp_tile_dims(tile, bm, bn);
    if (tile >= 80 && tile < 90) { int bn2; return smap_convf_tile_dims(tile, bm, bn, &bn2); }
    if (tile >= 90 && tile < 100) { int bn2; return smap_convb_tile_dims(tile, bm, bn, &bn2); }
    switch (tile) {
        case 20: case 24: *bm = 128; *bn = 128; return 0;      // 20..27: BK = 32 staging (smaller LDS, more workgroups per CU)
        case 21: case 25: *bm = 128; *bn = 64; return 0;
        case 22: case 26: *bm = 64; *bn = 64; return 0;
        case 23: case 27: *bm = 64; *bn = 128; return 0;
        case 55: *bm = 128; *bn = 128; return 0;                    // 55: deep pipeline (3 K tiles in flight; 56, 57 retired: no table entry)
        case 50: case 51: case 52: *bm = 128; *bn = 128; return 0;   // 50..54: eight-wave workgroups
        case 53: *bm = 256; *bn = 128; return 0;
        case 54: *bm = 128; *bn = 256; return 0;
        case 56: *bm = 256; *bn = 256; return 0;                    // 56: eight waves of 128 x 64, register epilogue (split precision only)
        case 0: case 5: *bm = 128; *bn = 128; return 0;
        case 1: case 6: *bm = 128; *bn = 64; return 0;
        case 2: case 7: *bm = 64; *bn = 64; return 0;
        case 3: case 8: *bm = 128; *bn = 32; return 0;
        case 4: case 9: *bm = 64; *bn = 128; return 0;
        default: return -1;
    }
}

// tiles that have a split-K instance (both precisions)
int smap_conv_tile_has_splitk(int tile) { return tile == 2 || tile == 7 || tile == 20 || tile == 22; }

// tiles that have a split-precision instance (plan.hip::validate asks)
int smap_conv_tile_has_x3(int tile)
{
    return (tile >= 0 && tile <= 4) || tile == 7 || (tile >= 20 && tile <= 27) || (tile >= 30 && tile <= 45) || (tile >= 50 && tile <= 56) || (tile >= 60 && tile <= 65) ||
           (tile >= 80 && tile <= 82) || (tile >= 90 && tile <= 94);
}

hipError_t smap_launch_conv(const ConvArgs& a, int tile, hipStream_t st)
{
    if (tile >= 60 && tile < 80) return smap_launch_convp(a, tile, st);      // persistent wave-specialised kernel, both precisions
    if (tile >= 80 && tile < 90) return smap_launch_convf(a, tile, st);      // 3x3 + fused 1x1 tail, both precisions
    if (tile >= 90 && tile < 100) return smap_launch_convb(a, tile, st);     // whole identity Bottleneck, split precision
    if (a.tap_n > 0) return tile == 54 ? (a.x3 ? launch_tapdot<true>(a, st) : launch_tapdot<false>(a, st)) : hipErrorInvalidValue;
    if (a.Cin2 > 0 && a.bias_b) {                           // ... as the sum of two ACTIVATED convs (smap_op.in2_mode = 1): the eight-wave tiles only
        switch (tile) {                                     // (the four-wave 128 x 128 instance needs 288 registers: one wave per SIMD)
            case 50: return a.x3 ? launch_relusum<128, 128, 2, 4, 2, 32, true>(a, st) : launch_relusum<128, 128, 2, 4, 2, 32, false>(a, st);
            case 51: return a.x3 ? launch_relusum<128, 128, 4, 2, 2, 32, true>(a, st) : launch_relusum<128, 128, 4, 2, 2, 32, false>(a, st);
            case 53: return a.x3 ? launch_relusum<256, 128, 4, 2, 2, 32, true>(a, st) : launch_relusum<256, 128, 4, 2, 2, 32, false>(a, st);     // 249 / 251 registers:
            case 54: return a.x3 ? launch_relusum<128, 256, 2, 4, 2, 32, true>(a, st) : launch_relusum<128, 256, 2, 4, 2, 32, false>(a, st);     // two waves per SIMD still
            default: return hipErrorInvalidValue;
        }
    }
    if (a.Cin2 > 0) {                                       // second input along K: its own instances of three tiles (smap_conv_tile_has_dual)
        switch (tile) {
            case 20: return a.x3 ? launch_dual<128, 128, 2, 2, 2, 32, true>(a, st) : launch_dual<128, 128, 2, 2, 2, 32, false>(a, st);
            case 50: return a.x3 ? launch_dual<128, 128, 2, 4, 2, 32, true>(a, st) : launch_dual<128, 128, 2, 4, 2, 32, false>(a, st);
            case 51: return a.x3 ? launch_dual<128, 128, 4, 2, 2, 32, true>(a, st) : launch_dual<128, 128, 4, 2, 2, 32, false>(a, st);
            case 53: return a.x3 ? launch_dual<256, 128, 4, 2, 2, 32, true>(a, st) : launch_dual<256, 128, 4, 2, 2, 32, false>(a, st);
            case 54: return a.x3 ? launch_dual<128, 256, 2, 4, 2, 32, true>(a, st) : launch_dual<128, 256, 2, 4, 2, 32, false>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.ksplit > 1) {                                     // split K: its own instances of three tiles (plan.hip::validate asked smap_conv_tile_has_splitk)
        switch (tile) {
            case 2: return a.x3 ? launch_splitk<64, 64, 2, 2, 2, 64, true>(a, st) : launch_splitk<64, 64, 2, 2, 2, 64, false>(a, st);
            case 22: return a.x3 ? launch_splitk<64, 64, 2, 2, 2, 32, true>(a, st) : launch_splitk<64, 64, 2, 2, 2, 32, false>(a, st);
            case 7: return a.x3 ? launch_splitk<64, 64, 2, 2, 4, 64, true>(a, st) : launch_splitk<64, 64, 2, 2, 4, 64, false>(a, st);
            case 20: return a.x3 ? launch_splitk<128, 128, 2, 2, 2, 32, true>(a, st) : launch_splitk<128, 128, 2, 2, 2, 32, false>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.x3) {
        if (tile >= 30 && tile < 50) return smap_launch_conv3(a, tile, st);     // halo-tiled 3x3, split-precision instance
        switch (tile) {                                     // LDS = max(STAGES * 2 * (BM + BN) * row bytes, fp32 epilogue tile)
            case 0: return launch_x3<128, 128, 2, 2, 2, 64>(a, st);   // 128 KiB: BK = 64, half the barriers per K
            case 1: return launch_x3<128, 64, 2, 2, 2, 64>(a, st);    // 96 KiB
            case 2: return launch_x3<64, 64, 2, 2, 2, 64>(a, st);     // 64 KiB
            case 4: return launch_x3<64, 128, 2, 2, 2, 64>(a, st);    // 96 KiB
            case 3: return launch_x3<128, 32, 4, 1, 2, 64>(a, st);    // 80 KiB (Cout <= 32 heads)
            case 20: return launch_x3<128, 128, 2, 2, 2, 32>(a, st);  // 64 KiB
            case 21: return launch_x3<128, 64, 2, 2, 2, 32>(a, st);   // 48 KiB
            case 22: return launch_x3<64, 64, 2, 2, 2, 32>(a, st);    // 32 KiB
            case 23: return launch_x3<64, 128, 2, 2, 2, 32>(a, st);   // 48 KiB
            case 24: return launch_x3<128, 128, 2, 2, 3, 32>(a, st);  // 96 KiB, 2 tiles in flight (fp16 id 24 is 4-stage)
            case 25: return launch_x3<128, 64, 2, 2, 3, 32>(a, st);   // 72 KiB, 2 tiles in flight
            case 26: return launch_x3<64, 64, 2, 2, 4, 32>(a, st);    // 64 KiB, 3 tiles in flight
            case 7: return launch_x3<64, 64, 2, 2, 4, 64>(a, st);     // 128 KiB: THREE 64-half K tiles in flight (round 5: the small grids of batch-1
                                                                      // schedules leave the LDS of a CU to one workgroup anyway; their K loops are bound by
                                                                      // the latency of the one tile a two-stage pipeline keeps in flight)
            case 27: return launch_x3<64, 128, 2, 2, 3, 32>(a, st);   // 72 KiB
            case 50: return launch_x3<128, 128, 2, 4, 2, 32>(a, st);  // 64 KiB, 8 waves of 64x32
            case 51: return launch_x3<128, 128, 4, 2, 2, 32>(a, st);  // 64 KiB, 8 waves of 32x64
            case 52: return launch_x3<128, 128, 2, 4, 2, 64>(a, st);  // 128 KiB, BK = 64
            case 53: return launch_x3<256, 128, 4, 2, 2, 32>(a, st);  // 96 KiB, 8 waves of 64x64
            case 54: return launch_x3<128, 256, 2, 4, 2, 32>(a, st);  // 96 KiB, 8 waves of 64x64
            case 55: return launch_x3<128, 128, 2, 4, 4, 32>(a, st);  // 128 KiB: 8 waves, 3 K tiles (96 KiB) in flight
            case 56: return launch_regepi<256, 256, 2, 4, 2, 32>(a, st);   // 128 KiB: 8 waves of 128 x 64, two 64 KiB stages, register epilogue
            default: return hipErrorInvalidValue;
        }
    }
    if (tile >= 30 && tile < 50) return smap_launch_conv3(a, tile, st);
    switch (tile) {
        case 20: return launch<128, 128, 2, 2, 2, 32>(a, st);   // 64 KiB (fp32 epilogue tile)
        case 21: return launch<128, 64, 2, 2, 2, 32>(a, st);    // 32 KiB
        case 22: return launch<64, 64, 2, 2, 2, 32>(a, st);     // 16 KiB
        case 23: return launch<64, 128, 2, 2, 2, 32>(a, st);    // 32 KiB
        case 24: return launch<128, 128, 2, 2, 4, 32>(a, st);   // 64 KiB, 3 tiles in flight
        case 25: return launch<128, 64, 2, 2, 3, 32>(a, st);    // 36 KiB
        case 26: return launch<64, 64, 2, 2, 4, 32>(a, st);     // 32 KiB
        case 27: return launch<64, 128, 2, 2, 3, 32>(a, st);    // 36 KiB
        case 0: return launch<128, 128, 2, 2, 2>(a, st);
        case 1: return launch<128, 64, 2, 2, 2>(a, st);
        case 2: return launch<64, 64, 2, 2, 2>(a, st);
        case 3: return launch<128, 32, 4, 1, 2>(a, st);
        case 4: return launch<64, 128, 2, 2, 2>(a, st);
        case 5: return launch<128, 128, 2, 2, 4>(a, st);    // 128 KiB LDS, 1 block/CU, 3 tiles in flight
        case 6: return launch<128, 64, 2, 2, 3>(a, st);     //  72 KiB, 2 blocks/CU
        case 7: return launch<64, 64, 2, 2, 4>(a, st);      //  64 KiB, 2 blocks/CU
        case 8: return launch<128, 32, 4, 1, 3>(a, st);     //  60 KiB, 2 blocks/CU
        case 9: return launch<64, 128, 2, 2, 3>(a, st);     //  72 KiB, 2 blocks/CU
        case 50: return launch<128, 128, 2, 4, 2, 32>(a, st);   // eight-wave workgroups (fp16: 64 KiB = the fp32 epilogue tile)
        case 51: return launch<128, 128, 4, 2, 2, 32>(a, st);
        case 52: return launch<128, 128, 2, 4, 2, 64>(a, st);
        case 53: return launch<256, 128, 4, 2, 2, 32>(a, st);   // 128 KiB (fp32 epilogue tile)
        case 54: return launch<128, 256, 2, 4, 2, 32>(a, st);
        case 55: return launch<128, 128, 2, 4, 4, 32>(a, st);   // 64 KiB: 8 waves, 3 K tiles in flight
        default: return hipErrorInvalidValue;
    }
}
