// plan.h -- internal declarations shared by conv.hip and plan.hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvArgs {
    const void* arena;       // activation arena base; its first 256 bytes are zeros (padding taps read them)
    long long in_off;        // byte offset of the NHWC fp16 input (channel stride in_stride_c, first channel in_c_off)
    const _Float16* w;       // [cout_pad][K] fp16
    const float* bias;       // [cout_pad] fp32
    void* out;               // fp16 or fp32 NHWC, channel stride out_stride_c, offset out_c_off
    const _Float16* res;     // dense [M][Cout8] added before ReLU, or null
    const _Float16* add1;    // dense [M][Cout8] added after ReLU, or null
    const _Float16* add2;
    const _Float16* up;      // optional low-res [B,up_h,up_w,Cout8] tensor: bilinear(align_corners) added before ReLU
    int up_h, up_w;
    int H, W, Cin, in_stride_c, in_c_off;
    int Ho, Wo, Cout8;       // Cout rounded up to 8 (channels actually written)
    int ksize, stride, pad, relu;
    int out_stride_c, out_c_off, out_fp32;
    int M, K, m_tiles, n_tiles;
    // split-precision mode (smap_op.precision = 1, conv.hip X3): every fp16 tensor is [pixel][hi(C) | lo(C)]
    int x3;                  // 0 = fp16 storage, 1 = hi/lo split storage, three MFMAs per K step
    int in_lo;               // halves from a pixel's hi channel c to its lo channel c (input tensor)
    int out_lo;              // same for the output tensor (fp16 outputs only)
    long long w_lo;          // bytes from the hi weight matrix to the lo weight matrix
    float acc_scale;         // accumulator multiplier undoing the power-of-two weight pre-scale
    int w_pairs;             // 32-half K tiles of the packed weights come in pairs (128-byte rows [tile 2p | tile 2p+1])
    // fused 1x1 tail (convf.hip, tile ids 80..89): out / res / add1 / add2 / relu then belong to the TAIL's output
    const _Float16* w2;      // packed [n2 chunk][k chunk][BN2 rows][128 B] weights of the 1x1, or null
    const float* bias2;      // [tail cout_pad] fp32
    int tail_cout8;          // channels the tail writes (multiple of 8)
    int tail_chunks;         // tail cout_pad / BN2
    float tail_acc_scale;
    // fused leading 1x1 (convb.hip, tile ids 90..99): the op's INPUT tensor then has head_cin channels, w / bias are the 3x3's
    const _Float16* w0;      // packed [k chunk][P rows][128 B] weights of the leading 1x1, or null
    const float* bias0;      // [P] fp32
    int head_cin;            // channels of the block's input (= of its output)
    float acc_scale0;
    const _Float16* wd;      // tile ids 92, 93: packed [n chunk][k chunk][64 rows][128 B] weights of the 1x1 SHORTCUT conv of a layer's first
    float acc_scale_d;       // block (head_cin -> tail_cout, no ReLU; its bias is folded into bias2), or null
    // split K (smap_op.ksplit, conv.hip only): K parts per output tile, scratch for the raw partial tiles, one ticket per tile
    int ksplit;
    float* kpart;
    unsigned* kcount;
    // second input, concatenated along K (smap_op.in2_*, conv.hip template DUAL): K = Cin + Cin2, 1x1; x2 sampled with spatial stride stride2
    long long in2_off;       // byte offset from the launch's base (the first input's window)
    int H2, W2, Cin2, in2_stride_c, stride2, in2_lo;
    // in2_mode 1 (conv.hip template RELUSUM): relu(W1 x + b1) + relu(W2 x2 + b2); bias / acc_scale are W1's, these W2's
    const float* bias_b;
    float acc_scale_b;
    // tap-dot epilogue (smap_op.tap_n, conv.hip template TAPDOT): instead of storing its [M][N] activation the launch stores, per pixel, the
    // tap_n dot products of that activation with tap_n weight vectors -- the per-pixel half of a following 3x3 conv with ONE output channel
    const _Float16* tap_w;   // [8 K steps][hi, lo][64 lanes][8 halves]: B fragments of v_mfma_f32_16x16x32_f16, or null
    int tap_n;
    float tap_scale;
    // N segments (smap_op.seg_*): rows >= seg_n1 / seg_n2 of the weight matrix belong to outputs 1 / 2 (INT_MAX = no such segment)
    int seg_n1, seg_n2;
    void* seg_out1; void* seg_out2;
    int seg_cout8_1, seg_cout8_2, seg_stride1, seg_stride2, seg_relu1, seg_relu2;
    float seg_scale1, seg_scale2;
#ifdef SMAP_TRACE
    long long* dbg;          // diagnostics build only (tools/build_ablate.py --trace): per-workgroup phase stamps
#endif
#ifdef SMAP_TIMELINE
    long long* tl;           // diagnostics build only (tools/build_ablate.py --timeline): [1 + blocks][4] int64 slice of this launch:
    long long tl_meta[4];    //   row 0 = tl_meta (op index, grid, stream, tile), row 1+b = start, end (s_memrealtime), HW_ID, XCC_ID
#endif
};

// ATen's index/weight rule for bilinear align_corners=True (UpSample.h compute_source_index_and_lambda):
// identity when sizes match; else src = dst*(in-1)/(out-1) in fp32, i0 = (int)src, i1 = i0 + (i0 < in-1),
// l1 = clamp(src - i0, 0, 1), l0 = 1 - l1.
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_index(int dst, int in_size, int out_size)
{
    Lerp r;
    if (in_size == out_size) { r.i0 = dst; r.i1 = dst; r.l0 = 1.f; r.l1 = 0.f; return r; }
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float src = scale * (float)dst;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)r.i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}

int smap_conv_tile_has_splitk(int tile);                                    // conv.hip: tile ids with a split-K instance
int smap_conv_tile_has_relusum(int tile);                                   // ... with a relu(conv) + relu(conv) instance (smap_op.in2_mode = 1)
int smap_conv_tile_has_dual(int tile);                                      // conv.hip: tile ids with a second-input (K-concatenated) instance
int smap_conv_tile_has_x3(int tile);                                        // conv.hip: tile ids with a split-precision instance
hipError_t smap_launch_conv(const ConvArgs& a, int tile, hipStream_t st);
int smap_conv3_tile_dims(int tile, int* bm, int* bn);                       // conv3.hip (tile ids 30..33, halo-tiled 3x3)
hipError_t smap_launch_conv3(const ConvArgs& a, int tile, hipStream_t st);
int smap_convp_tile_dims(int tile, int* bm, int* bn);                       // convp.hip (tile ids 60..69, persistent wave-specialised GEMM)
hipError_t smap_launch_convp(const ConvArgs& a, int tile, hipStream_t st);
int smap_convf_tile_dims(int tile, int* bm, int* bn, int* bn2);             // convf.hip (tile ids 80..89, 3x3 + fused 1x1 tail)
hipError_t smap_launch_convf(const ConvArgs& a, int tile, hipStream_t st);
int smap_convb_tile_dims(int tile, int* bm, int* bn, int* bn2);             // convb.hip (tile ids 90..99, whole identity Bottleneck)
hipError_t smap_launch_convb(const ConvArgs& a, int tile, hipStream_t st);
hipError_t smap_launch_convc(const ConvArgs& a, hipStream_t st);           // convc.hip (tile id 94: the same for 128 planes / 512 channels)

#ifdef SMAP_TIMELINE
// every workgroup of a conv kernel calls these two (first / last statement): 100 MHz device-wide clock
#define SMAP_TL_BEGIN const long long tl_t0 = __builtin_amdgcn_s_memrealtime();
#define SMAP_TL_END(a)                                                                                        \
    if ((a).tl && threadIdx.x == 0) {                                                                         \
        unsigned hw_, xcc_;                                                                                   \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                     \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                   \
        long long* d_ = (a).tl + 4 * (1 + (long long)blockIdx.x);                                             \
        d_[0] = tl_t0; d_[1] = __builtin_amdgcn_s_memrealtime(); d_[2] = hw_; d_[3] = xcc_;                   \
        if (blockIdx.x == 0) { (a).tl[0] = (a).tl_meta[0]; (a).tl[1] = (a).tl_meta[1]; (a).tl[2] = (a).tl_meta[2]; (a).tl[3] = (a).tl_meta[3]; } \
    }
#else
#define SMAP_TL_BEGIN
#define SMAP_TL_END(a)
#endif
