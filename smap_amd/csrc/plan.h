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
    int H, W, Cin, in_stride_c, in_c_off;
    int Ho, Wo, Cout8;       // Cout rounded up to 8 (channels actually written)
    int ksize, stride, pad, relu;
    int out_stride_c, out_c_off, out_fp32;
    int M, K, m_tiles, n_tiles;
};

int smap_conv_tile_dims(int tile, int* bm, int* bn);
hipError_t smap_launch_conv(const ConvArgs& a, int tile, hipStream_t st);
