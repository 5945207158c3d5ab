/*
 * smap_hip.h -- C ABI of libsmap_hip.so, the MI355X (gfx950) implementation of
 * the SMAP inference hot path.  Plain pointers and sizes only; no torch types.
 *
 * Every entry point is stream-ordered on the hipStream_t passed as `stream`
 * (void* so that this header needs no HIP include), never allocates, never
 * synchronises, borrows all buffers from the caller and returns
 *      0            on success,
 *      SMAP_E_ARG   (-1) on an argument error,
 *      -(1000+e)    when a HIP call failed with hipError_t e.
 * All pointers are DEVICE pointers unless the parameter says "host".
 *
 * The reference interface each entry replaces is cited as file:line relative
 * to the reference checkout (zju3dv/SMAP).
 */
#ifndef SMAP_HIP_H
#define SMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the file are its
 * whole dynamic symbol table (tests/test_abi_cpu.py compares `nm -D` with this header in both directions). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define SMAP_E_ARG (-1)

#define SMAP_NJ 15        /* association.cpp:18 nJoints  */
#define SMAP_NL 14        /* association.cpp:19 nLimbs   */
#define SMAP_MAXP 127     /* association.cpp:20 maxPeaks */
#define SMAP_HMS_C 43     /* association.cpp:21 heatmapDim[0] = 15 keypoints + 28 PAF */

/* Library / build identification (also used by the tests to prove the HIP
 * library, not a fallback, is what is loaded). */
const char* smap_version(void);

/* ---- association: replaces extensions/association.cpp::extract/connect ---- */

/* test.py:111-112  hmsIn[:15] /= 255 ; hmsIn[15:] /= 127  (in place, fp32 IEEE divide)
 * hms: [B,43,H,W] fp32. */
int smap_scale_hms(float* hms, int B, int H, int W, void* stream);

/* test.py:55-70 flip-TTA merge, in place on hms [B,43,H,W]: hms[:,i] += s_i * flip_x(hms_flip)[:,pair43[i]]
 * (s_i = -1 on PAF-x channels 15,17,..), then hms[:,15:] *= 0.5.  pair43: HOST pointer to 43 ints
 * (KEYPOINT.FLIP_ORDER followed by 15 + PAF.FLIP_CHANNEL, dataset/data_settings.py:22,33-34). */
int smap_flip_merge(float* hms, const float* hms_flip, const int* pair43, int B, int H, int W, void* stream);

/* nmsBase.cu:10-175 (nmsRegisterKernel + exclusive_scan + writeResultKernel) fused.
 * hms  : [B,C,H,W] fp32, C >= 15 (only channels 0..14 are read)
 * peaks: [B,15,128,3] fp32 out; slot 0 = (count,0,0), slots > count zero-filled.
 * Requires H*W <= 32768. */
int smap_nms(const float* hms, int B, int C, int H, int W, float threshold,
             float* peaks, void* stream);
/* The same result from TWO launches and a caller-provided workspace (the library never allocates): nmsRegisterKernel's mask
 * (nmsBase.cu:10-41) as one thread per pixel over the whole batch -- B x 15 x H x W / 256 workgroups instead of the B x 15 of the
 * fused form, whose 120 workgroups at eight frames walk 26 serial chunks each -- into a bit mask per 64 pixels, then the scan and the
 * 7x7 centroid write-out (nmsBase.cu:43-135, 166) per channel.  workspace: DEVICE memory, 8-byte aligned, at least
 * smap_nms_workspace_bytes(B, H, W) = B * 15 * ceil(H * W / 64) * 8 bytes; not read or written outside the call's launches. */
int64_t smap_nms_workspace_bytes(int B, int H, int W);
int smap_nms_ws(const float* hms, int B, int C, int H, int W, float threshold, float* peaks, void* workspace,
                int64_t workspace_bytes, void* stream);

/* bodyPartConnectorBase.cu:11-63,104-189 (process + pafScoreKernel).
 * hms: [B,43,H,W]; peaks as above; scores: [B,14,127,127] fp32 out (-1 where no pair). */
int smap_paf_score(const float* hms, const float* peaks, int B, int H, int W,
                   float* scores, void* stream);

/* association.cpp:123-233 (findConnectedJoints), batched: one frame per workgroup.
 * rdepth: [B,H,W] fp32 root-depth maps; bodys: [B,127,15,4] fp32 out (x,y,0,score in
 * heat-map pixels; rows >= counts[b] are zero); counts: [B] int32 out. */
int smap_group(const float* peaks, const float* scores, const float* rdepth, int B,
               int H, int W, int root_idx, int dist_flag, float* bodys, int32_t* counts,
               void* stream);

/* test.py:116-134 + test_util.py:45-99 + post_3d.py:4-27 (x4, nearest x4 upsample,
 * generate_relZ, chain_bones, gen_3d_pose, back_projection), batched.
 * det_d : [B,14,H,W] fp32, root_d: [B,H,W] fp32 (heat-map resolution)
 * cams  : [B,9] f64 = scale,img_w,img_h,net_w,net_h,f_x,f_y,cx,cy
 * pred_2d: [B,127,15,4] fp32 out; pred_3d: [B,127,15,4] f64 out; root_z: [B,127] f64 out. */
int smap_lift(const float* bodys, const int32_t* counts, const float* det_d,
              const float* root_d, const double* cams, int B, int H, int W,
              float* pred_2d, double* pred_3d, double* root_z, void* stream);

/* test_util.py:102-131 + refinenet.py:5-37 (lift_and_refine_3d_pose + RefineNet MLP).
 * wt: 5 device pointers (host array) to BN-folded TRANSPOSED weights [in][out] fp32,
 * bs: 5 device pointers (host array) to folded biases [out]; dims 75,160,256,256,128,45.
 * refined: [B,127,15,4] f64 out. */
int smap_refine(const float* pred_2d, const double* pred_3d, const int32_t* counts, int B,
                const float* const* wt, const float* const* bs, double* refined, void* stream);

/* ---- ground-truth modes of test.py (-t generate_result / generate_train), SURVEY.md 8f rank 4 ----
 * register_pred WITH ground truth (test_util.py:18-42), batched: greedy nearest-root matching (< 30 px,
 * ties in row-major (annotation, prediction) order) of connect's persons to the kept annotations.
 * bodys/counts: smap_group's outputs; gt_roots: [B,G,2] fp32 root joint (x,y) of the annotations whose root is
 * visible (test.py:76-80), network pixels; gt_counts: [B] int32 (<= G <= 64).
 * matched: [B,127,15,4] fp32 out -- row g = the prediction assigned to annotation g (heat-map pixels) or zeros;
 * matched_counts: [B] int32 out = gt_counts[b], or 0 when the frame has no prediction / no annotation. */
int smap_register_gt(const float* bodys, const int32_t* counts, const float* gt_roots,
                     const int32_t* gt_counts, int B, int G, float* matched, int32_t* matched_counts,
                     void* stream);

/* smap_lift for those modes: the person array is float64 there (test_util.py:37), so pred_2d is f64 and no
 * intermediate is rounded to fp32; cams carries the annotation's intrinsics (test.py:84-93). */
int smap_lift_gt(const float* bodys, const int32_t* counts, const float* det_d, const float* root_d,
                 const double* cams, int B, int H, int W, double* pred_2d, double* pred_3d, double* root_z,
                 void* stream);

/* smap_refine on the f64 pred_2d of smap_lift_gt. */
int smap_refine_gt(const double* pred_2d, const double* pred_3d, const int32_t* counts, int B,
                   const float* const* wt, const float* const* bs, double* refined, void* stream);

/* refinenet.py:34-37 RefineNet.forward alone: x [N,75] fp32 -> y [N,45] fp32 (same weight format). */
int smap_refine_mlp(const float* x, int N, const float* const* wt, const float* const* bs, float* y,
                    void* stream);

/* dataset/custom_dataset.py:27-68 (cv2.resize(img, (0,0), fx, fy) to fit, centre pad with 128, ToTensor, Normalize) for one
 * image on the device.  src: uint8 [h][w][3] BGR; the resized image is nh x nw = cvRound(h*fy) x cvRound(w*fx) and sits at
 * (top,left) of the net_h x net_w canvas; dst: fp32 [3][net_h][net_w]; mean3/std3: HOST pointers to 3 floats.  The resize is
 * OpenCV's 8-bit INTER_LINEAR restated operation by operation (11-bit fixed-point coefficients, source coordinate
 * (d + 0.5) / f - 0.5, the >>4 / >>16 / +2 >>2 vertical pass; exact 2x shrink = 2x2 box mean). */
int smap_preprocess(const unsigned char* src, int h, int w, int nh, int nw, int top, int left, float* dst,
                    int net_h, int net_w, const float* mean3, const float* std3, double fx, double fy, void* stream);

/* ---- backbone: replaces model/smap.py SMAP.forward (eval) ------------------ */

/* One op of the static inference schedule.  Offsets are BYTE offsets into the
 * caller's activation arena / weight blob; -1 = absent. */
enum smap_op_kind {
    SMAP_OP_CONV = 0,       /* conv_bn_relu (smap.py:13-45) with folded BN, implicit GEMM on MFMA; with tail_cout / head_cin a
                               Bottleneck's 3x3 + 1x1, or the whole Bottleneck (smap.py:48-77), in one launch */
    SMAP_OP_STEM = 1,       /* ResNet_top conv 7x7 s2 (smap.py:83-84) from fp32 NCHW input      */
    SMAP_OP_MAXPOOL = 2,    /* ResNet_top maxpool 3x3 s2 p1 (smap.py:86)                         */
    SMAP_OP_UPADD = 3,      /* out = relu(a + bilinear_align_corners(t)) (smap.py:213-217)       */
    SMAP_OP_HEADSUM = 4,    /* fp32 NCHW out = sum of bilinear-upsampled heads (smap.py:221-229,417-419) */
    SMAP_OP_STEMPOOL = 5,   /* ResNet_top whole (smap.py:83-86): STEM followed by MAXPOOL in one kernel; H,W = image,
                               Ho,Wo = pooled size, same weight format as STEM                                  */
    SMAP_OP_TAPSUM = 6,     /* the second half of a 3x3 conv with ONE output channel whose first half ran inside the producing launch
                               (CONV with tap_n = 9): out[b,0,y,x] = bias + sum over (dy,dx) in {-1,0,1}^2 of t[b, y+dy, x+dx][3 (dy+1) + dx+1]
                               (zero outside the map) -- fp32 NCHW map at ext_off of the output buffer, like HEADSUM (same status words).
                               aux_off[0] = t, fp32 [B,H,W,Cin] (Cin = 16: nine used); Ho,Wo = H,W; Cout = 1; bias: one fp32 at bias_off */
};

typedef struct smap_op {
    int32_t kind;
    int32_t B, H, W, Cin;           /* input geometry (CONV/MAXPOOL/UPADD: NHWC fp16; STEM: NCHW fp32) */
    int32_t in_stride_c, in_c_off;  /* CONV: channel stride / first channel of the input pixel   */
    int32_t Ho, Wo, Cout;           /* output geometry                                           */
    int32_t ksize, stride, pad;
    int32_t relu;                   /* apply ReLU after bias (+residual)                         */
    int32_t cout_pad;               /* rows of the padded weight matrix (multiple of the N tile) */
    int32_t out_stride_c;           /* channel stride of the output pixel (>= Cout)              */
    int32_t out_c_off;              /* channel offset inside the output pixel                    */
    int32_t out_fp32;               /* CONV: 1 = fp32 NHWC output (heads), 0 = fp16              */
    int32_t tile;                   /* CONV tile selector: 0=128x128 1=128x64 2=64x64 3=128x32 4=64x128 */
    int32_t n_aux;                  /* HEADSUM: number of source tensors (1..3)                  */
    int64_t in_off, out_off;        /* arena byte offsets                                        */
    int64_t w_off, bias_off;        /* weight-blob byte offsets.  CONV: fp16 weights of [cout_pad][K], K = (kh, kw, cin), stored
                                       as ONE CONTIGUOUS BLOCK PER STAGED WEIGHT TILE in the kernel's LDS order, so that
                                       every wave-wide LDS-DMA reads one contiguous KiB (strided 64-byte row segments
                                       stream from L2 at half the rate):
                                         [n tile][K tile][plane][BN rows][16-byte slots]
                                       BN = the N extent of `tile` (smap_conv_tile_dims), K tile = smap_conv_tile_bk(tile,
                                       precision) halves in K order, slot s of row r = K granule s ^ ((r>>1)&7) (64-half
                                       tiles) or s ^ ((r>>2)&3) (32-half tiles; with w_pairs = 1 those are stored in PAIRS,
                                       [n tile][pair][plane][BN rows][tile 2p (64 B) | tile 2p+1 (64 B)]: 128-byte rows); halo tiles 30..49: blocks ordered
                                       [n tile][channel chunk][tap], 128-byte rows of 64 channels (precision 1: of 32
                                       channels as granules 0..3 = hi, 4..7 = lo), slot s = granule s ^ ((r>>1)&7).
                                       Reference packer: smap_amd/engine.py::pack_conv_weights.  + fp32 bias [cout_pad].
                                       STEM: fp16 [64][176] with K = (kh, c, kw padded to 8) + 8 zero columns, + fp32 [64] */
    int64_t res_off;                /* dense [M][Cout] tensor added before ReLU, or -1           */
    int64_t add1_off, add2_off;     /* dense tensors added AFTER ReLU (smap.py:142-153), or -1   */
    int64_t aux_off[3];             /* CONV: aux[0] = optional low-res fp16 [B,aux_h,aux_w,Cout] tensor, bilinearly
                                       (align_corners) upsampled and added before ReLU (smap.py:213-217);
                                       UPADD: aux[0] = low-res t ; HEADSUM: fp32 NHWC head tensors (arena) */
    int32_t aux_h[3], aux_w[3];     /* their spatial sizes                                       */
    int64_t ext_off;                /* HEADSUM: byte offset of the [B,Cout,Ho,Wo] block in the fp32 output buffer */
    int32_t precision;              /* CONV/STEM/MAXPOOL: 0 = fp16 activations and weights (one rounding per stored value,
                                       ~1e-3 relative error through the 200-layer graph); 1 = SPLIT precision, the mode
                                       that reproduces the reference's fp32 arithmetic (smap.py runs fp32 end to end):
                                       every fp16 tensor is stored as two planes, pixel = [hi(C) | lo(C)] with
                                       hi = fp16(v), lo = fp16(v - hi) (in_stride_c / out_stride_c are then 2*C and
                                       res/add/aux tensors have pixel stride 2*Cout8); weights are two consecutive
                                       matrices hi | lo of (w * 2^s); three MFMAs per K step (hi*hi + hi*lo + lo*hi)
                                       into fp32.  fp32 head outputs (out_fp32) and HEADSUM are the same in both modes. */
    float acc_scale;                /* precision 1: 2^-s, applied to the accumulator before the bias             */
    int32_t flip_from;              /* flip-TTA inside the schedule (test.py:55-70), 0 = off.  STEM: frames b >= flip_from
                                       are computed from the x-MIRRORED image of input frame b - flip_from (the input
                                       holds flip_from frames, the schedule runs B = 2 * flip_from).  HEADSUM: B is the
                                       number of OUTPUT frames; when flip_from > 0 the summed maps of frame b + flip_from
                                       are merged into frame b: out[b,c,y,x] = v[b,c,y,x] + s_c * v[b+flip_from, pair[c], y,
                                       W-1-x], s_c = -1 on PAF-x channels (c >= in_c_off, (c - in_c_off) even), then
                                       channels >= in_c_off are halved; pair = int32[Cout] at w_off in the weight blob. */
    int32_t w_pairs;                /* CONV, 32-half K tiles only: 1 = the weight blob stores them in pairs (see w_off), 0 = one
                                       contiguous block per K tile */
    int32_t status_off;             /* HEADSUM: byte offset (> 0) in the fp32 output buffer of SMAP_STATUS_WORDS(B) int32 STATUS words, or
                                       0 = none.  smap_plan_run clears them; when a value the head sum writes for output frame b is not
                                       finite it ORs bit 0 and bit 1 + b % 31 into word b / 31, and bit 0 into word 0 (so word 0's bit 0
                                       says "some frame of the launch", and a host can drop exactly the affected frames of a launch of
                                       any size and keep the others): split precision keeps
                                       fp16's RANGE, an activation beyond 65504 turns into inf / NaN downstream; the host
                                       checks the word when it collects the maps. */
    int32_t tail_cout;              /* CONV, tile ids 80..89 only (else 0): the op is a Bottleneck TAIL in one launch
                                       (smap.py:48-77) -- a 3x3 stride-1 conv Cin -> Cout (= the tile's N extent, bias + ReLU,
                                       never stored) followed by a 1x1 conv Cout -> tail_cout whose bias / res_off / relu /
                                       add1_off / add2_off / out_* fields are this op's: the output tensor has tail_cout
                                       channels (multiple of 8).  0 = a plain conv. */
    int32_t tail_cout_pad;          /* rows of the padded 1x1 weight matrix: multiple of smap_conv_tile_tail_bn(tile) */
    float tail_acc_scale;           /* precision 1: 2^-s of the 1x1 weights */
    int64_t tail_w_off;             /* weight-blob byte offsets of the 1x1: blocks [n chunk][k chunk][BN2 rows][128 B], rows */
    int64_t tail_bias_off;          /* in the halo tiles' format (64 channels, or hi32 | lo32); fp32 bias [tail_cout_pad] */
    int32_t head_cin;               /* CONV, tile ids 90..99 only (else 0): the op is a WHOLE stride-1 identity Bottleneck in one launch
                                       (smap.py:48-77; csrc/convb.hip: Cin = 64 planes, tile ids 90..93; csrc/convc.hip: 128 planes, tile id 94;
                                       split precision): a LEADING 1x1 conv head_cin -> Cin (bias +
                                       ReLU, never stored, recomputed on the 3x3's halo) in front of the 3x3 and its tail.  The input
                                       tensor then has head_cin channels (in_stride_c = 2 * head_cin), is read once, and is also
                                       the residual: res_off must equal in_off and tail_cout = head_cin. */
    float head_acc_scale;           /* 2^-s of the leading 1x1's weights */
    int64_t head_w_off;             /* weight-blob byte offsets of the leading 1x1: blocks [k chunk][Cin rows][128 B] in the halo */
    int64_t head_bias_off;          /* tiles' row format (hi32 | lo32 of 32 input channels: tile ids 92..94; tile ids 90, 91: 16-channel stages, 64-byte rows,
                                       smap_amd/engine.py::pack_rows16); fp32 bias [Cin] */
    int64_t short_w_off;            /* tile ids 92, 93 only (present iff short_acc_scale > 0): the FIRST block of a layer (smap.py:124-129) -- head_cin = 64 input
                                       channels and, instead of "+ input", a 1x1 SHORTCUT conv head_cin -> tail_cout (folded BN, no ReLU) on
                                       the same input, added before the final ReLU.  Weight blocks [n chunk][k chunk][64 rows][128 B] like
                                       the tail's; its bias is folded into the tail's bias by the packer; res_off = -1 */
    float short_acc_scale;          /* 2^-s of the shortcut conv's weights; 0 = the op has no shortcut conv */
    int32_t scale_hms;              /* HEADSUM: 1 = the maps are written ALREADY SCALED as exps/stage3_root2/test.py:111-112 scales them before the
                                       association: channels < in_c_off (the key points) / 255, the others (the PAFs) / 127 -- the same fp32
                                       divisions smap_scale_hms performs, applied to the summed (and, with flip_from, merged) value in the
                                       head sum's store, so that no separate pass over the maps runs between the backbone and smap_nms.
                                       0 = raw maps: what model.smap.SMAP.forward returns (smap.py:417-419).  Other kinds: 0. */
    /* N SEGMENTS: several 1x1 convs that read the SAME input as one launch with up to three outputs (Upsample_unit, smap.py:210-241:
       u_skip and skip1 both read x; skip2, cross_conv / res_conv1 and the next unit's up_conv all read `out`).  The weight
       matrix [cout_pad][K] is the concatenation of the convs' rows, every segment starting on a multiple of the tile's N extent
       (rows in between are zero); segment 0 = rows [0, seg_n[0]) is described by the op's own fields (Cout, out_off, relu,
       acc_scale, res / add / aux: those apply to segment 0 only); segment j = 1, 2 = rows [seg_n[j-1], seg_n[j] or cout_pad)
       writes seg_cout[j-1] channels (multiple of 8) to the dense fp16 tensor at seg_out_off[j-1] (channel stride
       seg_out_stride_c[j-1]) with its own ReLU flag and accumulator scale.  seg_n[0] = 0: a plain conv.  Tile ids of
       csrc/conv.hip only (0..9, 20..27, 50..55), ksize 1, fp16 outputs. */
    int32_t seg_n[2];
    int32_t seg_cout[2], seg_relu[2], seg_out_stride_c[2];
    float seg_acc_scale[2];
    int64_t seg_out_off[2];
    /* SPLIT K (tile ids 2, 7, 20, 22 of csrc/conv.hip; small schedules -- batch 1 -- whose launches have fewer output tiles than the chip has
       CUs): ksplit = S > 1 workgroups per output tile, each over 1/S of the K tiles; their raw fp32 accumulators meet in the scratch
       area at kpart_off (arena bytes: m_tiles * n_tiles * S * BM * BN * 4, BM x BN = smap_conv_tile_dims) and the last workgroup to
       arrive -- one uint32 ticket per tile at kcount_off (arena; smap_plan_run zeroes the tickets of a schedule before its first op,
       the kernel leaves them at zero) -- sums the parts in the fixed order 0 .. S-1 and runs the epilogue: deterministic, one launch.
       0 or 1 = off.  S <= 16 and S <= the number of K tiles (K / smap_conv_tile_bk). */
    int32_t ksplit, reserved1;
    int64_t kpart_off, kcount_off;
    /* LANES (smap_plan_set_lanes): independent branches of the schedule -- the head chains of an Upsample_unit (model/smap.py:219-229)
       next to the unit that follows -- may run on forked streams.  lane = 0 (the caller's stream) .. SMAP_MAX_LANES - 1; wait_op = indices
       of EARLIER ops on OTHER lanes whose results this op reads (-1 = unused; n_wait of them).  smap_plan_run makes the op's stream wait
       for those ops and joins every lane into the caller's stream at the end; inside a stream capture the lanes become parallel branches
       of the graph.  With lanes off (the default) every op runs on the caller's stream in order and the fields are ignored.  The packer
       guarantees that buffers touched by an op on a side lane are not reused before the end of the schedule. */
    int32_t lane, n_wait;
    int32_t wait_op[4];
    /* SECOND INPUT, concatenated along K (round 6; csrc/conv.hip tiles 20, 50, 51, 53, 54; ksize 1): the op computes
           out = act( W [x | x2_sampled] + bias (+ res) )
       -- ONE accumulator over K = Cin + in2_C, the weight matrix [cout_pad][Cin + in2_C] being the two convs' matrices side by side.  This is
       the last 1x1 of a stride-2 / widening Bottleneck TOGETHER with its 1x1 shortcut conv (model/smap.py:60-77, 124-129:
       out = relu(bn3(conv3(y)) + downsample(x)), no activation between the two sums): the shortcut's output is never written or read back and
       its launch disappears.  x2 = fp16 NHWC tensor [B, in2_H, in2_W] of in2_C channels (channel stride in2_stride_c, hi | lo planes in split
       precision like every tensor) at in2_off, sampled at pixel (oy * in2_stride, ox * in2_stride) for output pixel (oy, ox):
       Ho = (in2_H - 1) / in2_stride + 1, likewise Wo.  in2_C = 0: no second input.  x2 lies in the same 4 GiB window as the first input
       (one 64-bit base per launch, include/smap_hip.h "windows").  Not with split K, N segments, the fused bilinear add or a 3x3. */
    int64_t in2_off;
    int32_t in2_H, in2_W, in2_C, in2_stride_c, in2_stride;
    /* in2_mode = 1 (needs in2_C > 0, in2_stride = 1): NOT one sum over the concatenated K but the sum of two ACTIVATED convs,
           out = relu(W1 x + b1) + relu(W2 x2 + b2),
       in one launch: the K loop walks x's tiles, the accumulators are turned into relu(acc * acc_scale + bias) and parked in registers, then
       it walks x2's tiles from zero.  The inter-stage skips of an Upsample_unit, skip1(x) + skip2(out) (smap.py:218-241, 142-153: both are only
       ever ADDED to the next stage's feature map), as ONE tensor: one write and one read of an in_planes-wide tensor less per level.  The
       packed weight matrix is [W1 2^s1 | W2 2^s2] along K; bias_off / acc_scale belong to W1, in2_bias_off / in2_acc_scale to W2; the op's
       relu field is ignored.  in2_mode = 0: the K-concatenated sum described above. */
    int32_t in2_mode;
    float in2_acc_scale;
    int64_t in2_bias_off;
    /* TAP-DOT EPILOGUE (round 6; csrc/conv.hip tile 54 = 128 x 256, cout_pad = 256 = ONE N tile, ksize 1): tap_n = 9 -- the launch's
       activation y = act(W x + b) is NOT stored; its only consumer is a 3x3 conv with one output channel (smap.py:227-229, res_rd_conv2 on
       res_rd_conv1), whose per-pixel half the epilogue computes instead, on the matrix cores: t[m][k] = < tapw[k], y[m] >, k = 0..8.
       tapw at tap_w_off in the weight blob: the folded 3x3 weights [kh*3+kw][channel] * 2^s as fp16 hi | lo in the B-fragment order of
       v_mfma_f32_16x16x32_f16 -- [K step 0..7][plane hi, lo][lane 0..63][8 halves], lane l holding tap l % 16 (taps 9..15: zeros), channels
       32 step + 8 (l / 16) .. +7 -- and tap_scale = 2^-s.  `out` is then the fp32 tensor t, pixel stride out_stride_c = 16 (out_fp32 = 1,
       entries 9..15 written as zeros), and a TAPSUM op finishes the conv.  0 = off.  Not with res / adds / segments / split K / in2. */
    int32_t tap_n;
    float tap_scale;
    int64_t tap_w_off;
} smap_op;

#define SMAP_STATUS_WORDS(frames) (((frames) + 30) / 31)      /* int32 status words of a schedule with `frames` output frames */

/* sizeof(smap_op) as compiled: lets a foreign-language binding verify its struct mirror. */
int smap_sizeof_op(void);
/* Geometry of a CONV tile id, for whoever packs the weight blob: M x N extent of the output tile (0 on success, -1 for an
 * unknown id) and the halves per staged K tile for `precision` (0 for an unknown id). */
int smap_conv_tile_dims(int tile, int* bm, int* bn);
int smap_conv_tile_bk(int tile, int precision);
/* Tile ids 80..89 (3x3 + fused 1x1 tail): output channels per chunk of the tail (its weight rows are padded to a multiple); 0 for
 * every other id. */
int smap_conv_tile_tail_bn(int tile);

typedef struct smap_plan smap_plan;
#define SMAP_MAX_LANES 4

/* Copies `ops`; validates geometry.  n_ops <= 4096.  Arena contract: the conv kernels address their input with a 64-bit
 * base (the input's WINDOW: its arena offset rounded down to a multiple of 4 GiB) plus 32-bit lane offsets, and read zeros
 * for padding taps at the start of that window.  So bytes [k * 2^32, k * 2^32 + 16384) are RESERVED for every k >= 0
 * (smap_plan_run zeroes the ones its launches use), no tensor overlaps them (hence no tensor crosses a 4 GiB boundary and
 * none exceeds 4 GiB - 16 KiB), and the arena may be of any size. */
int smap_plan_create(const smap_op* ops, int n_ops, smap_plan** plan);
void smap_plan_destroy(smap_plan* plan);
/* Runs the whole schedule on `stream`.
 * input : [B,3,H,W] fp32 NCHW images; arena: activation arena; weights: weight blob;
 * out   : fp32 output buffer (hms [B,43,h,w] | det_d [B,14,h,w] | root_d [B,1,h,w] at the
 *         offsets recorded in the HEADSUM ops, plus the SMAP_STATUS_WORDS(B) int32 status words at status_off when the ops name them). */
int smap_plan_run(const smap_plan* plan, const float* input, void* arena, const void* weights,
                  float* out, void* stream);
/* The same with the images in n_inputs (1..SMAP_MAX_INPUTS, a divisor of B) separate buffers of B / n_inputs frames each,
 * in order: `inputs` is a HOST array of device pointers.  A launch that serves several of the caller's batches reads them
 * where they are (the stem indexes the buffers) instead of gathering them first (test.py:43-50 hands one batch at a time). */
#define SMAP_MAX_INPUTS 8
int smap_plan_run_inputs(const smap_plan* plan, const float* const* inputs, int n_inputs, void* arena,
                         const void* weights, float* out, void* stream);
/* Lanes on (1) / off (0, the default): see smap_op.lane.  Switching them on creates the side streams and events IN THIS CALL, on the current
 * device (all or nothing: a failure is this call's return code and leaves nothing behind); they are destroyed with the plan.  smap_plan_run
 * never creates anything (a first run inside a stream capture is fine) and joins the side lanes into the caller's stream on error returns
 * too.  A plan with lanes ON has one caller at a time -- its side streams are shared by whoever runs it; with lanes off a plan is immutable
 * and may be run from several host threads / executors at once. */
int smap_plan_set_lanes(smap_plan* plan, int on);
/* Bytes of arena and of output buffer the schedule touches, computed from the ops (either pointer may be NULL). */
int smap_workspace_bytes(const smap_plan* plan, int64_t* arena_bytes, int64_t* out_bytes);

/* ---- plan blob: the whole schedule as ONE relocatable byte image, so that a host that cannot run the Python schedule builder
 * (smap_amd/engine.py describes model/smap.py:313-353 as ops and packs the weights) can still run SMAP.forward from this header
 * alone: read the file, smap_plan_create_from_blob, hipMalloc info.arena_bytes / info.out_bytes, upload the blob's weight
 * section [weights_offset, +weights_bytes) to the device, smap_plan_run.  Little-endian, written by
 * smap_amd/engine.py::BackboneEngine.blob():
 *   smap_blob_header | smap_op[n_ops] at ops_offset | weight section at weights_offset (256-byte aligned) */
#define SMAP_BLOB_VERSION 2u
typedef struct smap_blob_info {
    int32_t frames, H, W;               /* input: [frames,3,H,W] fp32 NCHW                                      */
    int32_t out_h, out_w;               /* map size                                                              */
    int32_t n_hms, n_det, n_root;       /* channels of the three outputs (43, 14, 1)                             */
    int32_t precision;                  /* 0 = fp16 storage, 1 = split precision                                 */
    int32_t reserved;
    int64_t arena_bytes, out_bytes;     /* buffers to allocate (out_bytes includes the status word)              */
    int64_t weights_offset, weights_bytes;   /* the weight section inside the blob                               */
    int64_t hms_off, det_off, root_off, status_off;   /* byte offsets of [frames,C,out_h,out_w] fp32 maps / the int32 status word in `out` */
} smap_blob_info;
typedef struct smap_blob_header {
    char magic[8];                      /* "SMAPPLN1"                                                            */
    uint32_t version, sizeof_op, header_bytes;   /* version = SMAP_BLOB_VERSION: any other value is refused (a blob written for an older smap_op
                                           MEANING -- version 1: status_off named ONE status word, no seg_ / ksplit / lane / scale_hms
                                           fields -- must not load just because the struct sizes happen to agree) */
    int32_t n_ops;
    int64_t ops_offset;
    int64_t weights_offset, weights_bytes, arena_bytes, out_bytes;     /* = info's, kept flat for readers without the struct */
    smap_blob_info info;
} smap_blob_header;
/* blob: HOST memory.  Validates header, op size, every op (as smap_plan_create) and that the ops stay inside the arena /
 * output sizes the header states.  info may be NULL. */
int smap_plan_create_from_blob(const void* blob, size_t blob_bytes, smap_plan** plan, smap_blob_info* info);

/* Runs ops [first, first+count) only (tests, per-layer profiling). */
int smap_plan_run_range(const smap_plan* plan, int first, int count, const float* input,
                        void* arena, const void* weights, float* out, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SMAP_HIP_H */
