/*
 * simple_pose_hip.h - C ABI of libsimple_pose_hip.so (MI355X / gfx950 only).
 *
 * The upstream reference (liangheming/simple_pose) has no native code and no FFI: its hot path is
 * reached through Python objects that call torch.nn.functional.  This header is the boundary a
 * maintainer would bind instead (ctypes stubs in INTEGRATION.md); every entry point names the
 * reference lines it replaces.  Conventions:
 *
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (the library never allocates, frees or retains pointers past the call);
 *   - asynchronous on the caller's `hipStream_t` passed as `void* stream`; no implicit device sync;
 *   - returns SP_OK (0) or a negative SP_E* code; sp_last_error() gives the message of the last
 *     failure on the calling thread; nothing throws or aborts across the boundary;
 *   - activations are fp32 NHWC inside the network (channels % 4 == 0); the public tensors keep
 *     the reference's NCHW fp32 layout (input [B,3,H,W], heat maps [B,J,H,W]).
 */
#ifndef SIMPLE_POSE_HIP_H
#define SIMPLE_POSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL (-1)   /* bad argument / unsupported shape (message in sp_last_error) */
#define SP_ELAUNCH (-2)  /* HIP launch or runtime failure */

#define SP_ABI_VERSION 35

/* epilogue / layout flags of sp_conv_desc.flags */
#define SP_CONV_RELU 0x1u          /* y = max(y, 0) after scale/shift (+ residual) */
#define SP_CONV_OUT_NCHW 0x2u      /* store y as NCHW [B, N, out_h, out_w] (final_layer -> heat maps) */
#define SP_CONV_PIXEL_SHUFFLE 0x4u /* fused nn.PixelShuffle(2): weights packed with sp_pack order, see below */
#define SP_CONV_OUT_F32 0x10u      /* with SP_CONV_BF16: NHWC y and residual are fp32 (activation gradients in the bf16
                                      train step keep fp32 until the BatchNorm backward has removed their mean) */
#define SP_CONV_BN_Y_MASK 0x20u    /* sp_conv2d_dgrad_bn_bwd_stats* / sp_conv2d_dgrad_phases with bf16 activations AND gradients: `bn_y` is the
                                    * ReLU bit mask sp_bn_fold_apply_nhwc / sp_bn_apply_nhwc left (one byte per 8 channels), not the tensor y */
#define SP_CONV_BF16 0x8u          /* x, w_packed, residual and NHWC y are bf16 (fp32 accumulate; scale/shift and the NCHW
                                      output stay fp32); c_in % 8 == 0, k_pad % 64 == 0 */

/*
 * One launch of the fp32 implicit-GEMM convolution family:
 *   rows    M = batch * grid_h * grid_w      (output pixels of one phase)
 *   columns N = c_out
 *   depth   K = taps_h * taps_w * c_in       (zero taps allowed through k_pad)
 *   x[b, iy, ix, c] with iy = gy * stride + dy0 + ty * dy_step  (same for x), zero outside the image
 *   y[b, gy * oy_mul + oy_add, gx * ox_mul + ox_add, n] = act(acc * scale[n] + shift[n] + res[...])
 * A stride-s conv with padding p is (dy0 = -p, dy_step = +1, oy_mul = 1, oy_add = 0); output phase
 * (py,px) of ConvTranspose2d(k=4, s=2, p=1) is a 2x2-tap launch with (stride = 1, dy0 = py,
 * dy_step = -1, oy_mul = 2, oy_add = py).  `w` is packed [phases][n_pad][k_pad] fp32, K contiguous,
 * K ordered (ty, tx, c) - see sp_conv_weight_index().
 */
typedef struct sp_conv_desc {
    int32_t batch, in_h, in_w, c_in; /* x: NHWC [batch, in_h, in_w, c_in], c_in % 4 == 0 */
    int32_t grid_h, grid_w;          /* logical output grid per phase */
    int32_t c_out;                   /* N (real) */
    int32_t n_pad;                   /* packed rows per phase: multiple of 32, >= c_out (extra rows are zero) */
    int32_t taps_h, taps_w;          /* taps as stored in the packed weights (taps_w may exceed the real kw) */
    int32_t k_pad;                   /* packed K per row: multiple of 32, >= taps_h*taps_w*c_in */
    int32_t stride;
    int32_t dy0, dy_step, dx0, dx_step;
    int32_t out_h, out_w, out_c;     /* y: NHWC [batch, out_h, out_w, out_c] (or NCHW with SP_CONV_OUT_NCHW) */
    int32_t oy_mul, oy_add, ox_mul, ox_add;
    int32_t phases_y, phases_x;      /* 1,1 for conv; 2,2 for the k4s2p1 transposed conv (dy0/oy_add become per-phase) */
    uint32_t flags;
    int32_t tile_m, tile_n;          /* workgroup tile (rows x columns); 0,0 = sp_conv2d_default_tile().  Results do not
                                        depend on the tile: every output's K reduction order is the same for all shapes. */
    int32_t stride_x;                /* 0: same as `stride`.  Otherwise the x stride where it differs from the y stride (`stride`):
                                        the bf16 stem reads the 4-channel image as x-PAIRS of 8 values, where a stride of two
                                        pixels is a stride of one pair */
    int32_t kernel;                  /* SP_CONV_KERNEL_IGEMM (0, default), SP_CONV_KERNEL_RING / _RING_LW or SP_CONV_KERNEL_PW: which kernel structure runs the
                                        launch; same results bit for bit (same K order, same MFMA chain per output) */
    int32_t c_in_group;              /* 0: dense.  > 0 (ABI 33): grouped convolution with c_out == c_in (ResNeXt's conv2, nets/pose_resnet_dconv.py:101):
                                        an N tile of tile_n == c_in_group output channels reads only the c_in_group input channels of its own groups;
                                        k_pad = taps * c_in_group and the weights are packed by sp_pack_conv_weights_grouped (one block-diagonal panel
                                        [tile_n][k_pad] per N tile).  Implicit-GEMM kernel only, c_in_group a whole number of K tiles and of groups. */
} sp_conv_desc;

/* sp_conv_desc.kernel */
#define SP_CONV_KERNEL_IGEMM 0 /* 4-wave workgroups, register-staged double buffer: every dtype / flag / tile listed at sp_conv2d_default_tile */
#define SP_CONV_KERNEL_RING 1  /* bf16 only: persistent 8-wave workgroups fed by an LDS-DMA ring (buffer_load ... lds, counted vmcnt);
                                  tiles 256x256 256x128 128x256 256x64 128x128 192x128 192x256; needs sp_conv2d_ring_ok(desc) == 1 */
#define SP_CONV_KERNEL_RING_LW 3 /* the same ring with four extra "loader" waves per workgroup (one per SIMD) that issue every LDS-DMA piece; the eight
                                  MFMA waves only read fragments and multiply (round 5; same bits as 0 / 1); tiles 256x128 128x256 256x64 128x128
                                  192x128; needs sp_conv2d_ring_ok(desc) == 1 with desc.kernel set to this id */
#define SP_CONV_KERNEL_RING_LW4 4 /* (ABI 34) the loader-wave ring with FOUR MFMA waves, one per SIMD, instead of eight: the same workgroup tile as four wave
                                  * tiles of twice the area (fewer LDS fragment bytes per MFMA), and the 96-row tiles eight waves cannot cut; tiles
                                  * 192x128, 128x128, 96x128, 256x128, 128x256, 96x256, 64x128; same bits */
#define SP_CONV_KERNEL_PW 2    /* fp32 1x1 stride-1 NHWC layers with c_in 64 / 128 and c_out % 256 == 0 (bottleneck conv3, projection shortcut):
                                  one persistent workgroup per CU streams 64-row tiles, weights in registers (sp_conv2d_pw_ok); tile_m / tile_n unused */

/* ---- library ---------------------------------------------------------------------------------- */
int sp_abi_version(void);
const char* sp_last_error(void);

/* ---- network ops: replace torch.nn.functional calls inside nets/pose_resnet_dconv.py:251-265,
 *      nets/pose_resnet_duc.py (_forward_impl), nets/pose_hrnet.py:419-454 ------------------------ */

/* input [B,C,H,W] fp32 NCHW (C<=4) -> NHWC4 [B,H,W,4] (channel C..3 = 0); feeds conv1 (pose_resnet_dconv.py:158) */
int sp_nchw_to_nhwc4(const float* x_nchw, float* y_nhwc4, int batch, int channels, int h, int w, void* stream);

/* nn.Conv2d / nn.ConvTranspose2d(4,2,1) + folded eval-mode nn.BatchNorm2d (scale/shift) or bias (shift)
 * + residual add + nn.ReLU + nn.PixelShuffle(2), one kernel.  Replaces: conv1/bn1/relu (:252-254),
 * Bottleneck.forward (:112-133), deconv_layers (:230-249), final_layer (:173-178), DUC (nets/commons.py:36-41).
 * scale/shift: [c_out] or NULL (scale NULL -> 1, shift NULL -> 0); residual: same layout as y or NULL. */
int sp_conv2d_fwd(const sp_conv_desc* desc, const void* x, const void* w_packed, const float* scale,
                  const float* shift, const void* residual, void* y, void* stream);

/* Direct (non-im2col) 3x3 stride-1 pad-1 convolution for 32 -> 32 and 64 -> 64 channels in bf16 (HRNet's two high-resolution branches,
 * nets/pose_hrnet.py BasicBlock; ResNet-50 layer1.*.conv2): the halo tile of an 8x16 / 16x8 output tile goes through LDS once instead of
 * once per tap; the filter stays in registers (32 channels) or in LDS in MFMA-fragment order (64 channels) for a persistent workgroup.  Same arguments and bit-identical results as sp_conv2d_fwd for
 * the descriptors sp_conv3x3_direct_ok accepts (returns 1 / 0); tile fields are ignored. */
int sp_conv3x3_direct_ok(const sp_conv_desc* desc);
int sp_conv3x3_direct(const sp_conv_desc* desc, const void* x, const void* w_packed, const float* scale, const float* shift,
                      const void* residual, void* y, void* stream);

/* One HRNet BasicBlock (nets/pose_hrnet.py:34-51) of a 32-channel branch in ONE launch, bf16: y = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x).
 * `desc` describes either of the block's two convolutions (same geometry; sp_basic_block_c32_ok(desc) == 1: what sp_conv3x3_direct_ok
 * accepts at 32 channels); w1 / w2 packed as for sp_conv2d_fwd, scale / shift = the folded BatchNorms.  The intermediate stays in LDS
 * (rounded to bf16 exactly as the two-launch path stores it) and the residual comes from the input halo the workgroup already holds:
 * 2.5x less HBM traffic, bit-identical results.  y must not alias x. */
int sp_basic_block_c32_ok(const sp_conv_desc* desc);
int sp_basic_block_c32(const sp_conv_desc* desc, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                       const void* w2_packed, const float* scale2, const float* shift2, void* y, void* stream);

/* (ABI 35) The same block on a 64-channel branch (HRNet's 32 x 24 maps): both 64 x 576 filters stay in registers because the eight waves take roles - four
 * run conv1 of strip j while four run conv2 of strip j - 1 (4 x 24-pixel strips, one barrier per strip; csrc/conv_block64.hip).  Same arguments and
 * contract as sp_basic_block_c32 (`desc` = either convolution, sp_basic_block_c64_ok(desc) == 1: what sp_conv3x3_direct_ok accepts at 64 channels);
 * bit-identical to the two launches. */
int sp_basic_block_c64_ok(const sp_conv_desc* desc);
int sp_basic_block_c64(const sp_conv_desc* desc, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                       const void* w2_packed, const float* scale2, const float* shift2, void* y, void* stream);

/* (ABI 34) The tail of a stage-opening Bottleneck with 64 mid channels as ONE launch (bf16 NHWC, stride 1: layer1.0 of the ResNet pose nets and of HRNet):
 *     y = relu( bn3(conv1x1_{64->256}(a_main)) + bn_d(conv1x1_{64->256}(a_short)) )          nets/pose_resnet_dconv.py:99-103,120-131
 * a_main = the block's 3x3 output, a_short = the block input ([rows][64] each), weights packed by sp_pack_conv_weights ([256][64]), the folded
 * BatchNorms as (scale, shift).  Replaces the projection shortcut's launch + conv3's launch (the 256-channel shortcut tensor is neither written nor
 * read: 703 -> 301 MB at bs=128) with the same bits: the shortcut value is rounded to bf16 where the two-launch program stores it. */
/* sp_dual_pw_f32: the fp32 twin (csrc/conv_pw.hip; c_out a multiple of 256): 1,406 -> 602 MB at bs=128, the same bits as the two fp32 launches. */
int sp_dual_pw_bf16_ok(int64_t rows, int c_main, int c_short, int c_out);
int sp_dual_pw_f32_ok(int64_t rows, int c_main, int c_short, int c_out);
int sp_dual_pw_f32(const float* a_main, const float* w_main_packed, const float* scale_main, const float* shift_main, const float* a_short,
                   const float* w_short_packed, const float* scale_short, const float* shift_short, float* y, int64_t rows, int c_main, int c_short,
                   int c_out, int relu, void* stream);
int sp_dual_pw_bf16(const void* a_main, const void* w_main_packed, const float* scale_main, const float* shift_main, const void* a_short,
                    const void* w_short_packed, const float* scale_short, const float* shift_short, void* y, int64_t rows, int c_main, int c_short,
                    int c_out, int relu, void* stream);
/* One ResNet Bottleneck (nets/pose_resnet_dconv.py:112-133) with an identity shortcut, stride 1, 256 -> 64 -> 64 -> 256 channels, in ONE
 * launch, bf16: y = relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + x).  `desc` describes the block's 3x3 convolution
 * (sp_bottleneck_c64_ok(desc) == 1: what sp_conv3x3_direct_ok accepts at 64 channels); w1 [>=64][256], w2 [>=64][576], w3 [256][64]
 * packed as for sp_conv2d_fwd, scale / shift = the folded BatchNorms.  x is read once (halo included) and y written once: 2.4x less
 * HBM traffic than the three launches; the intermediates are rounded to bf16 exactly where those store them: bit-identical results.
 * y must not alias x. */
int sp_bottleneck_c64_ok(const sp_conv_desc* desc);
int sp_bottleneck_c64(const sp_conv_desc* desc, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                      const void* w2_packed, const float* scale2, const float* shift2, const void* w3_packed, const float* scale3,
                      const float* shift3, void* y, void* stream);

/* The ResNet stem (nets/pose_resnet_dconv.py:158-162: conv1 7x7 s2 p3, bn1, relu, maxpool 3x3 s2 p1) in ONE launch, fp32 or bf16 compute:
 * x fp32 NCHW [batch,3,h,w] -> y NHWC [batch,hp,wp,64] (fp32, or bf16 when `bf16`), hp = ((h+6-7)/2+1 + 2-3)/2+1.  w_packed / k_pad: the
 * stem weights exactly as sp_conv2d_fwd takes them for conv1 (fp32: sp_pack_conv_weights with c_in_packed 4, taps_w_packed 8 ->
 * [64][224]; bf16: c_in_packed 8, taps_w_packed 4, pair_s0 1 -> [64][256]); scale / shift = the folded bn1.  The 128 x 96 x 64 map
 * between conv and pooling never reaches HBM and K is not padded to a GEMM tile: bit-identical to
 * sp_nchw_to_nhwc4[_bf16] -> sp_conv2d_fwd -> sp_maxpool3x3s2_nhwc[_bf16].  sp_stem7_pool_ok: the sizes the kernel's 32-bit offsets cover. */
int sp_stem7_pool_ok(int batch, int h, int w);
int sp_stem7_pool(const float* x, const void* w_packed, int k_pad, const float* scale, const float* shift, void* y, int bf16,
                  int batch, int h, int w, void* stream);
/* The same launch on uint8 BGR crops [batch,h,w,3] (what cv.warpAffine / sp_warp_affine_u8c3 produce): the collate normalisation of
 * datasets/coco.py:136 (`x / 255 - mean`, BGR -> RGB; mean_rgb_host = {0.485, 0.456, 0.406}) happens while the patch is loaded - the
 * arithmetic of sp_u8hwc_bgr_to_nhwc, same bits as that launch followed by sp_conv2d_fwd and sp_maxpool3x3s2_nhwc. */
int sp_stem7_pool_u8(const unsigned char* crops_bgr, const float* mean_rgb_host, const void* w_packed, int k_pad, const float* scale,
                     const float* shift, void* y, int bf16, int batch, int h, int w, void* stream);

/* HRNet's stem (nets/pose_hrnet.py:419-425: conv1 3x3 s2 + bn1 + relu, conv2 3x3 s2 + bn2 + relu) in ONE launch, bf16 compute:
 * x fp32 NCHW [batch,3,h,w] -> y bf16 NHWC [batch,h/4,w/4,64].  w1_packed / k1_pad: conv1's weights as sp_conv2d_fwd takes them on the bf16
 * NHWC4 image (sp_pack_conv_weights with c_in_packed 8, taps_w_packed 2, pair_s0 1 -> [64][64]); w2_packed: conv2's [64][576]; scale / shift:
 * the folded bn1 / bn2.  conv1's map never reaches HBM: bit-identical to sp_nchw_to_nhwc4_bf16 -> sp_conv2d_fwd -> sp_conv2d_fwd. */
/* HRNet transition1 as ONE launch (bf16; nets/pose_hrnet.py:327-366, used at :431-437): y_hi = relu(bn(conv3x3(x, 256 -> 32, stride 1, pad 1))),
 * y_lo = relu(bn(conv3x3(x, 256 -> 64, stride 2, pad 1))) from one staging of x [batch, h, w, 256] (h, w even).  wa_packed / wb_packed:
 * sp_pack_conv_weights layouts [32][k_pad] / [64][k_pad] with k_pad = 2304 = (tap, channel); scale / shift: folded BatchNorm (sp_fold_bn).
 * Replaces the two sp_conv2d_fwd launches of those layers; reduction order (64-channel chunk, tap, channel): equal to them up to fp32
 * summation order.  sp_hrnet_transition1_ok: whether (c_in, h, w) is a shape the kernel takes. */
int sp_hrnet_transition1_ok(int c_in, int h, int w);
int sp_hrnet_transition1(const void* x, int batch, int h, int w, const void* wa_packed, int k_pad, const float* scale_a, const float* shift_a,
                         const void* wb_packed, const float* scale_b, const float* shift_b, void* y_hi, void* y_lo, void* stream);
int sp_hrnet_stem_ok(int batch, int h, int w);
int sp_hrnet_stem(const float* x, const void* w1_packed, int k1_pad, const float* scale1, const float* shift1, const void* w2_packed,
                  const float* scale2, const float* shift2, void* y, int batch, int h, int w, void* stream);

/* 1 when `desc` can run with kernel = SP_CONV_KERNEL_PW (see there).  Same bits as the tiled kernel. */
int sp_conv2d_pw_ok(const sp_conv_desc* desc);

/* 1 when `desc` (flags, shapes, tile_m x tile_n) can run with kernel = SP_CONV_KERNEL_RING: bf16 NHWC in and out (ReLU, residual and
 * fused PixelShuffle allowed; no NCHW / fp32 output), c_in % 64 == 0, taps <= 32, k_pad / 64 >= the tile's ring depth, tile_n | n_pad. */
int sp_conv2d_ring_ok(const sp_conv_desc* desc);

/* Name of the kernel instantiation a launch of `desc` resolves to, as rocprofv3's kernel trace reports it (without the
 * "void (anonymous namespace)::" prefix and the argument list).  variant 0 = sp_conv2d_fwd, 1 = sp_conv2d_fwd_bn_stats,
 * 2 = sp_conv2d_dgrad_bn_bwd_stats, 3 = sp_conv3x3_direct, 4 = sp_basic_block_c32, 5 = sp_bottleneck_c64, 6 = sp_basic_block_c64 (`desc` = the
 * block's 3x3 convolution).  Produced by the launch dispatch itself (nothing is launched), so
 * profiles and bench.py's roofline line key on exactly what ran. */
int sp_conv2d_kernel_name(const sp_conv_desc* desc, int has_residual, int variant, char* buf, int cap);

/* The tile sp_conv2d_fwd picks when desc->tile_m == tile_n == 0; legal tiles: 128x128 64x128 128x64 64x64 256x64 128x32
 * (tile_n must divide n_pad).  Host code may time the legal tiles once per layer shape and pin the fastest. */
int sp_conv2d_default_tile(const sp_conv_desc* desc, int* tile_m, int* tile_n);

/* nn.MaxPool2d(3, 2, 1) on NHWC fp32 (pose_resnet_dconv.py:162,255) */
int sp_maxpool3x3s2_nhwc(const float* x, float* y, int batch, int h, int w, int c, void* stream);

/* nn.PixelShuffle(2) on NHWC fp32: x [B,h,w,c] -> y [B,2h,2w,c/4], y[b,2Y+i,2X+j,k] = x[b,Y,X,4k+2i+j]
 * (the bare shuffle that opens the DUC head, nets/pose_resnet_duc.py:228; the two DUC blocks fuse theirs into the conv) */
int sp_pixel_shuffle2_nhwc(const float* x, float* y, int batch, int h, int w, int c, void* stream);

/* SELayer (nets/commons.py:4-18; first block of each layer when reduction=True, pose_resnet_dconv.py:108-110,126-127):
 * squeeze y[b,c] = mean_hw x[b,hw,c]; the two 1x1 FCs run through sp_conv2d_fwd on the [B,1,1,C] tensor;
 * excite + block tail y = relu(x * sigmoid(gate_logits[b,c]) + identity). */
int sp_global_avg_pool_nhwc(const float* x, float* y, int batch, int hw, int c, void* stream);
int sp_se_gate_add_relu_nhwc(const float* x, const float* gate_logits, const float* identity, float* y, int batch, int hw,
                             int c, void* stream);

/* y = base + nearest_upsample(x, factor) (+ ReLU): x NHWC [B,h,w,c], base / y NHWC [B,h*f,w*f,c] (base may alias y;
 * factor 1 = plain add).  HighResolutionModule fuse sum, nets/pose_hrnet.py:192-202,250-257 */
int sp_upsample_add_nhwc(const float* x, const float* base, float* y, int batch, int h, int w, int c, int factor, int relu,
                         void* stream);
/* All upsampled terms of one HRNet fuse output in one pass (pose_hrnet.py:250-257, `y = y + fuse_layers[i][j](x[j])` for j >= i):
 * y = [relu](((base + up(xs[0], f0)) + up(xs[1], f1)) + up(xs[2], f2)), 1..3 terms, factor 1 = the identity term; xs[k] is
 * [batch, out_h / fk, out_w / fk, c] of the tensors' dtype (bf16 != 0: bf16, else fp32).  fp32: the bits of the chained sp_upsample_add_nhwc
 * launches; bf16: the sum is formed in fp32 and rounded once (the chain rounded after every term).  xs / factors: host arrays. */
int sp_upsample_add_n_nhwc(const void* base, int bf16, int n_terms, const void* const* xs, const int32_t* factors, void* y, int batch,
                           int out_h, int out_w, int c, int relu, void* stream);

/* bf16 NHWC (8 channels = 16 B per lane) variants of the layout / pooling / fuse kernels, for SP_CONV_BF16 networks.
 * The network input stays the reference's fp32 NCHW tensor; channels are padded to 8. */
int sp_nchw_to_nhwc8_bf16(const float* x_nchw, void* y_nhwc8, int batch, int channels, int h, int w, void* stream);
int sp_maxpool3x3s2_nhwc_bf16(const void* x, void* y, int batch, int h, int w, int c, void* stream);
int sp_pixel_shuffle2_nhwc_bf16(const void* x, void* y, int batch, int h, int w, int c, void* stream);
int sp_upsample_add_nhwc_bf16(const void* x, const void* base, void* y, int batch, int h, int w, int c, int factor,
                              int relu, void* stream);
/* SELayer (nets/commons.py:4-18) on bf16 tensors: squeeze (fp64 sums, bf16 result [B, c]) and excite + add + ReLU; gate_logits = the
 * bf16 output of the second FC (a 1x1 sp_conv2d_fwd on the [B,1,1,c] tensor) */
int sp_global_avg_pool_nhwc_bf16(const void* x, void* y, int batch, int hw, int c, void* stream);
int sp_se_gate_add_relu_nhwc_bf16(const void* x, const void* gate_logits, const void* identity, void* y, int batch, int hw,
                                  int c, void* stream);

/* ---- decoders: metrics/pose_metrics.py --------------------------------------------------------- */

/* BasicKeyPointDecoder.heat_map_to_axis (:11-24): coords [B,J,2], max_val [B,J] */
int sp_heat_map_to_axis(const float* heat_nchw, int batch, int joints, int h, int w, float* coords, float* max_val,
                        void* stream);
/* GaussTaylorKeyPointDecoder.__call__ (:62-107), kernel_size odd <= 15; trans_inv [B,2,3]; kps [B,J,2]; max_val [B,J] */
int sp_decode_gauss_taylor(const float* heat_nchw, const float* trans_inv, int batch, int joints, int h, int w,
                           int kernel_size, float* kps, float* max_val, void* stream);
/* BasicKeyPointDecoder.__call__ (:26-52) */
int sp_decode_basic(const float* heat_nchw, const float* trans_inv, int batch, int joints, int h, int w, float* kps,
                    float* max_val, void* stream);

/* HeatMapAcc.__call__ (:212-245) on the arg-max coordinates of predictions and targets (two sp_heat_map_to_axis calls):
 * acc_out = one device float, no host sync (the reference's .item() calls at :237 are gone).  mask [B,J] (may be NULL): the
 * solver evaluates maps multiplied by the joint mask (ddp...:130-131); a zero mask takes the joint out exactly as that does */
int sp_heat_map_acc(const float* pred_coords, const float* label_coords, const float* mask, int batch, int joints, int h, int w,
                    float distance_thresh, float norm_frac, float* acc_out, void* stream);

/* ---- input contract: datasets/coco.py:124-148 (collate_fn) -----------------------------------------
 * BGR u8 HWC crops [B,h,w,3] (device) -> RGB fp32 NCHW [B,3,h,w] = x/255 - mean_rgb[c] (no std division, coco.py:136);
 * mean_rgb_host: 3 floats in HOST memory */
int sp_u8hwc_bgr_to_nchw_f32(const unsigned char* img, float* out, int batch, int h, int w, const float* mean_rgb_host,
                             void* stream);

/* person crops from the full image: dst[n] = cv.warpAffine(src, m_fwd[n], (out_w, out_h), flags=INTER_LINEAR) for
 * `crops` forward (src -> dst) 2x3 float64 matrices (HOST memory; inverted on the host as OpenCV does) over one uint8 HxWx3 image; BORDER_CONSTANT 0; OpenCV's
 * fixed-point bilinear arithmetic restated (not pinned against cv2: it is absent from the build image) - naive_data.py:50 */
int sp_warp_affine_u8c3(const unsigned char* src, int src_h, int src_w, const double* m_fwd, int crops, unsigned char* dst, int out_h,
                        int out_w, void* stream);

/* ---- after decode: result scores and per-image OKS-NMS -----------------------------------
 * kps_to_dict_ (metrics/pose_metrics.py:172-179): score[b] = mean_j(max_val[b,j]) + max_j(max_val[b,j]) */
int sp_pose_score(const float* max_val, int batch, int joints, float* score, void* stream);
/* eval.py:166-174: score[p] = box_score[p] * mean(kps[p,:,2][kps[p,:,2] > in_vis_thre]) (0 without a visible joint), float64 as
 * the reference computes after its JSON round trip; kps [P,J,3] = (x, y, max_val) fp32 from the decoder; kps64 (may be NULL)
 * receives the float64 copy sp_oks_nms consumes */
int sp_pose_rescore(const float* kps, const double* box_score, int persons, int joints, double in_vis_thre, double* kps64,
                    double* score, void* stream);
/* oks_nms (datasets/naive_data.py:153-173; oks_iou :120-150) for `groups` images in one launch; the persons of image g are
 * rows seg[g]..seg[g+1]-1 (seg: device int32 [groups+1]; max_group = the largest image, <= 2048, known to the host).
 * sigmas_host: `joints` doubles in HOST memory or NULL (COCO's 17); vis_thresh < 0 = in_vis_thresh None.
 * keep [P]: for image g the picked GLOBAL row indices in pick order at keep[seg[g]..], padded with -1; keep_count [groups].
 * Order: descending score, equal scores: higher index first (numpy's own tie order is unspecified). */
int sp_oks_nms(const double* kps, const double* scores, const double* areas, const int32_t* seg, int groups, int max_group,
               int joints, const double* sigmas_host, double thresh, double vis_thresh, int32_t* keep, int32_t* keep_count,
               void* stream);

/* the same normalisation written directly in the network's input layout (the stem's loader format): NHWC4 fp32 [B,h,w,4] or,
 * out_bf16 == 1: NHWC8 bf16 [B,h,w,8]; out_bf16 == 2: NHWC4 bf16 [B,h,w,4] (what the bf16 inference stem reads); pad channels are zero.  Replaces sp_u8hwc_bgr_to_nchw_f32 + sp_nchw_to_nhwc4 when the crops
 * arrive as uint8 (1 byte per value over PCIe, one pass on the GPU). */
int sp_u8hwc_bgr_to_nhwc(const unsigned char* img, void* out, int out_bf16, int batch, int h, int w, const float* mean_rgb_host,
                         void* stream);

/* NCHW fp32 [B,C<=4,h,w] -> NHWC4 bf16 [B,h,w,4] (8 bytes per pixel; w even: the stem conv reads pixel pairs as 8 channels) */
int sp_nchw_to_nhwc4_bf16(const float* x, void* y, int batch, int channels, int h, int w, void* stream);

/* ---- encoders: commons/transforms.py ----------------------------------------------------------- */

/* RefineSimpleTransform.get_heat_map (:167-191), batched: joints [B,J,3] (x,y,vis in heat-map px) ->
 * targets [B,J,h,w], weights [B,J] */
int sp_encode_gauss_refine(const float* joints, int batch, int joints_n, int h, int w, float sigma, float* targets,
                           float* weights, void* stream);
/* BasicSimpleTransform.get_heat_map (:80-116): joints in INPUT px, quantised centre, truncated patch */
int sp_encode_gauss_basic(const float* joints, int batch, int joints_n, int h, int w, float sigma, int stride,
                          float* targets, float* weights, void* stream);

/* ---- loss: processors/ddp_pose_resnet_solver.py:94,117 ------------------------------------------
 * loss = 0.5 * mean((pred*m - target*m)^2) over all B*J*h*w elements; grad (may be NULL) = d loss / d pred.
 * `loss_out` is one device float, written by the kernel (no host sync). `workspace`: >= 4096 bytes, zeroed by the call. */
int sp_masked_mse(const float* pred, const float* target, const float* mask, int batch, int joints, int hw,
                  float* loss_out, float* grad, void* workspace, void* stream);

/* ---- training step: processors/ddp_pose_resnet_solver.py:110-133 (model.train() forward, loss.backward(),
 *      optimizer.step()) ---------------------------------------------------------------------------------------------
 * Activations NHWC; argument `bf16`: bit 0 = activations (z, relu_src, dz, pool input) are bf16, bit 1 = activation
 * gradients (dy, dres, pool dx/dy) are bf16; statistics and the gradients of the affine parameters are always fp32;
 * `rows` = B*H*W.  `workspace` of the reductions: >= SP_REDUCE_WORKSPACE_BYTES(c) bytes (per-workgroup partial sums in fp64,
 * folded in a fixed order by a second small kernel: deterministic, no float atomics); c <= 4096. */
#define SP_REDUCE_WORKSPACE_BYTES(c) ((int64_t)4 << 20)

/* nn.BatchNorm2d in train mode, statistics half: per-channel batch mean and 1/sqrt(biased var + eps) of z [rows, c];
 * running_mean/var (may both be NULL) are updated with `momentum` and the UNBIASED variance, as torch does. */
int sp_bn_train_stats_nhwc(const void* z, int bf16, int64_t rows, int c, float eps, float momentum, float* mean, float* invstd,
                           float* running_mean, float* running_var, void* workspace, void* stream);
/* The same statistics without a pass over z: sp_conv2d_fwd_bn_stats is sp_conv2d_fwd (no scale/shift/residual/ReLU, NHWC store in
 * the conv's dtype) whose epilogue also writes, per (phase, M tile) of the launch (the tile's wave rows are added inside the launch), the per-channel sum and sum of squares
 * of the values it stores ([partial_rows][n_pad] fp32 each; sp_conv2d_bn_stats_rows gives partial_rows for the tile the launch
 * will use); sp_bn_train_stats_from_conv folds them in index order in fp64 and finishes like sp_bn_train_stats_nhwc. */
int sp_conv2d_bn_stats_rows(const sp_conv_desc* desc, int* partial_rows);
int sp_conv2d_fwd_bn_stats(const sp_conv_desc* desc, const void* x, const void* w_packed, void* y, float* stats_sum,
                           float* stats_sumsq, int stats_rows_capacity, void* stream);
/* sp_conv2d_fwd_bn_stats of a bf16 1x1 stride-1 convolution (c_in <= 512) whose input is relu(BatchNorm(z_in)) of the previous layer
 * (`Bottleneck.forward`: out = relu(bn2(conv2(.))); out = conv3(out), nets/pose_resnet_dconv.py:118-122): the map is applied while z_in is staged
 * (in_mean / in_invstd: that layer's batch statistics, in_gamma / in_beta its affine pair), and the launch also writes the activation `y_in`
 * [same layout as z_in] and, when not null, its ReLU bit mask (one byte per 8 channels) - what the backward pass and the weight gradient read.
 * Replaces sp_bn_apply_nhwc(relu = 1) + sp_conv2d_fwd_bn_stats with the same bits (ABI 33). */
int sp_conv2d_fwd_bn_stats_abn(const sp_conv_desc* desc, const void* z_in, const float* in_mean, const float* in_invstd, const float* in_gamma,
                               const float* in_beta, void* y_in, void* relu_mask_in, const void* w_packed, void* y, float* stats_sum,
                               float* stats_sumsq, int stats_rows_capacity, void* stream);
int sp_bn_train_stats_from_conv(const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride, int64_t rows, int c,
                                float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                void* stream);
/* Backward counterpart: a dgrad launch (sp_conv2d_fwd on the dgrad packing, fp32 NHWC store, optional in-place accumulate) whose
 * output IS dy of a BatchNorm+ReLU layer.  Given that layer's saved output bn_y (ReLU mask), input bn_z and statistics, the
 * epilogue also writes the partial sums of g = dy*(bn_y > 0) and g*xhat, one row per (phase, M tile) as above;
 * several launches that together cover dy (the phases of a stride-2 dgrad) fill consecutive row ranges of the same buffers.
 * sp_bn_bwd_sums_from_conv folds them (index order, fp64) into dbeta / dgamma; sp_bn_train_bwd_apply_nhwc finishes the layer. */
int sp_conv2d_dgrad_bn_bwd_stats(const sp_conv_desc* desc, const void* dz, const void* w_packed, const void* accumulate, void* dx,
                                 const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                                 float* sum_g_xhat, int stats_rows_capacity, void* stream);
int sp_bn_bwd_sums_from_conv(const float* sum_g, const float* sum_g_xhat, int partial_rows, int stride, int c, float* dgamma,
                             float* dbeta, void* stream);
/* The same launch when dy is the gradient of a block output y = relu(bn3(z) + bn_ds(z2)) with a projection shortcut (the first
 * Bottleneck of a stage, pose_resnet_dconv.py:124-131): the shortcut BatchNorm's dy is the same g, so the epilogue also leaves
 * the partial sums of g * xhat2 (its d gamma; its d beta is sum g): sp_bn_bwd_sums_from_conv(sum_g, sum_g_xhat2, ...) folds them. */
int sp_conv2d_dgrad_bn_bwd_stats2(const sp_conv_desc* desc, const void* dz, const void* w_packed, const void* accumulate, void* dx,
                                  const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                                  float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd,
                                  float* sum_g_xhat2, int stats_rows_capacity, void* stream);
/* sp_conv2d_dgrad_bn_bwd_stats(2) for conv1 of an identity Bottleneck with bf16 gradients: the block input's gradient also receives the
 * residual share g = dy_out * (ReLU mask of the block output, pose_resnet_dconv.py:124-131).  Instead of `accumulate` = g written by bn3's
 * backward pass, this launch takes (acc_dy = dy_out, acc_mask = the block output's ReLU bit mask) and adds dy_out where the bit is set: the
 * pass writes 2 bytes per element less, the bits are the same.  bn2_* all NULL: one BatchNorm. */
int sp_conv2d_dgrad_bn_bwd_stats_macc(const sp_conv_desc* desc, const void* dz, const void* w_packed, const void* acc_dy, const void* acc_mask,
                                      void* dx, const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                                      float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd, float* sum_g_xhat2,
                                      int stats_rows_capacity, void* stream);
/* The dgrad of a STRIDE-2 conv is a family of launches, one descriptor per output phase (their tap counts differ: 3x3 -> 2x2, 2x1, 1x2, 1x1).
 * This entry runs the family as ONE launch (blockIdx.y = phase, per-phase geometry / packed weights in the kernel arguments): descs[i] and
 * w_packed[i] are exactly what sp_conv2d_fwd / sp_conv2d_dgrad_bn_bwd_stats(2) would take phase by phase (same batch, tensors, flags and
 * tile; 2-4 phases), the partial rows land phase-major as that sequence leaves them, the bits of dx and of the sums are the same.
 * bn_z NULL: plain dgrad (bn_* / sum_* unused).  Replaces the input-gradient half of `loss.backward()` for nn.Conv2d(stride=2)
 * (nets/pose_resnet_dconv.py:100-101: conv2 of a stage's first Bottleneck). */
int sp_conv2d_dgrad_phases(const sp_conv_desc* descs, int n_phases, const void* dz, const void* const* w_packed, const void* accumulate,
                           void* dx, const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                           float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd, float* sum_g_xhat2,
                           int stats_rows_capacity, void* stream);
/* The fold as the prologue of the pass that consumes it (round 4): sp_bn_fold_apply_nhwc = sp_bn_train_stats_from_conv + sp_bn_apply_nhwc
 * in ONE launch - every workgroup folds the partial rows of its 64-channel slab itself (same order as the stand-alone fold: same bits),
 * the first row stripe publishes mean / invstd / running statistics; sp_bn_fold_bwd_apply_nhwc = sp_bn_bwd_sums_from_conv (+ the same for
 * a second BatchNorm sharing g: sum_g_xhat2 -> dgamma2, sum g -> dbeta2; all three NULL when there is none) + sp_bn_train_bwd_apply_nhwc.
 * Meant for tensors with few partial rows (the caller decides: layer3 / layer4 of a ResNet at 32 images have 24-96); replaces
 * `nn.BatchNorm2d` forward / backward of processors/ddp_pose_resnet_solver.py:115,118 together with the conv launch that left the rows. */
int sp_bn_fold_apply_nhwc(const void* z, int bf16, const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride,
                          int64_t total_rows, float eps, float momentum, const float* gamma, const float* beta, const void* residual, void* y,
                          int64_t rows, int c, int relu, float* mean, float* invstd, float* running_mean, float* running_var, void* relu_mask,
                          void* stream);
/* `bf16` of the backward passes is a bit word: 1 = activations (z, relu_src, dz) are bf16; 2 = the activation gradients (dy, dres) are bf16 too
 * (PoseTrainer grad_dtype "bf16"); 4 = relu_src is the ReLU BIT MASK the forward pass left in `relu_mask` (uint8 [rows * c / 8], bit e of
 * byte i = channel 8 i + e passed the ReLU; bf16 activations, c % 8 == 0) instead of the tensor y: 1/16 of the bytes. */
int sp_bn_fold_bwd_apply_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* sum_g, const float* sum_g_xhat,
                              const float* sum_g_xhat2, int partial_rows, int stride, const float* mean, const float* invstd,
                              const float* gamma, int64_t total_rows, int64_t rows, int c, float* dgamma, float* dbeta, float* dgamma2,
                              float* dbeta2, void* dz, void* dres, int dres_accumulate, void* stream);
/* y = [relu]((z - mean) * invstd * gamma + beta [+ residual])   (Bottleneck.forward tail, pose_resnet_dconv.py:124-131) */
int sp_bn_apply_nhwc(const void* z, int bf16, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* residual, void* y, int64_t rows, int c, int relu, void* relu_mask /* NULL, or see above */, void* stream);
/* backward of [relu](bn(z) [+ residual]): g = dy * (relu_src > 0) (relu_src NULL: g = dy); dgamma = sum g*xhat,
 * dbeta = sum g, dz = gamma*invstd*(g - dbeta/rows - xhat*dgamma/rows); dres (NULL or tensor) = g or += g */
int sp_bn_train_bwd_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                         const float* gamma, int64_t rows, int c, void* dz, float* dgamma, float* dbeta, void* dres,
                         int dres_accumulate, void* workspace, void* stream);
/* The ResNet stem in training (pose_resnet_dconv.py:251-256: bn1 -> relu -> maxpool): relu(bn1(z)) feeds the pooling only, so
 * sp_bn_apply_maxpool_nhwc applies the BatchNorm + ReLU map to the taps of each window and writes the POOLED map + the winning tap per element
 * (same values and indices as sp_bn_apply_nhwc + sp_maxpool3x3s2_idx_nhwc, without the full-resolution tensor in between), and
 * sp_bn_maxpool_bwd_nhwc is the backward of that pair from the pooled gradient: d gamma / d beta summed over the pooled grid (a pooled
 * gradient reaches exactly one stem pixel; xhat and the ReLU mask re-formed from z at the winner; fp64 partials, fixed order), then dz of
 * the stem conv by gathering each pixel's <= 4 windows.  `bf16`: bit 0 activations, bit 1 gradients (as the other backward passes);
 * workspace: SP_REDUCE_WORKSPACE_BYTES. */
int sp_bn_apply_maxpool_nhwc(const void* z, int bf16, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y,
                             void* idx, int batch, int h, int w, int c, void* stream);
int sp_bn_maxpool_bwd_nhwc(const void* dy_pooled, int bf16, const void* idx, const void* z, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int batch, int h, int w, int c, float* dgamma, float* dbeta, void* dz,
                           void* workspace, void* stream);
/* nn.SyncBatchNorm (ddp...:89-90) = the same three steps with a cross-rank SUM between the halves:
 *   forward : sp_bn_train_partial_nhwc -> all-reduce(sums [c][2] fp64: sum, sum of squares) -> sp_bn_train_finalize(total_rows)
 *   backward: sp_bn_train_bwd_reduce_nhwc (LOCAL dgamma, dbeta = this rank's parameter gradients) -> all-reduce(copy of them)
 *             -> sp_bn_train_bwd_apply_nhwc(global sums, total_rows = rows over all ranks)
 * sp_bn_train_stats_nhwc / sp_bn_train_bwd_nhwc are exactly these halves back to back with total_rows = rows. */
int sp_bn_train_partial_nhwc(const void* z, int bf16, int64_t rows, int c, double* sums, void* workspace, void* stream);
/* the forward half without the pass over z: fold the partial rows sp_conv2d_fwd_bn_stats left into this rank's fp64 (sum, sum of
 * squares) per channel, [c][2] as above -> all-reduce -> sp_bn_train_finalize.  Several layers may fold into consecutive slices of
 * ONE buffer and share one all-reduce (the conv1 / downsample pair of a block's first bottleneck reads the same input). */
int sp_bn_sums_from_conv(const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride, int c, double* sums, void* stream);
int sp_bn_train_finalize(const double* sums, int64_t total_rows, int c, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, void* stream);
/* SyncBatchNorm with one launch less on each side of the exchange: sp_bn_apply_sums_nhwc = sp_bn_train_finalize + sp_bn_apply_nhwc (every thread
 * finalises its own 4 channels from the all-reduced sums, the first row's threads publish mean / invstd / running statistics; same values);
 * sp_bn_bwd_sums_from_conv2 = sp_bn_bwd_sums_from_conv that ALSO writes the two sums where the message is assembled (the parameter gradients keep
 * this rank's sums, the message carries them to the other ranks: no concatenation launch). */
int sp_bn_apply_sums_nhwc(const void* z, int bf16, const double* sums, int64_t total_rows, float eps, float momentum, const float* gamma,
                          const float* beta, const void* residual, void* y, int64_t rows, int c, int relu, float* mean, float* invstd,
                          float* running_mean, float* running_var, void* stream);
/* sp_bn_bwd_sums_from_conv for a BatchNorm AND the projection shortcut's BatchNorm that shares its g (sum_g_xhat2: the third array of
 * sp_conv2d_dgrad_bn_bwd_stats2; d beta2 = the same sum g) in one launch; the bits of two separate calls. */
int sp_bn_bwd_sums_from_conv_pair(const float* sum_g, const float* sum_g_xhat, const float* sum_g_xhat2, int partial_rows, int stride, int c,
                                  float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, void* stream);
int sp_bn_bwd_sums_from_conv2(const float* sum_g, const float* sum_g_xhat, int partial_rows, int stride, int c, float* dgamma, float* dbeta,
                              float* dgamma_copy, float* dbeta_copy, void* stream);
int sp_bn_train_bwd_reduce_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                                int64_t rows, int c, float* dgamma, float* dbeta, void* workspace, void* stream);
int sp_bn_train_bwd_apply_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                               const float* gamma, const float* sum_dgamma, const float* sum_dbeta, int64_t total_rows, int64_t rows,
                               int c, void* dz, void* dres, int dres_accumulate, void* stream);
/* backward of nn.PixelShuffle(2) (DUC head, nets/commons.py:36-41): dy [B,2h,2w,c/4] fp32 -> dx [B,h,w,c] fp32 */
int sp_pixel_unshuffle2_nhwc(const float* dy, float* dx, int batch, int h, int w, int c, void* stream);
/* the same permutation on a bf16 gradient (activation gradients kept in bf16: PoseTrainer grad_dtype "bf16" on the DUC head); c % 32 == 0 */
int sp_pixel_unshuffle2_nhwc_bf16(const void* dy, void* dx, int batch, int h, int w, int c, void* stream);
/* x [B,channels,h,w] fp32 -> y [B,h,w,c_pad] (fp32, or bf16 when y_bf16), channels >= `channels` zero-filled: puts
 * d loss / d heat-map (sp_masked_mse's grad) into the layout and K-tile padding the final layer's backward launches read */
int sp_nchw_to_nhwc_pad(const float* x, void* y, int y_bf16, int batch, int channels, int h, int w, int c_pad, void* stream);
/* sum over batch and pixels of an NCHW tensor [batch, channels, hw] -> [channels]: the final layer's bias gradient straight from
 * d loss / d heat maps (sp_masked_mse's grad), written where the caller's flat gradient buffer keeps it */
int sp_channel_sum_nchw(const float* x, int batch, int channels, int hw, float* sum, void* stream);
/* sum over rows of a [rows, c] tensor (conv bias gradient) */
int sp_channel_sum_nhwc(const float* a, int64_t rows, int c, float* sum, void* workspace, void* stream);
/* backward of nn.MaxPool2d(3,2,1) (first maximum of a window wins, as torch); x = the pool's input */
int sp_maxpool3x3s2_bwd_nhwc(const void* x, int bf16, const void* dy, void* dx, int batch, int h, int w, int c, void* stream);
/* the training pair: forward that also records the winning tap (ky*3+kx, one byte per output element, `idx` = [B,ho,wo,c] bytes),
 * and the backward that gathers through it (<= 4 (index, dy) pairs per input pixel); same results as the function above */
int sp_maxpool3x3s2_idx_nhwc(const void* x, int bf16, void* y, void* idx, int batch, int h, int w, int c, void* stream);
int sp_maxpool3x3s2_bwd_idx_nhwc(const void* idx, const void* dy, int bf16, void* dx, int batch, int h, int w, int c, void* stream);
/* torch.optim.Adam (amsgrad False, weight_decay 0) over flat buffers of n (multiple of 4) floats, `step` = 1,2,...;
 * grad is multiplied by grad_scale first (1/world_size after a SUM all-reduce) - ddp...:70-72,119.  lr / betas / eps are doubles:
 * 1 - beta, the bias corrections and lr / (1 - beta1^step) are formed in double and rounded to fp32 once, as torch does */
int sp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                 double beta2, double eps, int step, float grad_scale, void* stream);

/* The same update with the step's eight scalars (step size, betas, 1 - betas, eps, sqrt of the second bias correction, gradient scale) in
 * device memory: sp_adam_set_scalars forms them on the host exactly as sp_adam_step does and writes them with one tiny launch;
 * sp_adam_step_dev launches then have arguments that never change from step to step, so a captured train step (hipGraph: PoseTrainer.capture)
 * replays - bit-identical to sp_adam_step with the same hyper-parameters. */
int sp_adam_set_scalars(double lr, double beta1, double beta2, double eps, int step, float grad_scale, float* scalars8, void* stream);
int sp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* scalars8, void* stream);
/* weight gradient of one conv / transposed-conv launch family: dW[n][(ty,tx,c)] = sum_m g[m][n] * im2col(a)[m][(ty,tx,c)],
 * `desc` describing how `a` is gathered (as in sp_conv2d_fwd; tile/out_* ignored; flags & SP_CONV_BF16: g and a are
 * bf16, dW stays fp32), g = [rows, g_channels] NHWC.
 * The result is written in the reference's weight layout: dw[n*dst_stride_n + c*dst_stride_c + ty*kw_valid + tx]
 * for n < n_valid, c < c_valid, tx < kw_valid (every such element is overwritten).
 *
 * sp_conv2d_wgrad_batched runs the weight gradients of up to 64 layers in a handful of launches (nets/pose_resnet_dconv.py:99-103,
 * 158,236-244 under loss.backward(), ddp...:117-119): the pixel range of every layer is cut into units of about equal cost (the
 * cut depends on the layer alone, so a layer's bits do not depend on its companions and equal sp_conv2d_wgrad's), all units of
 * the layers that share a dW tile shape run as ONE launch, every unit leaves its partial tile in `workspace`, and one fold
 * launch sums each layer's partials in index order into `dw`: deterministic, no float atomics.
 * workspace: sp_conv2d_wgrad_workspace(jobs, n_jobs, &bytes) bytes (the error message of a short one states the need). */
typedef struct sp_wgrad_job {
    sp_conv_desc desc;
    const void* g;   /* [rows, g_channels] */
    const void* a;   /* NHWC tensor gathered through desc */
    float* dw;
    int64_t dst_stride_n, dst_stride_c;
    int32_t g_channels, n_valid, c_valid, kw_valid;
} sp_wgrad_job;
int sp_conv2d_wgrad_workspace(const sp_wgrad_job* jobs, int n_jobs, int64_t* bytes);
int sp_conv2d_wgrad_batched(const sp_wgrad_job* jobs, int n_jobs, void* workspace, int64_t workspace_bytes, void* stream);
/* one layer: sp_conv2d_wgrad_batched with a single job */
int sp_conv2d_wgrad(const sp_conv_desc* desc, const void* g, int g_channels, const void* a, int n_valid, int c_valid,
                    int kw_valid, int64_t dst_stride_n, int64_t dst_stride_c, float* dw, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* generic 4-D gather-copy used to (re)pack weights after every optimizer step:
 * dst[dst_offset + ((i0*d1 + i1)*d2 + i2)*d3 + i3] = all(i_k < valid[k]) ? src[src_base + sum i_k*src_strides[k]] : 0 */
int sp_permute4_f32(const float* src, void* dst, int dst_bf16, const int32_t* dst_dims, const int64_t* src_strides,
                    const int32_t* valid, int64_t src_base, int64_t dst_offset, void* stream);

/* every pack job of a network in one launch: `jobs_device` = device array of n_jobs records
 *   { int32 dst_dims[4]; int64 src_strides[4]; int32 valid[4]; int64 src_base; int64 dst_address; int64 total; int32 dst_bf16; int32 walk;
 *     int32 tap0; int32 tile0; }
 * (dst_address = device pointer of the destination incl. its offset), blocks_per_job workgroups grid-stride over a job.
 * walk: how the job's elements are visited (same result, other access pattern): 0 destination order (sources contiguous along the
 * last destination index), 1 multi-tap filters - a thread takes one (i0, i3) pair and loops over the (i1, i2) taps, whose sources
 * are one contiguous run, 2 a 2-D transpose (dims 1 and 2 of extent 1, source contiguous along i0) in 32x32 tiles through LDS,
 * 3 multi-tap filters whose fastest destination index is the source's slowest (stride[0] = floats per (i0, i3) block < |stride[3]|): tiles of
 * 32 i3 x tile0 i0 values through LDS, coalesced both ways; the record then ends in two more int32: tap0 (offset <= 0 from `base` to the
 * block's first tap) and tile0 (32 * (tile0 * stride[0] | 1) <= 10240).  Record size: 104 bytes (ABI 33; 96 before).
 * Walk 3 needs a 40 KB LDS tile: it exists only in sp_permute4_batched_tiled (ABI 34: same arguments, same results; a launch of its own so
 * that the default repack keeps 8 workgroups per CU); sp_permute4_batched visits a walk-3 job as walk 1. */
int sp_permute4_batched(const float* src, const void* jobs_device, int n_jobs, int blocks_per_job, void* stream);
int sp_permute4_batched_tiled(const float* src, const void* jobs_device, int n_jobs, int blocks_per_job, void* stream);

/* measurement aid: occupies `stream` for `us` microseconds (one idle wave on the 100 MHz constant clock) - bench.py's stand-in for the
 * latency of a SyncBatchNorm message (ddp...:89-90) on a box with a single GPU */
/* Backward of sp_upsample_add_nhwc (HRNet fuse layers, nets/pose_hrnet.py:192-202,250-257): dr = y_relu_src ? dy (y > 0) : dy (y in the
 * activation dtype, null when the forward had no ReLU); dbase (+)= dr at the high resolution [batch, h*factor, w*factor, c]; dx (+)= the sum
 * of dr over each factor x factor block [batch, h, w, c].  Gradients are fp32; the accumulate flags say whether the target already holds other
 * consumers' shares. */
int sp_upsample_add_bwd_nhwc(const float* dy, int bf16, const void* y_relu_src, int batch, int h, int w, int c, int factor, float* dbase,
                             int dbase_accumulate, float* dx, int dx_accumulate, void* stream);

/* SELayer backward (nets/commons.py:4-18 inside Bottleneck.forward, pose_resnet_dconv.py:124-131; `bf16`: dtype of the saved activations,
 * gradients w.r.t. activations are fp32):  y = relu(u * sigmoid(g[b,c]) + identity), u = bn3(conv3(.)), g = fc2(relu(fc0(mean_hw u))).
 *   sp_se_gate_bwd_reduce: da[b,c] = sum_hw (y > 0 ? dy : 0) * u
 *   sp_se_sigmoid_bwd:     dg = da * a (1 - a), a = sigmoid(g) (written in the activation dtype: the FC layers' MFMA operand); dbias[c] = sum_b dg
 *   sp_relu_bwd_rows:      out = dh (h > 0) on a [batch, c] map, dbias[c] = sum_b out (dbias may be null)
 *   sp_se_gate_bwd_apply:  du = (y > 0 ? dy : 0) * a + ds[b,c] / hw;  dres (+)= (y > 0 ? dy : 0)
 * The FC layers themselves run as 1x1 convolutions (sp_conv2d_fwd / dgrad re-packing / sp_conv2d_wgrad_batched). */
int sp_se_gate_bwd_reduce(const float* dy, int bf16, const void* y, const void* u, int batch, int hw, int c, float* da, void* stream);
int sp_se_sigmoid_bwd(const float* da, int bf16, const void* gate_logits, int batch, int c, void* dg, float* dbias, void* stream);
int sp_relu_bwd_rows(const float* dh, int bf16, const void* h, int batch, int c, void* out, float* dbias, void* stream);
int sp_se_gate_bwd_apply(const float* dy, int bf16, const void* y, const void* gate_logits, const float* ds, int batch, int hw, int c,
                         float* du, float* dres, int dres_accumulate, void* stream);

int sp_stream_delay_us(double us, void* stream);

/* RCCL called directly, for the collectives that sit on the train step's critical path (replaces the `SyncBatchNorm` exchanges and the
 * DistributedDataParallel gradient all-reduces of ddp...:89-93 issued through torch.distributed): sp_comm_allreduce_sum_f32 enqueues ONE
 * in-place fp32 SUM all-reduce on the caller's stream - stream order is the dependency, one host call per message.  librccl is resolved
 * at run time (sp_comm_available() == 0 where it is absent; everything single-GPU still works).  sp_comm_unique_id: rank 0 makes the 128-byte
 * id, the caller ships it to the other ranks (any channel), every rank calls sp_comm_create on its own device. */
int sp_comm_available(void);
int sp_comm_unique_id(void* id128);
int sp_comm_create(const void* id128, int world, int rank, void** comm);
int sp_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
/* the same for fp64 buffers (the SyncBatchNorm forward message: per-channel (sum, sum of squares) in double); n = number of doubles */
int sp_comm_allreduce_sum_f64(void* comm, double* buf, int64_t n, void* stream);
/* what RCCL reports for a communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice */
int sp_comm_info(void* comm, int* world, int* rank, int* device);
int sp_comm_destroy(void* comm);

/* ---- parameter packing: the reference's tensors -> what the launches above read ------------------------------------------
 * (all pointers are device memory; every call fills its whole destination, padding included)
 *
 * sp_conv_packed_dims: rows / depth of a packed weight matrix: n_pad = c_out rounded up to the widest tile that fits (128, 64 or
 * 32), k_pad = k rounded up to one K tile (32 fp32 / 64 bf16 elements = 128 bytes). */
int sp_conv_packed_dims(int c_out, int k, int bf16, int* n_pad, int* k_pad);
/* nn.Conv2d weight [c_out, c_in, kh, kw] fp32 (nets/pose_resnet_dconv.py:19-27) -> w_packed [n_pad][k_pad] (fp32, or bf16 when
 * dst_bf16), K ordered (ty, tx, c) with c_in_packed >= c_in channels per tap and taps_w_packed >= kw taps per row (the stem's
 * 3-channel 7x7 filter on the NHWC4 image: c_in_packed 4, taps_w_packed 8).  pixel_shuffle: rows in sub-pixel-major order
 * n' = sub*(c_out/4) + c <- channel 4*c + sub, what SP_CONV_PIXEL_SHUFFLE expects (nets/commons.py:36-41).  pair_s0 >= 0: the bf16
 * stem's x-paired form (sp_conv_desc.stride_x): packed channel = sub*4 + c of pixel pair tx, i.e. kx = 2*tx + sub - pair_s0,
 * c_in_packed = 8; pair_s0 = -1: plain. */
int sp_pack_conv_weights(const float* w, int c_out, int c_in, int kh, int kw, int c_in_packed, int taps_w_packed, int pixel_shuffle,
                         int pair_s0, int n_pad, int k_pad, void* dst, int dst_bf16, void* stream);
/* nn.ConvTranspose2d(k=4, s=2, p=1) weight [c_in, c_out, 4, 4] (nets/pose_resnet_dconv.py:236-244) -> [4 phases][n_pad][4*c_in]:
 * phase (py,px) holds the 2x2 taps W[:, :, 2ty+1-py, 2tx+1-px] that reach output pixels (2y+py, 2x+px) - no multiply is spent
 * on the zeros a transposed conv inserts */
/* nn.Conv2d(groups = g) weight [c_out][c_in / g][kh][kw] with c_out == c_in -> block-diagonal panels [c_out][kh * kw * panel] (K = (tap, channel
 * within the panel); row n belongs to panel n / panel and holds its group's weights at the group's channels inside the panel, zeros elsewhere).
 * `panel` = sp_conv_desc.c_in_group = tile_n of the launch: a multiple of c_in / g and of the K tile (32 fp32 / 64 bf16 elements). */
int sp_pack_conv_weights_grouped(const float* w, int c_out, int groups, int kh, int kw, int panel, void* dst, int dst_bf16, void* stream);
/* (ABI 34) the same block-diagonal panels for the TRAIN step of that grouped conv (loss.backward() through nets/pose_resnet_dconv.py:101): a tap subset
 * (ky, kx) = (ky0 + ky_step * ty, kx0 + kx_step * tx), ty < taps_h, tx < taps_w, of the kh x kw filter (taps outside it are zero) - the forward
 * (0, 1, 0, 1), the flipped taps of the stride-1 input gradient (kh-1, -1, kw-1, -1), one output phase of the stride-2 input gradient (ky0, 2, kx0, 2) -
 * and, with `transpose`, rows = the layer's INPUT channels: the input gradient is a grouped conv of dz with Wd[c][o_local][tap] = W[o][c_local][tap].
 * dst: [c][taps_h * taps_w * panel]. */
int sp_pack_conv_weights_grouped_taps(const float* w, int c, int groups, int kh, int kw, int transpose, int taps_h, int taps_w, int ky0, int ky_step,
                                      int kx0, int kx_step, int panel, void* dst, int dst_bf16, void* stream);
/* (ABI 34) weight gradient of that grouped conv (c_out == c_in = c, `groups` groups): dw[o][cl][ky][kx] (fp32, the nn.Conv2d weight layout, WRITTEN) =
 * sum over (b, oy, ox) of dz[b, oy, ox, o] * x[b, oy*stride - pad + ky, ox*stride - pad + kx, group(o) * (c / groups) + cl]; x and dz NHWC, fp32 or
 * (bf16 = 1) bf16.  A streaming reduction (1 / groups of the dense layer's FLOPs): partial sums per row chunk in `workspace`, folded in a fixed
 * order in fp64 (no atomics).  c / groups must divide 256, kh * kw <= 9. */
int sp_conv2d_wgrad_grouped_workspace(int batch, int out_h, int c, int groups, int kh, int kw, int64_t* bytes);
int sp_conv2d_wgrad_grouped(const void* x, const void* dz, int bf16, int batch, int in_h, int in_w, int out_h, int out_w, int c, int groups, int kh,
                            int kw, int stride, int pad, float* dw, void* workspace, int64_t workspace_bytes, void* stream);
int sp_pack_deconv_k4s2p1(const float* w, int c_in, int c_out, int n_pad, void* dst, int dst_bf16, void* stream);
/* eval-mode nn.BatchNorm2d as the conv epilogue's (scale, shift): scale = weight / sqrt(running_var + eps),
 * shift = bias - running_mean * scale (each operation rounded on its own, as the torch expressions); weight / bias NULL = 1 / 0;
 * pixel_shuffle: outputs in the packed row order of sp_pack_conv_weights */
int sp_fold_bn(const float* weight, const float* bias, const float* running_mean, const float* running_var, int c, float eps,
               int pixel_shuffle, float* scale, float* shift, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLE_POSE_HIP_H */
