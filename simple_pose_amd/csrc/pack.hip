// pack.hip - reference-layout parameters -> the layouts the conv kernels read (C ABI: sp_pack_conv_weights,
// sp_pack_deconv_k4s2p1, sp_fold_bn, sp_conv_packed_dims).
//
// The reference keeps nn.Conv2d weights as [O,I,kh,kw] (nets/pose_resnet_dconv.py:19-27), nn.ConvTranspose2d(4,2,1) weights as
// [I,O,4,4] (:236-244) and eval-mode nn.BatchNorm2d as four [C] vectors (:253,:259); the implicit GEMM wants [phases][n_pad][k_pad]
// with K ordered (tap_y, tap_x, channel) to match NHWC activations and BatchNorm as one (scale, shift) pair per channel.  One
// gather kernel per destination: every destination element computes where it comes from (or that it is padding), so stores are
// coalesced and each launch fills its whole buffer - nothing has to be zeroed first.
#include "sp_common.h"

namespace {

struct PackConv {
    int O, I, kh, kw;      // source [O][I][kh][kw]
    int ci, tw;            // packed channels per tap / taps per row
    int n_pad, k_pad;
    int shuffle;           // rows in sub-pixel-major order (fused PixelShuffle)
    int pair, s0;          // x-paired 4-channel image: packed channel = sub*4 + c, kx = 2*tx + sub - s0
};

template <bool BF16OUT>
__global__ void pack_conv_kernel(const float* __restrict__ w, void* __restrict__ dst, const PackConv p) {
    const long long total = (long long)p.n_pad * p.k_pad;
    const int kreal = p.kh * p.tw * p.ci;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / p.k_pad), k = (int)(i - (long long)n * p.k_pad);
        float v = 0.f;
        if (n < p.O && k < kreal) {
            const int c = k % p.ci, t = k / p.ci;
            const int tx = t % p.tw, ty = t / p.tw;
            int cc = c, kx = tx;
            if (p.pair) {
                const int sub = c >> 2;
                cc = c & 3;
                kx = 2 * tx + sub - p.s0;
            }
            const int o = p.shuffle ? (n % (p.O >> 2)) * 4 + n / (p.O >> 2) : n;
            if (cc < p.I && kx >= 0 && kx < p.kw) v = w[(((long long)o * p.I + cc) * p.kh + ty) * p.kw + kx];
        }
        if constexpr (BF16OUT) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

// grouped conv (c_out == c_in, `groups` groups of cpg channels): dst [O][taps][panel], row n's panel covers input channels
// [(n / panel) * panel, +panel); element (n, tap, cl) = W[n][c % cpg][tap] when channel c = (n / panel) * panel + cl lies in n's group, else 0
template <bool BF16OUT>
__global__ void pack_conv_grouped_kernel(const float* __restrict__ w, void* __restrict__ dst, int O, int cpg, int taps, int panel) {
    const int K = taps * panel;
    const long long total = (long long)O * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i - (long long)n * K);
        const int tap = k / panel, cl = k - tap * panel;
        const int c = (n / panel) * panel + cl;
        float v = 0.f;
        if (c / cpg == n / cpg) v = w[((long long)n * cpg + (c % cpg)) * taps + tap];
        if constexpr (BF16OUT) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

// ConvTranspose2d(k=4, s=2, p=1) weight [I][O][4][4] -> [4 phases][n_pad][4*I]: output pixel (2y+py, 2x+px) = sum over the 2x2 taps
// (ty,tx) of x[y+py-ty, x+px-tx] * W[:, :, 2ty+1-py, 2tx+1-px]
template <bool BF16OUT>
__global__ void pack_deconv_kernel(const float* __restrict__ w, void* __restrict__ dst, int I, int O, int n_pad) {
    const int K = 4 * I;
    const long long total = 4ll * n_pad * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long long r = i / K;
        const int n = (int)(r % n_pad), ph = (int)(r / n_pad);
        float v = 0.f;
        if (n < O) {
            const int c = k % I, t = k / I;
            const int ty = t >> 1, tx = t & 1, py = ph >> 1, px = ph & 1;
            v = w[(((long long)c * O + n) * 4 + (2 * ty + 1 - py)) * 4 + (2 * tx + 1 - px)];
        }
        if constexpr (BF16OUT) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

// eval-mode BatchNorm as y = x*scale + shift with ATen's factoring (alpha = w / sqrt(var + eps), beta = b - mean*alpha), each
// operation rounded on its own (no fused multiply-add): the same floats as the torch expressions this replaces
__global__ void fold_bn_kernel(const float* __restrict__ weight, const float* __restrict__ bias, const float* __restrict__ mean,
                               const float* __restrict__ var, int c, float eps, int shuffle, float* __restrict__ scale, float* __restrict__ shift) {
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c) return;
    const int o = shuffle ? (n % (c >> 2)) * 4 + n / (c >> 2) : n;
    // (sqrt through fp64: hipcc lowers the fp32 forms to the bare v_sqrt_f32 approximation, 1 ulp off the correctly rounded root
    // that ATen's CPU kernel divides by; the fp64 root of an fp32 value rounds to the correctly rounded fp32 root)
    const float invstd = 1.0f / (float)sqrt((double)(var[o] + eps));
    const float g = weight ? weight[o] : 1.0f;
    const float sc = g * invstd;
    const float prod = mean[o] * sc;
    scale[n] = sc;
    shift[n] = (bias ? bias[o] : 0.0f) - prod;
}

int grid_of(long long total) {
    const long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int sp_conv_packed_dims(int c_out, int k, int bf16, int* n_pad, int* k_pad) {
    SP_REQUIRE(c_out > 0 && k > 0 && n_pad && k_pad, "sp_conv_packed_dims: bad argument");
    const int km = bf16 ? 64 : 32;                       // one K tile = 128 bytes
    *k_pad = (k + km - 1) / km * km;
    *n_pad = c_out >= 128 ? (c_out + 127) / 128 * 128 : (c_out > 32 ? (c_out + 63) / 64 * 64 : 32);
    return SP_OK;
}

extern "C" int sp_pack_conv_weights(const float* w, int c_out, int c_in, int kh, int kw, int c_in_packed, int taps_w_packed, int pixel_shuffle,
                                    int pair_s0, int n_pad, int k_pad, void* dst, int dst_bf16, void* stream) {
    SP_REQUIRE(w && dst, "sp_pack_conv_weights: null pointer");
    SP_REQUIRE(c_out > 0 && c_in > 0 && kh > 0 && kw > 0 && c_in_packed > 0 && taps_w_packed > 0, "sp_pack_conv_weights: bad shape");
    PackConv p;
    p.O = c_out; p.I = c_in; p.kh = kh; p.kw = kw; p.ci = c_in_packed; p.tw = taps_w_packed; p.n_pad = n_pad; p.k_pad = k_pad;
    p.shuffle = pixel_shuffle ? 1 : 0;
    p.pair = pair_s0 >= 0 ? 1 : 0;
    p.s0 = pair_s0 >= 0 ? pair_s0 : 0;
    if (p.pair) {
        SP_REQUIRE(c_in <= 4 && c_in_packed == 8 && 2 * taps_w_packed - 1 - p.s0 >= kw - 1 && p.s0 <= 1,
                   "sp_pack_conv_weights: paired packing needs c_in <= 4, c_in_packed == 8 and taps_w_packed pairs that cover kw + s0 pixels");
    } else {
        SP_REQUIRE(c_in_packed >= c_in && taps_w_packed >= kw, "sp_pack_conv_weights: packed extents smaller than the weight's");
    }
    SP_REQUIRE(c_in_packed % (dst_bf16 ? 8 : 4) == 0, "sp_pack_conv_weights: c_in_packed must fill whole 16-byte chunks");
    SP_REQUIRE(n_pad >= c_out && n_pad % 32 == 0 && k_pad >= kh * taps_w_packed * c_in_packed && k_pad % (dst_bf16 ? 64 : 32) == 0,
               "sp_pack_conv_weights: n_pad / k_pad too small or not tile multiples (see sp_conv_packed_dims)");
    SP_REQUIRE(!pixel_shuffle || (c_out % 4 == 0 && n_pad == c_out), "sp_pack_conv_weights: PixelShuffle packing needs c_out %% 4 == 0 and n_pad == c_out");
    const long long total = (long long)n_pad * k_pad;
    if (dst_bf16) hipLaunchKernelGGL(pack_conv_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, p);
    else hipLaunchKernelGGL(pack_conv_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, p);
    return sp_check_launch("pack_conv_kernel");
}

extern "C" int sp_pack_conv_weights_grouped(const float* w, int c_out, int groups, int kh, int kw, int panel, void* dst, int dst_bf16, void* stream) {
    SP_REQUIRE(w && dst, "sp_pack_conv_weights_grouped: null pointer");
    SP_REQUIRE(c_out > 0 && groups > 0 && c_out % groups == 0 && kh > 0 && kw > 0, "sp_pack_conv_weights_grouped: bad shape");
    const int cpg = c_out / groups;
    SP_REQUIRE(panel > 0 && panel % cpg == 0 && c_out % panel == 0 && panel % (dst_bf16 ? 64 : 32) == 0,
               "sp_pack_conv_weights_grouped: panel=%d must be a multiple of the group size %d and of the K tile, and divide c_out=%d", panel, cpg, c_out);
    const long long total = (long long)c_out * kh * kw * panel;
    if (dst_bf16) hipLaunchKernelGGL(pack_conv_grouped_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c_out, cpg, kh * kw, panel);
    else hipLaunchKernelGGL(pack_conv_grouped_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c_out, cpg, kh * kw, panel);
    return sp_check_launch("pack_conv_grouped_kernel");
}

// The same block-diagonal panels for the TRAIN step of a grouped conv (round 6): a tap subset (ky, kx) = (ky0 + ky_step * ty, kx0 + kx_step * tx) of
// the kh x kw filter - the forward (0, 1, 0, 1), the flipped taps of the stride-1 input gradient (kh-1, -1, kw-1, -1), one output phase of the
// stride-2 input gradient (ky0, 2, kx0, 2) - and, with `transpose`, rows = the layer's INPUT channels (the input gradient is a grouped conv of dz
// with Wd[c][o_local][tap] = W[o][c_local][tap]): row n's panel covers channels [(n / panel) * panel, +panel) of the K operand; element (n, t, j) is
// W[o][cl][ky][kx] with (o, c) = (n, P*panel + j) [forward] or (P*panel + j, n) [transpose] when o and c share a group, cl = c % cpg; else 0.
template <bool BF16OUT>
__global__ void pack_conv_grouped_taps_kernel(const float* __restrict__ w, void* __restrict__ dst, int C, int cpg, int kh, int kw, int transpose, int th,
                                              int tw, int ky0, int ky_step, int kx0, int kx_step, int panel) {
    const int K = th * tw * panel;
    const long long total = (long long)C * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i - (long long)n * K);
        const int t = k / panel, j = k - t * panel;
        const int ty = t / tw, tx = t - ty * tw;
        const int ky = ky0 + ky_step * ty, kx = kx0 + kx_step * tx;
        const int m = (n / panel) * panel + j;
        const int o = transpose ? m : n, c = transpose ? n : m;
        float v = 0.f;
        if (o / cpg == c / cpg && (unsigned)ky < (unsigned)kh && (unsigned)kx < (unsigned)kw) v = w[(((long long)o * cpg + (c % cpg)) * kh + ky) * kw + kx];
        if constexpr (BF16OUT) reinterpret_cast<__bf16*>(dst)[i] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

extern "C" int sp_pack_conv_weights_grouped_taps(const float* w, int c, int groups, int kh, int kw, int transpose, int taps_h, int taps_w, int ky0,
                                                 int ky_step, int kx0, int kx_step, int panel, void* dst, int dst_bf16, void* stream) {
    SP_REQUIRE(w && dst, "sp_pack_conv_weights_grouped_taps: null pointer");
    SP_REQUIRE(c > 0 && groups > 0 && c % groups == 0 && kh > 0 && kw > 0 && taps_h > 0 && taps_w > 0, "sp_pack_conv_weights_grouped_taps: bad shape");
    const int cpg = c / groups;
    SP_REQUIRE(panel > 0 && panel % cpg == 0 && c % panel == 0 && panel % (dst_bf16 ? 64 : 32) == 0,
               "sp_pack_conv_weights_grouped_taps: panel=%d must be a multiple of the group size %d and of the K tile, and divide c=%d", panel, cpg, c);
    const long long total = (long long)c * taps_h * taps_w * panel;
    if (dst_bf16)
        hipLaunchKernelGGL(pack_conv_grouped_taps_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c, cpg, kh, kw, transpose ? 1 : 0,
                           taps_h, taps_w, ky0, ky_step, kx0, kx_step, panel);
    else
        hipLaunchKernelGGL(pack_conv_grouped_taps_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c, cpg, kh, kw, transpose ? 1 : 0,
                           taps_h, taps_w, ky0, ky_step, kx0, kx_step, panel);
    return sp_check_launch("pack_conv_grouped_taps_kernel");
}

extern "C" int sp_pack_deconv_k4s2p1(const float* w, int c_in, int c_out, int n_pad, void* dst, int dst_bf16, void* stream) {
    SP_REQUIRE(w && dst, "sp_pack_deconv_k4s2p1: null pointer");
    SP_REQUIRE(c_in > 0 && c_out > 0 && c_in % (dst_bf16 ? 16 : 8) == 0, "sp_pack_deconv_k4s2p1: 4*c_in must be a whole number of K tiles");
    SP_REQUIRE(n_pad >= c_out && n_pad % 32 == 0, "sp_pack_deconv_k4s2p1: bad n_pad");
    const long long total = 16ll * n_pad * c_in;
    if (dst_bf16) hipLaunchKernelGGL(pack_deconv_kernel<true>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c_in, c_out, n_pad);
    else hipLaunchKernelGGL(pack_deconv_kernel<false>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, w, dst, c_in, c_out, n_pad);
    return sp_check_launch("pack_deconv_kernel");
}

extern "C" int sp_fold_bn(const float* weight, const float* bias, const float* running_mean, const float* running_var, int c, float eps,
                          int pixel_shuffle, float* scale, float* shift, void* stream) {
    SP_REQUIRE(running_mean && running_var && scale && shift && c > 0, "sp_fold_bn: bad argument");
    SP_REQUIRE(!pixel_shuffle || c % 4 == 0, "sp_fold_bn: PixelShuffle order needs c %% 4 == 0");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, bias, running_mean, running_var, c, eps,
                       pixel_shuffle ? 1 : 0, scale, shift);
    return sp_check_launch("fold_bn_kernel");
}
