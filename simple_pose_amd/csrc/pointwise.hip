// pointwise.hip - HBM-bound layout / pooling / fuse / loss kernels (NHWC fp32, 16 B per lane).
#include "sp_common.h"

namespace {

// [B,C,H,W] (C <= 4) -> [B,H,W,4], zero-filled tail channels.  Reads are coalesced along W per plane.
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ x, f32x4* __restrict__ y, int C, int hw, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw;
        const int pix = (int)(i - b * hw);
        const float* src = x + b * C * hw + pix;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        v[0] = src[0];
        if (C > 1) v[1] = src[hw];
        if (C > 2) v[2] = src[2 * (long long)hw];
        if (C > 3) v[3] = src[3 * (long long)hw];
        y[i] = v;
    }
}

__device__ __forceinline__ float pmax(float m, float v) { return (v > m || v != v) ? v : m; }  // NaN propagates like torch

// nn.MaxPool2d(3,2,1), NHWC, one lane = 4 channels of one output pixel
__global__ void maxpool3x3s2_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int H, int W, int C4, int Ho, int Wo,
                                    long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = x[((b * H + iy) * W + ix) * C4 + c];
                m[0] = pmax(m[0], v[0]); m[1] = pmax(m[1], v[1]); m[2] = pmax(m[2], v[2]); m[3] = pmax(m[3], v[3]);
            }
        }
        y[i] = m;
    }
}

// y[b, Y, X, :] = base[b, Y, X, :] + x[b, Y/f, X/f, :]  (nearest upsample + add [+ relu]); base may alias y
__global__ void upsample_add_kernel(const f32x4* __restrict__ x, const f32x4* base, f32x4* y, int h, int w, int C4, int f, int relu,
                                    long long total) {
    const int W = w * f, H = h * f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const long long b = r / H;
        const f32x4 a = x[((b * h + Y / f) * w + X / f) * C4 + c];
        f32x4 v = base[i];
        v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3];
        if (relu) {
            v[0] = v[0] > 0.f ? v[0] : 0.f; v[1] = v[1] > 0.f ? v[1] : 0.f;
            v[2] = v[2] > 0.f ? v[2] : 0.f; v[3] = v[3] > 0.f ? v[3] : 0.f;
        }
        y[i] = v;
    }
}

// nn.PixelShuffle(2): one lane gathers 4 output channels (stride 4 in the source pixel) and stores 16 B
__global__ void pixel_shuffle2_kernel(const float* __restrict__ x, f32x4* __restrict__ y, int h, int w, int C, long long total) {
    const int Co = C >> 2, Co4 = Co >> 2, W2 = 2 * w, H2 = 2 * h;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % Co4);
        long long r = i / Co4;
        const int X = (int)(r % W2); r /= W2;
        const int Y = (int)(r % H2);
        const long long b = r / H2;
        const int sub = ((Y & 1) << 1) | (X & 1);
        const float* src = x + ((b * h + (Y >> 1)) * w + (X >> 1)) * C + (k4 << 4) + sub;
        f32x4 v = {src[0], src[4], src[8], src[12]};
        y[i] = v;
    }
}

// backward of nn.PixelShuffle(2) (a permutation): one lane reads 16 B of dy (4 channels of one output pixel) and scatters them to
// stride-4 channels of the source pixel
__global__ void pixel_unshuffle2_kernel(const f32x4* __restrict__ dy, float* __restrict__ dx, int h, int w, int C, long long total) {
    const int Co = C >> 2, Co4 = Co >> 2, W2 = 2 * w, H2 = 2 * h;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % Co4);
        long long r = i / Co4;
        const int X = (int)(r % W2); r /= W2;
        const int Y = (int)(r % H2);
        const long long b = r / H2;
        const int sub = ((Y & 1) << 1) | (X & 1);
        float* dst = dx + ((b * h + (Y >> 1)) * w + (X >> 1)) * C + (k4 << 4) + sub;
        const f32x4 v = dy[i];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
}

// SELayer squeeze (nets/commons.py:8,15): y[b, c] = mean over the HW pixels of x[b, :, c].  One workgroup per (b, 256-channel
// slab... up to 64 float4 lanes x 4 pixel stripes), double accumulation, fixed order.
__global__ __launch_bounds__(256) void global_avg_pool_kernel(const f32x4* __restrict__ x, float* __restrict__ y, int HW, int C4) {
    const int b = blockIdx.y;
    const int lanes_c = C4 < 64 ? C4 : 64;
    const int stripes = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, ts = threadIdx.x / lanes_c;
    const int c4 = blockIdx.x * lanes_c + tc;
    __shared__ double sm[256 * 4];
    double acc[4] = {0, 0, 0, 0};
    if (c4 < C4 && ts < stripes)
        for (int p = ts; p < HW; p += stripes) {
            const f32x4 v = x[((size_t)b * HW + p) * C4 + c4];
            acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[threadIdx.x * 4 + e] = acc[e];
    __syncthreads();
    if (ts == 0 && c4 < C4) {
        for (int k = 1; k < stripes; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += sm[(k * lanes_c + tc) * 4 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[((size_t)b * C4 + c4) * 4 + e] = (float)(acc[e] / (double)HW);
    }
}

// SELayer excite + block tail: y = relu(x * sigmoid(g[b, c]) + identity)   (nets/commons.py:17-18, pose_resnet_dconv.py:126-131)
__global__ void se_gate_add_relu_kernel(const f32x4* __restrict__ x, const float* __restrict__ g, const f32x4* __restrict__ idn,
                                        f32x4* __restrict__ y, int HW, int C4, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long b = i / ((long long)HW * C4);
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[b * C4 + c4];
        const f32x4 v = x[i], r = idn[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = 1.f / (1.f + expf(-gv[e]));
            const float t = v[e] * sg + r[e];
            o[e] = t > 0.f ? t : 0.f;
        }
        y[i] = o;
    }
}

// datasets/coco.py:136 collate normalisation on the GPU: BGR u8 HWC -> RGB fp32 NCHW, x/255 - mean[c] (no std division)
__global__ void u8hwc_bgr_to_nchw_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int hw, float m0, float m1, float m2,
                                         long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw;
        const int pix = (int)(i - b * hw);
        const unsigned char* p = img + i * 3;
        float* o = out + b * 3 * hw + pix;
        o[0] = (float)p[2] / 255.0f - m0;
        o[hw] = (float)p[1] / 255.0f - m1;
        o[2 * (long long)hw] = (float)p[0] / 255.0f - m2;
    }
}

// The same normalisation straight into the network's input layout: BGR u8 HWC -> RGB NHWC4 fp32 / NHWC8 bf16 (pad channels 0).
// One pass instead of normalise (NCHW fp32) + layout change: 3 B read, 16 B written per pixel.
template <int BF16OUT>   // 0: NHWC4 fp32, 1: NHWC8 bf16, 2: NHWC4 bf16
__global__ void u8hwc_bgr_to_nhwc_kernel(const unsigned char* __restrict__ img, u32x4* __restrict__ out, float m0, float m1, float m2,
                                         long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const unsigned char* p = img + i * 3;
        const float r = (float)p[2] / 255.0f - m0, g = (float)p[1] / 255.0f - m1, b = (float)p[0] / 255.0f - m2;
        if constexpr (BF16OUT == 2) {
            typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
            const bf16x4_ v = {(__bf16)r, (__bf16)g, (__bf16)b, (__bf16)0.f};
            reinterpret_cast<unsigned long long*>(out)[i] = __builtin_bit_cast(unsigned long long, v);
        } else if constexpr (BF16OUT == 1) {
            typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
            bf16x8_ v = {(__bf16)r, (__bf16)g, (__bf16)b, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            out[i] = __builtin_bit_cast(u32x4, v);
        } else {
            const f32x4 v = {r, g, b, 0.f};
            out[i] = __builtin_bit_cast(u32x4, v);
        }
    }
}

// metrics/pose_metrics.py:212-245 HeatMapAcc on arg-max coordinates: per joint, share of valid samples (label x,y > 1) whose
// normalised distance |pred - label| / (W/f, H/f) is below the threshold; mean over joints that have a valid sample.
__global__ __launch_bounds__(256) void heat_map_acc_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                           const float* __restrict__ mask, int B, int J, float nx, float ny, float thresh,
                                                           float* __restrict__ acc) {
    __shared__ float jacc[256];
    __shared__ int jok[256];
    for (int j = threadIdx.x; j < J; j += 256) {
        int valid = 0, hit = 0;
        for (int b = 0; b < B; ++b) {
            // ddp...:130-131 feeds maps multiplied by the joint mask: a zero mask zeroes both maps -> both arg-max cells are (0,0)
            if (mask && mask[b * J + j] == 0.f) continue;
            const float lx = label[(b * J + j) * 2], ly = label[(b * J + j) * 2 + 1];
            if (lx > 1.f && ly > 1.f) {
                const float dx = pred[(b * J + j) * 2] / nx - lx / nx, dy = pred[(b * J + j) * 2 + 1] / ny - ly / ny;
                ++valid;
                if (sqrtf(dx * dx + dy * dy) < thresh) ++hit;
            }
        }
        jok[j] = valid > 0;
        jacc[j] = valid > 0 ? (float)hit / (float)valid : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        int cnt = 0;
        for (int j = 0; j < J; ++j) if (jok[j]) { s += jacc[j]; ++cnt; }
        *acc = cnt > 0 ? s / (float)cnt : 0.f;
    }
}

// masked MSE: per-block double partial sums (deterministic), then one block folds them.
__global__ void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ mask,
                                   float* __restrict__ grad, int hw, long long total, double inv_n, double* __restrict__ part) {
    double acc = 0.0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float w = mask[i / hw];
        const float d = pred[i] * w - tgt[i] * w;
        acc += (double)d * (double)d;
        if (grad) grad[i] = (float)((double)d * (double)w * inv_n);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, SP_WAVE);
    __shared__ double wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ void mse_final_kernel(const double* __restrict__ part, int n, double inv_n, float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, SP_WAVE);
    if (threadIdx.x == 0) *loss = (float)(0.5 * acc * inv_n);
}

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));  // cap + grid-stride (guide, Guideline 11)
}

}  // namespace

extern "C" int sp_nchw_to_nhwc4(const float* x, float* y, int batch, int channels, int h, int w, void* stream) {
    SP_REQUIRE(x && y, "sp_nchw_to_nhwc4: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && channels >= 1 && channels <= 4, "sp_nchw_to_nhwc4: bad shape B=%d C=%d H=%d W=%d", batch, channels, h, w);
    const long long total = (long long)batch * h * w;
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_nchw_to_nhwc4: tensor too large");
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<f32x4*>(y), channels, h * w, total);
    return sp_check_launch("nchw_to_nhwc4_kernel");
}

extern "C" int sp_maxpool3x3s2_nhwc(const float* x, float* y, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_maxpool3x3s2_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_maxpool3x3s2_nhwc: bad shape (c %% 4 != 0?)");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * ho * wo * (c / 4);
    SP_REQUIRE((long long)batch * h * w * c < (1ll << 31), "sp_maxpool3x3s2_nhwc: tensor too large");
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<f32x4*>(y), h, w, c / 4, ho, wo, total);
    return sp_check_launch("maxpool3x3s2_kernel");
}

extern "C" int sp_pixel_shuffle2_nhwc(const float* x, float* y, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_pixel_shuffle2_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 16 == 0, "sp_pixel_shuffle2_nhwc: c=%d must be a multiple of 16", c);
    const long long total = (long long)batch * h * w * c / 4;
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_pixel_shuffle2_nhwc: tensor too large");
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<f32x4*>(y), h, w, c, total);
    return sp_check_launch("pixel_shuffle2_kernel");
}

extern "C" int sp_pixel_unshuffle2_nhwc(const float* dy, float* dx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(dy && dx, "sp_pixel_unshuffle2_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 16 == 0, "sp_pixel_unshuffle2_nhwc: c=%d must be a multiple of 16", c);
    const long long total = (long long)batch * h * w * c / 4;
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_pixel_unshuffle2_nhwc: tensor too large");
    hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(dy), dx, h, w, c, total);
    return sp_check_launch("pixel_unshuffle2_kernel");
}

extern "C" int sp_u8hwc_bgr_to_nchw_f32(const unsigned char* img, float* out, int batch, int h, int w, const float* mean_rgb_host, void* stream) {
    SP_REQUIRE(img && out && mean_rgb_host, "sp_u8hwc_bgr_to_nchw_f32: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0, "sp_u8hwc_bgr_to_nchw_f32: bad shape");
    const long long total = (long long)batch * h * w;
    hipLaunchKernelGGL(u8hwc_bgr_to_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, h * w,
                       mean_rgb_host[0], mean_rgb_host[1], mean_rgb_host[2], total);
    return sp_check_launch("u8hwc_bgr_to_nchw_kernel");
}

extern "C" int sp_u8hwc_bgr_to_nhwc(const unsigned char* img, void* out, int out_bf16, int batch, int h, int w, const float* mean_rgb_host,
                                    void* stream) {
    SP_REQUIRE(img && out && mean_rgb_host, "sp_u8hwc_bgr_to_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0, "sp_u8hwc_bgr_to_nhwc: bad shape");
    const long long total = (long long)batch * h * w;
    SP_REQUIRE(total * 16 < (1ll << 31), "sp_u8hwc_bgr_to_nhwc: tensor too large");
    SP_REQUIRE(out_bf16 >= 0 && out_bf16 <= 2 && (out_bf16 != 2 || w % 2 == 0), "sp_u8hwc_bgr_to_nhwc: out_bf16 must be 0, 1 or 2 (2: w even)");
#define SP_U8_LAUNCH(MODE) hipLaunchKernelGGL(u8hwc_bgr_to_nhwc_kernel<MODE>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, \
                                              reinterpret_cast<u32x4*>(out), mean_rgb_host[0], mean_rgb_host[1], mean_rgb_host[2], total)
    if (out_bf16 == 2) SP_U8_LAUNCH(2);
    else if (out_bf16 == 1) SP_U8_LAUNCH(1);
    else SP_U8_LAUNCH(0);
#undef SP_U8_LAUNCH
    return sp_check_launch("u8hwc_bgr_to_nhwc_kernel");
}

extern "C" int sp_heat_map_acc(const float* pred_coords, const float* label_coords, const float* mask, int batch, int joints, int h, int w,
                               float distance_thresh, float norm_frac, float* acc_out, void* stream) {
    SP_REQUIRE(pred_coords && label_coords && acc_out, "sp_heat_map_acc: null pointer");
    SP_REQUIRE(batch > 0 && joints > 0 && joints <= 256 && h > 0 && w > 0 && norm_frac > 0.f, "sp_heat_map_acc: bad argument (joints <= 256)");
    hipLaunchKernelGGL(heat_map_acc_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred_coords, label_coords, mask, batch, joints,
                       (float)w / norm_frac, (float)h / norm_frac, distance_thresh, acc_out);
    return sp_check_launch("heat_map_acc_kernel");
}

extern "C" int sp_global_avg_pool_nhwc(const float* x, float* y, int batch, int hw, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_global_avg_pool_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && hw > 0 && c > 0 && c % 4 == 0, "sp_global_avg_pool_nhwc: bad shape");
    const int c4 = c / 4, lanes = c4 < 64 ? c4 : 64;
    hipLaunchKernelGGL(global_avg_pool_kernel, dim3((c4 + lanes - 1) / lanes, batch), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), y, hw, c4);
    return sp_check_launch("global_avg_pool_kernel");
}

extern "C" int sp_se_gate_add_relu_nhwc(const float* x, const float* gate_logits, const float* identity, float* y, int batch, int hw, int c,
                                        void* stream) {
    SP_REQUIRE(x && gate_logits && identity && y, "sp_se_gate_add_relu_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && hw > 0 && c > 0 && c % 4 == 0, "sp_se_gate_add_relu_nhwc: bad shape");
    const long long total = (long long)batch * hw * (c / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_se_gate_add_relu_nhwc: tensor too large");
    hipLaunchKernelGGL(se_gate_add_relu_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), gate_logits, reinterpret_cast<const f32x4*>(identity), reinterpret_cast<f32x4*>(y), hw,
                       c / 4, total);
    return sp_check_launch("se_gate_add_relu_kernel");
}

extern "C" int sp_upsample_add_nhwc(const float* x, const float* base, float* y, int batch, int h, int w, int c, int factor, int relu,
                                    void* stream) {
    SP_REQUIRE(x && base && y, "sp_upsample_add_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && factor >= 1, "sp_upsample_add_nhwc: bad shape");
    const long long total = (long long)batch * h * factor * w * factor * (c / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_upsample_add_nhwc: tensor too large");
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4*>(x), reinterpret_cast<const f32x4*>(base), reinterpret_cast<f32x4*>(y), h, w, c / 4, factor,
                       relu, total);
    return sp_check_launch("upsample_add_kernel");
}

extern "C" int sp_masked_mse(const float* pred, const float* target, const float* mask, int batch, int joints, int hw,
                             float* loss_out, float* grad, void* workspace, void* stream) {
    SP_REQUIRE(pred && target && mask && loss_out && workspace, "sp_masked_mse: null pointer");
    SP_REQUIRE(batch > 0 && joints > 0 && hw > 0, "sp_masked_mse: bad shape");
    const long long total = (long long)batch * joints * hw;
    int g = grid_for(total, 256);
    if (g > 512) g = 512;  // workspace holds 512 doubles
    const double inv_n = 1.0 / (double)total;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, pred, target, mask, grad, hw, total, inv_n,
                       reinterpret_cast<double*>(workspace));
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const double*>(workspace), g, inv_n, loss_out);
    return sp_check_launch("mse kernels");
}
