// pointwise_bf16.hip - bf16 NHWC variants of the HBM-bound layout / pooling / fuse kernels (16 B = 8 channels per lane).
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// fp32 NCHW [B,C,H,W] (C <= 8) -> bf16 NHWC8 [B,H,W,8], zero tail channels
__global__ void nchw_to_nhwc8_bf16_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int C, int hw, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw;
        const int pix = (int)(i - b * hw);
        const float* src = x + b * C * hw + pix;
        bf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (__bf16)(c < C ? src[(long long)c * hw] : 0.f);
        y[i] = __builtin_bit_cast(u32x4, v);
    }
}

// [B,C<=4,H,W] fp32 -> [B,H,W,4] bf16: the bf16 stem reads two neighbouring pixels as one 8-channel "pair pixel"
__global__ void nchw_to_nhwc4_bf16_kernel(const float* __restrict__ x, unsigned long long* __restrict__ y, int C, int hw, long long total) {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw;
        const int pix = (int)(i - b * hw);
        const float* src = x + b * C * hw + pix;
        bf16x4_ v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (__bf16)(c < C ? src[(long long)c * hw] : 0.f);
        y[i] = __builtin_bit_cast(unsigned long long, v);
    }
}

__device__ __forceinline__ float pmax(float m, float v) { return (v > m || v != v) ? v : m; }

__global__ void maxpool3x3s2_bf16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int H, int W, int C8, int Ho, int Wo,
                                         long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long long r = i / C8;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -__builtin_inff();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const bf16x8 v = __builtin_bit_cast(bf16x8, x[((b * H + iy) * W + ix) * C8 + c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = pmax(m[e], (float)v[e]);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)m[e];
        y[i] = __builtin_bit_cast(u32x4, o);
    }
}

// nn.PixelShuffle(2): y[b,2Y+i,2X+j,k] = x[b,Y,X,4k+2i+j]; one lane = 8 output channels
__global__ void pixel_shuffle2_bf16_kernel(const __bf16* __restrict__ x, u32x4* __restrict__ y, int h, int w, int C, long long total) {
    const int Co8 = C >> 5, W2 = 2 * w, H2 = 2 * h;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % Co8);
        long long r = i / Co8;
        const int X = (int)(r % W2); r /= W2;
        const int Y = (int)(r % H2);
        const long long b = r / H2;
        const int sub = ((Y & 1) << 1) | (X & 1);
        const __bf16* src = x + ((b * h + (Y >> 1)) * w + (X >> 1)) * C + (k8 << 5) + sub;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[4 * e];
        y[i] = __builtin_bit_cast(u32x4, v);
    }
}

// backward of nn.PixelShuffle(2) on a bf16 gradient (PoseTrainer grad_dtype "bf16", DUC head): gather form - one lane owns 8 consecutive
// channels of a source pixel = output channels k0, k0 + 1 of its four sub-pixels (source channel 4 k + sub), reads four 4-byte pairs and
// stores 16 bytes
__global__ void pixel_unshuffle2_bf16_kernel(const unsigned int* __restrict__ dy, u32x4* __restrict__ dx, int h, int w, int C, long long total) {
    const int C8 = C >> 3, Co = C >> 2, W2 = 2 * w;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long r = i / C8;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const long long b = r / h;
        const int k0 = c8 << 1;                       // output channels k0, k0 + 1
        unsigned int pr[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const long long o = ((b * 2 * h + (2 * y + (sub >> 1))) * W2 + (2 * x + (sub & 1))) * Co + k0;
            pr[sub] = dy[o >> 1];                     // (k0 is even: the pair is 4-byte aligned)
        }
        u32x4 v;                                      // source order: (k0,0) (k0,1) (k0,2) (k0,3) (k0+1,0) ... (k0+1,3)
        v[0] = (pr[0] & 0xffffu) | (pr[1] << 16);
        v[1] = (pr[2] & 0xffffu) | (pr[3] << 16);
        v[2] = (pr[0] >> 16) | (pr[1] & 0xffff0000u);
        v[3] = (pr[2] >> 16) | (pr[3] & 0xffff0000u);
        dx[i] = v;
    }
}

__global__ void upsample_add_bf16_kernel(const u32x4* __restrict__ x, const u32x4* base, u32x4* y, int h, int w, int C8, int f, int relu,
                                         long long total) {
    const int W = w * f, H = h * f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long long r = i / C8;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const long long b = r / H;
        const bf16x8 a = __builtin_bit_cast(bf16x8, x[((b * h + Y / f) * w + X / f) * C8 + c]);
        const bf16x8 v = __builtin_bit_cast(bf16x8, base[i]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (float)v[e] + (float)a[e];
            if (relu) t = t > 0.f ? t : 0.f;
            o[e] = (__bf16)t;
        }
        y[i] = __builtin_bit_cast(u32x4, o);
    }
}

// HRNet fuse stage, all upsampled terms of one output in ONE pass (round 4):  y = [relu]( ((base + up(x0, f0)) + up(x1, f1)) + up(x2, f2) )
// - `y = y + fuse_layers[i][j](x[j])` for j >= i of HighResolutionModule.forward (pose_hrnet.py:250-257), f = 1 being the identity term.
// The chained form ran one launch per term, each reading and re-writing the high-resolution sum (and, in bf16, rounding it every time);
// here base and every term are read once, the sum is formed in fp32 in the reference's order and rounded once.  fp32: same bits as the chain.
struct UpTerms {
    const void* x[3];
    int h[3], w[3], f[3];
    int n;
};
template <bool BF16>
__global__ void upsample_add_n_kernel(const void* __restrict__ base, const UpTerms t, void* __restrict__ y, int H, int W, int CV, int relu,
                                      long long total) {
    constexpr int V = BF16 ? 8 : 4;                     // channels per 16-byte vector
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV);
        long long r = i / CV;
        const int X = (int)(r % W); r /= W;
        const int Y = (int)(r % H);
        const long long b = r / H;
        float v[V];
        u32x4 raw[4];
        raw[0] = reinterpret_cast<const u32x4*>(base)[i];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < t.n) raw[k + 1] = reinterpret_cast<const u32x4*>(t.x[k])[((b * t.h[k] + Y / t.f[k]) * t.w[k] + X / t.f[k]) * CV + c];
        auto unpack = [&](const u32x4 q, float* o) {
            if constexpr (BF16) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, q);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
            } else {
                const f32x4 a = __builtin_bit_cast(f32x4, q);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = a[e];
            }
        };
        unpack(raw[0], v);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < t.n) {
                float a[V];
                unpack(raw[k + 1], a);
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] += a[e];
            }
        if (relu) {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if constexpr (BF16) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            reinterpret_cast<u32x4*>(y)[i] = __builtin_bit_cast(u32x4, o);
        } else {
            reinterpret_cast<f32x4*>(y)[i] = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// SELayer in bf16 (nets/commons.py:4-18): squeeze = mean over the pixels of one image, 8 channels per lane, fp64 sums, bf16 result (the
// two FCs run as bf16 1x1 convolutions on the [B,1,1,C] tensor)
__global__ __launch_bounds__(256) void global_avg_pool_bf16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int HW, int C8) {
    const int b = blockIdx.y;
    const int lanes_c = C8 < 64 ? C8 : 64;
    const int stripes = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, ts = threadIdx.x / lanes_c;
    const int c8 = blockIdx.x * lanes_c + tc;
    __shared__ double sm[256 * 8];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < C8 && ts < stripes)
        for (int p = ts; p < HW; p += stripes) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, x[((size_t)b * HW + p) * C8 + c8]);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (double)(float)v[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    if (ts == 0 && c8 < C8) {
        for (int k = 1; k < stripes; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += sm[(k * lanes_c + tc) * 8 + e];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(float)(acc[e] / (double)HW);
        y[(size_t)b * C8 + c8] = __builtin_bit_cast(u32x4, o);
    }
}

// excite + block tail: y = relu(x * sigmoid(gate[b, c]) + identity), fp32 arithmetic on bf16 tensors
__global__ void se_gate_add_relu_bf16_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ g, const u32x4* __restrict__ idn,
                                             u32x4* __restrict__ y, int HW, int C8, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long long b = i / ((long long)HW * C8);
        const bf16x8 gv = __builtin_bit_cast(bf16x8, g[b * C8 + c8]);
        const bf16x8 v = __builtin_bit_cast(bf16x8, x[i]), r = __builtin_bit_cast(bf16x8, idn[i]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = 1.f / (1.f + expf(-(float)gv[e]));
            const float t = (float)v[e] * sg + (float)r[e];
            o[e] = (__bf16)(t > 0.f ? t : 0.f);
        }
        y[i] = __builtin_bit_cast(u32x4, o);
    }
}

}  // namespace

extern "C" int sp_global_avg_pool_nhwc_bf16(const void* x, void* y, int batch, int hw, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_global_avg_pool_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && hw > 0 && c > 0 && c % 8 == 0, "sp_global_avg_pool_nhwc_bf16: bad shape");
    const int c8 = c / 8, lanes = c8 < 64 ? c8 : 64;
    hipLaunchKernelGGL(global_avg_pool_bf16_kernel, dim3((c8 + lanes - 1) / lanes, batch), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), hw, c8);
    return sp_check_launch("global_avg_pool_bf16_kernel");
}

extern "C" int sp_se_gate_add_relu_nhwc_bf16(const void* x, const void* gate_logits, const void* identity, void* y, int batch, int hw, int c,
                                             void* stream) {
    SP_REQUIRE(x && gate_logits && identity && y, "sp_se_gate_add_relu_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && hw > 0 && c > 0 && c % 8 == 0, "sp_se_gate_add_relu_nhwc_bf16: bad shape");
    const long long total = (long long)batch * hw * (c / 8);
    SP_REQUIRE(total * 16 < (1ll << 32), "sp_se_gate_add_relu_nhwc_bf16: tensor too large");
    hipLaunchKernelGGL(se_gate_add_relu_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(gate_logits), reinterpret_cast<const u32x4*>(identity),
                       reinterpret_cast<u32x4*>(y), hw, c / 8, total);
    return sp_check_launch("se_gate_add_relu_bf16_kernel");
}

extern "C" int sp_nchw_to_nhwc8_bf16(const float* x, void* y, int batch, int channels, int h, int w, void* stream) {
    SP_REQUIRE(x && y, "sp_nchw_to_nhwc8_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && channels >= 1 && channels <= 8, "sp_nchw_to_nhwc8_bf16: bad shape");
    const long long total = (long long)batch * h * w;
    SP_REQUIRE(total * 8 < (1ll << 30), "sp_nchw_to_nhwc8_bf16: tensor too large");
    hipLaunchKernelGGL(nchw_to_nhwc8_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<u32x4*>(y),
                       channels, h * w, total);
    return sp_check_launch("nchw_to_nhwc8_bf16_kernel");
}

extern "C" int sp_nchw_to_nhwc4_bf16(const float* x, void* y, int batch, int channels, int h, int w, void* stream) {
    SP_REQUIRE(x && y, "sp_nchw_to_nhwc4_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && w % 2 == 0 && channels >= 1 && channels <= 4, "sp_nchw_to_nhwc4_bf16: bad shape B=%d C=%d H=%d W=%d (W even)",
               batch, channels, h, w);
    const long long total = (long long)batch * h * w;
    SP_REQUIRE(total * 8 < (1ll << 31), "sp_nchw_to_nhwc4_bf16: tensor too large");
    hipLaunchKernelGGL(nchw_to_nhwc4_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<unsigned long long*>(y), channels, h * w, total);
    return sp_check_launch("nchw_to_nhwc4_bf16_kernel");
}

extern "C" int sp_maxpool3x3s2_nhwc_bf16(const void* x, void* y, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_maxpool3x3s2_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, "sp_maxpool3x3s2_nhwc_bf16: bad shape (c %% 8 != 0?)");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * ho * wo * (c / 8);
    SP_REQUIRE((long long)batch * h * w * c < (1ll << 30), "sp_maxpool3x3s2_nhwc_bf16: tensor too large");
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), h, w, c / 8, ho, wo, total);
    return sp_check_launch("maxpool3x3s2_bf16_kernel");
}

extern "C" int sp_pixel_shuffle2_nhwc_bf16(const void* x, void* y, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && y, "sp_pixel_shuffle2_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 32 == 0, "sp_pixel_shuffle2_nhwc_bf16: c=%d must be a multiple of 32", c);
    const long long total = (long long)batch * h * w * c / 8;
    hipLaunchKernelGGL(pixel_shuffle2_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(x), reinterpret_cast<u32x4*>(y), h, w, c, total);
    return sp_check_launch("pixel_shuffle2_bf16_kernel");
}

extern "C" int sp_pixel_unshuffle2_nhwc_bf16(const void* dy, void* dx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(dy && dx, "sp_pixel_unshuffle2_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 32 == 0, "sp_pixel_unshuffle2_nhwc_bf16: c=%d must be a multiple of 32", c);
    const long long total = (long long)batch * h * w * c / 8;
    SP_REQUIRE(total * 8 < (1ll << 31), "sp_pixel_unshuffle2_nhwc_bf16: tensor too large");
    hipLaunchKernelGGL(pixel_unshuffle2_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned int*>(dy), reinterpret_cast<u32x4*>(dx), h, w, c, total);
    return sp_check_launch("pixel_unshuffle2_bf16_kernel");
}

extern "C" int sp_upsample_add_nhwc_bf16(const void* x, const void* base, void* y, int batch, int h, int w, int c, int factor, int relu,
                                         void* stream) {
    SP_REQUIRE(x && base && y, "sp_upsample_add_nhwc_bf16: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && factor >= 1, "sp_upsample_add_nhwc_bf16: bad shape");
    const long long total = (long long)batch * h * factor * w * factor * (c / 8);
    hipLaunchKernelGGL(upsample_add_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<const u32x4*>(base), reinterpret_cast<u32x4*>(y), h, w, c / 8, factor,
                       relu, total);
    return sp_check_launch("upsample_add_bf16_kernel");
}

extern "C" int sp_upsample_add_n_nhwc(const void* base, int bf16, int n_terms, const void* const* xs, const int32_t* factors, void* y, int batch,
                                      int out_h, int out_w, int c, int relu, void* stream) {
    SP_REQUIRE(base && xs && factors && y, "sp_upsample_add_n_nhwc: null pointer");
    const int vec = bf16 ? 8 : 4;
    SP_REQUIRE(n_terms >= 1 && n_terms <= 3 && batch > 0 && out_h > 0 && out_w > 0 && c > 0 && c % vec == 0,
               "sp_upsample_add_n_nhwc: bad shape (1..3 terms, c %% %d == 0)", vec);
    UpTerms t;
    t.n = n_terms;
    for (int k = 0; k < 3; ++k) { t.x[k] = nullptr; t.h[k] = t.w[k] = t.f[k] = 1; }
    for (int k = 0; k < n_terms; ++k) {
        const int f = factors[k];
        SP_REQUIRE(xs[k] && f >= 1 && out_h % f == 0 && out_w % f == 0, "sp_upsample_add_n_nhwc: term %d: factor %d must divide %dx%d", k, f, out_h, out_w);
        t.x[k] = xs[k]; t.f[k] = f; t.h[k] = out_h / f; t.w[k] = out_w / f;
    }
    const long long total = (long long)batch * out_h * out_w * (c / vec);
    SP_REQUIRE(total * 16 < (1ll << 33), "sp_upsample_add_n_nhwc: tensor too large");
    if (bf16) hipLaunchKernelGGL(upsample_add_n_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, base, t, y, out_h, out_w, c / vec,
                                 relu, total);
    else hipLaunchKernelGGL(upsample_add_n_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, base, t, y, out_h, out_w, c / vec,
                            relu, total);
    return sp_check_launch("upsample_add_n_kernel");
}
