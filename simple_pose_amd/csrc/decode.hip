// decode.hip - fused key-point decoders: one workgroup per (batch, joint) heat map staged in LDS.
//
// Replaces metrics/pose_metrics.py:10-107 (BasicKeyPointDecoder.heat_map_to_axis / __call__,
// GaussTaylorKeyPointDecoder.__call__), which on a GPU is ~40 ATen launches with host syncs from
// boolean-mask indexing.  Here: one launch, 12 KB read + 12 B written per map, no sync.
//
// Numerics follow the reference operation by operation in fp32 (contraction OFF for this file):
//   - argmax: first index of the maximum, NaN wins (torch.max)
//   - 11x11 blur: per output pixel one (ky,kx)-ordered fmaf chain over the zero-padded map - bit-identical
//     to oneDNN's depthwise conv that F.conv2d(groups=J) dispatches to on CPU (oracle pinning)
//   - rescale / clamp / log on the 13 taps only (log correctly rounded via double), finite differences in the
//     reference's association order, closed-form 2x2 solve in double, affine in double, rounded once.
#include "sp_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MAX_KS = 15;
constexpr int STRIP = 12;  // blurred outputs per thread per strip

struct BlurWeights {
    float k[MAX_KS * MAX_KS];
};

enum { MODE_AXIS = 0, MODE_BASIC = 1, MODE_GAUSS_TAYLOR = 2 };

struct DecodeArgs {
    const float* heat;
    const float* trans_inv;
    float* kps;
    float* max_val;
    int J, H, W, ks, mode;
    int PW;       // padded row stride (floats) of the LDS image
    int nstrips;  // strips per row
};

__device__ __forceinline__ float log_cr(float v) { return (float)log((double)v); }

template <int KS>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs a, const BlurWeights bw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = a.H, W = a.W, HW = H * W;
    const int ks = KS > 0 ? KS : a.ks;
    const int P = (a.mode == MODE_GAUSS_TAYLOR) ? ks / 2 : 0;
    const int PW = a.PW;
    const int PH = H + 2 * P;
    float* pad = lds;                       // [PH][PW] zero-bordered raw map
    float* hb = lds + PH * PW;              // [H][W] blurred map (GaussTaylor only)
    __shared__ float red_v[4];
    __shared__ int red_i[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = blockIdx.x;
    const float* __restrict__ h = a.heat + (size_t)m * HW;

    // ---- stage the map (zero border) + per-thread running argmax ----
    for (int i = tid; i < PH * PW; i += 256) pad[i] = 0.f;
    __syncthreads();
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    bool have = false;
    if ((W & 3) == 0) {
        for (int i4 = tid; i4 < (HW >> 2); i4 += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(h)[i4];
            const int i = i4 << 2, y = i / W, x = i - y * W;
            float* d = pad + (y + P) * PW + x + P;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[e] = v[e];
                if (!have || sp_better(v[e], i + e, bv, bi)) { bv = v[e]; bi = i + e; have = true; }
            }
        }
    } else {
        for (int i = tid; i < HW; i += 256) {
            const float v = h[i];
            const int y = i / W, x = i - y * W;
            pad[(y + P) * PW + x + P] = v;
            if (!have || sp_better(v, i, bv, bi)) { bv = v; bi = i; have = true; }
        }
    }
    if (!have) { bv = -__builtin_inff(); bi = 0x7fffffff; }
    sp_wave_argmax(bv, bi);
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    float mx = red_v[0];
    int idx = red_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sp_better(red_v[w], red_i[w], mx, idx)) { mx = red_v[w]; idx = red_i[w]; }
    __syncthreads();  // red_* reused below

    float bmax = 0.f;
    if (a.mode == MODE_GAUSS_TAYLOR) {
        // ---- dense KSxKS blur, (ky,kx)-ordered fmaf chain per pixel; 12-wide register strips ----
        float lmax = -__builtin_inff();
        bool lnan = false;
        const int total = H * a.nstrips;
        for (int st = tid; st < total; st += 256) {
            const int y = st / a.nstrips, x0 = (st - y * a.nstrips) * STRIP;
            float acc[STRIP];
#pragma unroll
            for (int s = 0; s < STRIP; ++s) acc[s] = 0.f;
            if (KS > 0) {
                constexpr int NV = STRIP + (KS > 0 ? KS : 1) - 1;
                constexpr int NV4 = (NV + 3) / 4;
#pragma unroll 1
                for (int ky = 0; ky < KS; ++ky) {  // rolled: the 11 row weights come in as scalar loads per ky
                    float v[NV4 * 4];
                    const f32x4* row = reinterpret_cast<const f32x4*>(pad + (y + ky) * PW + x0);
#pragma unroll
                    for (int q = 0; q < NV4; ++q) {
                        const f32x4 t = row[q];
                        v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
                    }
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const float wgt = bw.k[ky * KS + kx];
#pragma unroll
                        for (int s = 0; s < STRIP; ++s) acc[s] = __builtin_fmaf(v[s + kx], wgt, acc[s]);
                    }
                }
            } else {
                for (int ky = 0; ky < ks; ++ky)
                    for (int kx = 0; kx < ks; ++kx) {
                        const float wgt = bw.k[ky * ks + kx];
#pragma unroll
                        for (int s = 0; s < STRIP; ++s) acc[s] = __builtin_fmaf(pad[(y + ky) * PW + x0 + s + kx], wgt, acc[s]);
                    }
            }
#pragma unroll
            for (int s = 0; s < STRIP; ++s) {
                if (x0 + s < W) {
                    hb[y * W + x0 + s] = acc[s];
                    if (acc[s] != acc[s]) lnan = true;
                    else if (acc[s] > lmax) lmax = acc[s];
                }
            }
        }
        if (lnan) lmax = __builtin_nanf("");
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(lmax, off, SP_WAVE);
            lmax = (o != o || lmax != lmax) ? __builtin_nanf("") : (o > lmax ? o : lmax);
        }
        if (lane == 0) red_v[wave] = lmax;
        __syncthreads();
        bmax = red_v[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float o = red_v[w];
            bmax = (o != o || bmax != bmax) ? __builtin_nanf("") : (o > bmax ? o : bmax);
        }
    }

    if (tid != 0) return;
    // ---- scalar tail (one lane): coordinates, refinement, affine ----
    const float keep = (mx > 0.f) ? 1.f : 0.f;                       // pose_metrics.py:23
    float cx = (float)(idx % W) * keep, cy = (float)(idx / W) * keep;  // :20-22
    a.max_val[m] = mx;
    if (a.mode == MODE_AXIS) {
        a.kps[2 * (size_t)m] = cx;
        a.kps[2 * (size_t)m + 1] = cy;
        return;
    }
    const int xi = (int)cx, yi = (int)cy;
    if (a.mode == MODE_BASIC) {
        if (xi > 1 && xi < W - 1 && yi > 1 && yi < H - 1) {          // :40
            const float ddx = pad[yi * PW + xi + 1] - pad[yi * PW + xi - 1];
            const float ddy = pad[(yi + 1) * PW + xi] - pad[(yi - 1) * PW + xi];
            const float sx = ddx > 0.f ? 1.f : (ddx < 0.f ? -1.f : ddx);
            const float sy = ddy > 0.f ? 1.f : (ddy < 0.f ? -1.f : ddy);
            cx = cx + sx * 0.25f;                                    // :49
            cy = cy + sy * 0.25f;
        }
    } else if (xi > 1 && xi < W - 2 && yi > 1 && yi < H - 2) {       // :78
#define L_(yy, xx) log_cr(fmaxf((hb[(yy) * W + (xx)] * mx) / bmax, 1e-10f))  /* :73 */
        const float c = L_(yi, xi);
        const float dx = 0.5f * (L_(yi, xi + 1) - L_(yi, xi - 1));                       // :80-81
        const float dy = 0.5f * (L_(yi + 1, xi) - L_(yi - 1, xi));                       // :82-83
        const float dxx = 0.25f * ((L_(yi, xi + 2) - 2.f * c) + L_(yi, xi - 2));         // :84-86
        const float dxy = 0.25f * (((L_(yi + 1, xi + 1) - L_(yi - 1, xi + 1)) - L_(yi + 1, xi - 1)) + L_(yi - 1, xi - 1));
        const float dyy = 0.25f * ((L_(yi + 2, xi) - 2.f * c) + L_(yi - 2, xi));         // :91-93
#undef L_
        const float p1 = dxx * dyy, p2 = dxy * dxy;
        const float det32 = p1 - p2;                                                     // :94
        if (det32 != 0.f) {
            const double det = (double)dxx * (double)dyy - (double)dxy * (double)dxy;
            const float ox = (float)(-((double)dyy * (double)dx - (double)dxy * (double)dy) / det);  // :95-100
            const float oy = (float)(-((double)dxx * (double)dy - (double)dxy * (double)dx) / det);
            const float nx = cx + ox, ny = cy + oy;
            cx = (ox != ox) ? ox : (nx > 0.f ? nx : 0.f);                                // :103 clamp(min=0), NaN propagates
            cy = (oy != oy) ? oy : (ny > 0.f ? ny : 0.f);
        }
    }
    const float* t = a.trans_inv + (size_t)(m / a.J) * 6;                                // :105-106
    a.kps[2 * (size_t)m] = (float)((double)cx * (double)t[0] + (double)cy * (double)t[1] + (double)t[2]);
    a.kps[2 * (size_t)m + 1] = (float)((double)cx * (double)t[3] + (double)cy * (double)t[4] + (double)t[5]);
}

// cv2.getGaussianKernel(ks, 0) closed form, outer product in double, cast to fp32 (pose_metrics.py:57-60)
void make_blur_weights(int ks, BlurWeights& bw) {
    double g[MAX_KS];
    const double sigma = 0.3 * ((ks - 1) * 0.5 - 1.0) + 0.8;
    double sum = 0.0;
    for (int i = 0; i < ks; ++i) {
        const double x = i - (ks - 1) * 0.5;
        g[i] = exp(-(x * x) / (2.0 * sigma * sigma));
        sum += g[i];
    }
    for (int i = 0; i < ks; ++i) g[i] /= sum;
    for (int i = 0; i < MAX_KS * MAX_KS; ++i) bw.k[i] = 0.f;
    for (int i = 0; i < ks; ++i)
        for (int j = 0; j < ks; ++j) bw.k[i * ks + j] = (float)(g[i] * g[j]);
}

int launch_decode(const float* heat, const float* trans_inv, int B, int J, int H, int W, int ks, int mode, float* kps,
                  float* max_val, hipStream_t stream, const char* who) {
    SP_REQUIRE(heat && kps && max_val && (mode == MODE_AXIS || trans_inv), "%s: null pointer", who);
    SP_REQUIRE(B > 0 && J > 0 && H > 0 && W > 0, "%s: bad shape B=%d J=%d H=%d W=%d", who, B, J, H, W);
    SP_REQUIRE((long long)B * J * H * W < (1ll << 31), "%s: tensor too large", who);
    DecodeArgs a;
    a.heat = heat; a.trans_inv = trans_inv; a.kps = kps; a.max_val = max_val;
    a.J = J; a.H = H; a.W = W; a.ks = ks; a.mode = mode;
    a.nstrips = (W + STRIP - 1) / STRIP;
    BlurWeights bw;
    size_t lds;
    if (mode == MODE_GAUSS_TAYLOR) {
        SP_REQUIRE(ks >= 1 && ks <= MAX_KS && (ks & 1), "%s: kernel_size=%d must be odd and <= %d", who, ks, MAX_KS);
        make_blur_weights(ks, bw);
        a.PW = ((a.nstrips * STRIP + ks + 3 + 3) / 4) * 4 + 4;  // room for the last strip's 16-B reads; +4 de-phases rows vs 256-B bank rows
        lds = ((size_t)(H + 2 * (ks / 2)) * a.PW + (size_t)H * W) * sizeof(float);
    } else {
        for (int i = 0; i < MAX_KS * MAX_KS; ++i) bw.k[i] = 0.f;
        a.PW = ((W + 3) / 4) * 4;
        lds = (size_t)H * a.PW * sizeof(float);
    }
    SP_REQUIRE(lds <= 64 * 1024, "%s: heat map %dx%d needs %zu B of LDS (> 64 KiB)", who, H, W, lds);
    const dim3 grid(B * J), block(256);
    if (mode == MODE_GAUSS_TAYLOR && ks == 11)
        hipLaunchKernelGGL(decode_kernel<11>, grid, block, lds, stream, a, bw);
    else
        hipLaunchKernelGGL(decode_kernel<0>, grid, block, lds, stream, a, bw);
    return sp_check_launch(who);
}

}  // namespace

extern "C" int sp_heat_map_to_axis(const float* heat, int batch, int joints, int h, int w, float* coords, float* max_val, void* stream) {
    return launch_decode(heat, nullptr, batch, joints, h, w, 1, MODE_AXIS, coords, max_val, (hipStream_t)stream, "sp_heat_map_to_axis");
}

extern "C" int sp_decode_gauss_taylor(const float* heat, const float* trans_inv, int batch, int joints, int h, int w, int kernel_size,
                                      float* kps, float* max_val, void* stream) {
    return launch_decode(heat, trans_inv, batch, joints, h, w, kernel_size, MODE_GAUSS_TAYLOR, kps, max_val, (hipStream_t)stream,
                         "sp_decode_gauss_taylor");
}

extern "C" int sp_decode_basic(const float* heat, const float* trans_inv, int batch, int joints, int h, int w, float* kps, float* max_val,
                               void* stream) {
    return launch_decode(heat, trans_inv, batch, joints, h, w, 1, MODE_BASIC, kps, max_val, (hipStream_t)stream, "sp_decode_basic");
}
