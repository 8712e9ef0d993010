// encode.hip - Gaussian heat-map target generators, one workgroup per (batch, joint) map.
// Replaces commons/transforms.py:167-191 (RefineSimpleTransform.get_heat_map, the one the dataset uses) and
// :80-116 (BasicSimpleTransform.get_heat_map).  204 B read + 12 KB written per sample-joint: store-bound.
#include "sp_common.h"

#pragma clang fp contract(off)

namespace {

struct EncodeArgs {
    const float* joints;  // [B*J][3]
    float* targets;       // [B*J][H][W]
    float* weights;       // [B*J]
    int H, W, stride;
    float sigma;
};

// Refine: un-quantised centre, full-map Gaussian evaluated in float64 (int64 grid - float32 mu promotes), stored fp32.
__global__ __launch_bounds__(256) void encode_refine_kernel(const EncodeArgs a) {
    const int m = blockIdx.x, H = a.H, W = a.W, HW = H * W;
    const float mux = a.joints[3 * m], muy = a.joints[3 * m + 1], vis = a.joints[3 * m + 2];
    const float tmp = a.sigma * 3.f;                                          // transforms.py:177
    // bounds in fp32 (numpy >= 2: np.float32 scalar +- python float stays fp32), int() truncates toward zero  :181-182
    const int ulx = (int)(mux - tmp), uly = (int)(muy - tmp);
    const int brx = (int)((mux + tmp) + 1.f), bry = (int)((muy + tmp) + 1.f);
    const bool outside = ulx >= W || uly >= H || brx < 0 || bry < 0;          // :183
    if (threadIdx.x == 0) a.weights[m] = outside ? 0.f : vis;                 // :175,184
    const bool draw = !outside && vis > 0.5f;                                 // :187
    const double two_s2 = 2.0 * (double)a.sigma * (double)a.sigma;
    float* __restrict__ t = a.targets + (size_t)m * HW;
    const double dmx = (double)mux, dmy = (double)muy;
    if ((W & 3) == 0) {
        for (int i4 = threadIdx.x; i4 < (HW >> 2); i4 += 256) {
            const int i = i4 << 2, y = i / W, x = i - y * W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (draw) {
                const double dy = (double)y - dmy, dy2 = dy * dy;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double dx = (double)(x + e) - dmx;
                    v[e] = (float)exp(-(dx * dx + dy2) / two_s2);             // :190
                }
            }
            reinterpret_cast<f32x4*>(t)[i4] = v;
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) {
            const int y = i / W, x = i - y * W;
            float v = 0.f;
            if (draw) {
                const double dx = (double)x - dmx, dy = (double)y - dmy;
                v = (float)exp(-(dx * dx + dy * dy) / two_s2);
            }
            t[i] = v;
        }
    }
}

// Basic: centre quantised to int(j/stride + 0.5), (6 sigma + 1)^2 fp32 patch, clipped paste.
__global__ __launch_bounds__(256) void encode_basic_kernel(const EncodeArgs a) {
    const int m = blockIdx.x, H = a.H, W = a.W, HW = H * W;
    const float jx = a.joints[3 * m], jy = a.joints[3 * m + 1], vis = a.joints[3 * m + 2];
    const float tmpf = a.sigma * 3.f;                                                     // :91
    const int size = (int)(2.f * tmpf + 1.f);                                             // :103
    const float c0 = (float)(size / 2);                                                   // :106
    const float two_s2 = 2.f * (a.sigma * a.sigma);
    const int mux = (int)(jx / (float)a.stride + 0.5f), muy = (int)(jy / (float)a.stride + 0.5f);  // :95-96
    const int ulx = (int)((double)mux - (double)tmpf), uly = (int)((double)muy - (double)tmpf);     // :98
    const int brx = (int)((double)mux + (double)tmpf + 1.0), bry = (int)((double)muy + (double)tmpf + 1.0);  // :99
    const bool outside = ulx >= W || uly >= H || brx < 0 || bry < 0;                      // :100
    if (threadIdx.x == 0) a.weights[m] = outside ? 0.f : vis;
    const bool draw = !outside && vis > 0.5f;                                             // :114
    const int x_lo = ulx > 0 ? ulx : 0, x_hi = brx < W ? brx : W;                        // :111
    const int y_lo = uly > 0 ? uly : 0, y_hi = bry < H ? bry : H;                        // :112
    float* __restrict__ t = a.targets + (size_t)m * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        float v = 0.f;
        if (draw && x >= x_lo && x < x_hi && y >= y_lo && y < y_hi) {
            const float ddx = (float)(x - ulx) - c0, ddy = (float)(y - uly) - c0;         // patch coords :104-107
            const float s = ddx * ddx + ddy * ddy;
            const float e = -s / two_s2;
            v = (float)exp((double)e);
        }
        t[i] = v;
    }
}

int launch_encode(bool refine, const float* joints, int B, int J, int H, int W, float sigma, int stride, float* targets,
                  float* weights, hipStream_t stream, const char* who) {
    SP_REQUIRE(joints && targets && weights, "%s: null pointer", who);
    SP_REQUIRE(B > 0 && J > 0 && H > 0 && W > 0 && sigma > 0.f && stride > 0, "%s: bad shape/sigma/stride", who);
    SP_REQUIRE((long long)B * J * H * W < (1ll << 31), "%s: tensor too large", who);
    EncodeArgs a;
    a.joints = joints; a.targets = targets; a.weights = weights; a.H = H; a.W = W; a.stride = stride; a.sigma = sigma;
    if (refine) hipLaunchKernelGGL(encode_refine_kernel, dim3(B * J), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(encode_basic_kernel, dim3(B * J), dim3(256), 0, stream, a);
    return sp_check_launch(who);
}

}  // namespace

extern "C" int sp_encode_gauss_refine(const float* joints, int batch, int joints_n, int h, int w, float sigma, float* targets,
                                      float* weights, void* stream) {
    return launch_encode(true, joints, batch, joints_n, h, w, sigma, 1, targets, weights, (hipStream_t)stream, "sp_encode_gauss_refine");
}

extern "C" int sp_encode_gauss_basic(const float* joints, int batch, int joints_n, int h, int w, float sigma, int stride,
                                     float* targets, float* weights, void* stream) {
    return launch_encode(false, joints, batch, joints_n, h, w, sigma, stride, targets, weights, (hipStream_t)stream, "sp_encode_gauss_basic");
}
