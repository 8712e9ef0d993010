// warp.hip - person crops straight from the full image on the GPU: cv.warpAffine(img, M, (w,h), flags=INTER_LINEAR) for N boxes of
// one 8-bit BGR image in one launch.  Replaces the per-person CPU warp of the detector-driven path (datasets/naive_data.py:50,
// commons/transforms.py:214).  Integer arithmetic of OpenCV's fixed-point bilinear remap, restated from the published algorithm
// (imgproc/imgwarp.cpp; opencv-python is not available to pin against: see oracle/pose_oracle.c sp_oracle_warp_affine_u8c3):
// coordinates in 1/1024 px rounded to 1/32 px, four 15-bit weights, (sum + 2^14) >> 15, BORDER_CONSTANT 0.
// HBM-bound gather: 3 B written per output pixel, <= 12 B read (neighbouring lanes share lines).
#include "sp_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }       // saturate_cast<int>(double): half to even
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

constexpr int WARP_BATCH = 32;                                // crops per launch: their inverse maps travel as kernel arguments
struct WarpMaps { double m[WARP_BATCH][6]; };

__global__ __launch_bounds__(256) void warp_affine_u8c3_kernel(const unsigned char* __restrict__ src, int H, int W, const WarpMaps maps,
                                                               unsigned char* __restrict__ dst, int oh, int ow) {
    const int n = blockIdx.y;
    double M[6];                                              // dst -> src map, inverted on the host exactly as cv::warpAffine does
#pragma unroll
    for (int i = 0; i < 6; ++i) M[i] = maps.m[n][i];
    constexpr int AB_SCALE = 1 << 10, round_delta = AB_SCALE / 32 / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < oh * ow; i += gridDim.x * 256) {
        const int y = i / ow, x = i - y * ow;
        const int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        const int X = (X0 + cv_round(M[0] * x * AB_SCALE)) >> 5, Y = (Y0 + cv_round(M[3] * x * AB_SCALE)) >> 5;
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5), fx = X & 31, fy = Y & 31;
        int w0 = (32 - fy) * (32 - fx) * 32, w1 = (32 - fy) * fx * 32, w2 = fy * (32 - fx) * 32, w3 = fy * fx * 32;
        if (w0 == 32768) { w0 = 32767; w3 = 1; }              // the table's one saturated entry and its correction
        unsigned char* d = dst + ((size_t)n * oh * ow + i) * 3;
        if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) { d[0] = 0; d[1] = 0; d[2] = 0; continue; }
        const bool x0 = sx >= 0, x1 = sx + 1 < W, y0 = sy >= 0, y1 = sy + 1 < H;
        const unsigned char* p = src + ((long long)sy * W + sx) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int v0 = (x0 && y0) ? p[k] : 0, v1 = (x1 && y0) ? p[3 + k] : 0;
            const int v2 = (x0 && y1) ? p[(long long)W * 3 + k] : 0, v3 = (x1 && y1) ? p[(long long)W * 3 + 3 + k] : 0;
            const int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
            d[k] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

}  // namespace

extern "C" int sp_warp_affine_u8c3(const unsigned char* src, int src_h, int src_w, const double* m_fwd_host, int crops, unsigned char* dst,
                                   int out_h, int out_w, void* stream) {
    SP_REQUIRE(src && m_fwd_host && dst, "sp_warp_affine_u8c3: null pointer");
    SP_REQUIRE(src_h > 0 && src_w > 0 && src_h <= 32767 && src_w <= 32767 && crops > 0 && out_h > 0 && out_w > 0,
               "sp_warp_affine_u8c3: bad shape src %dx%d crops=%d out %dx%d", src_h, src_w, crops, out_h, out_w);
    const int blocks = sp_ceil_div((long long)out_h * out_w, 256 * 4);
    for (int c0 = 0; c0 < crops; c0 += WARP_BATCH) {
        const int nb = crops - c0 < WARP_BATCH ? crops - c0 : WARP_BATCH;
        WarpMaps maps;
        for (int n = 0; n < nb; ++n) {                        // cv::warpAffine without WARP_INVERSE_MAP: invert in double, this way
            double M[6];
            for (int i = 0; i < 6; ++i) M[i] = m_fwd_host[(size_t)(c0 + n) * 6 + i];
            double D = M[0] * M[4] - M[1] * M[3];
            D = D != 0 ? 1. / D : 0;
            const double A11 = M[4] * D, A22 = M[0] * D;
            M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
            const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
            M[2] = b1; M[5] = b2;
            for (int i = 0; i < 6; ++i) maps.m[n][i] = M[i];
        }
        hipLaunchKernelGGL(warp_affine_u8c3_kernel, dim3(blocks, nb), dim3(256), 0, (hipStream_t)stream, src, src_h, src_w, maps,
                           dst + (size_t)c0 * out_h * out_w * 3, out_h, out_w);
    }
    return sp_check_launch("warp_affine_u8c3_kernel");
}
