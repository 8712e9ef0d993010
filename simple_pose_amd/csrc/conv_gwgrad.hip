// conv_gwgrad.hip - weight gradient of a GROUPED convolution (nn.Conv2d(groups = g) with c_out == c_in: the 3x3 of the resnext* Bottlenecks,
// nets/pose_resnet_dconv.py:97-101,342-368):
//
//     dW[o][cl][ky][kx] = sum over (b, oy, ox) of dz[b, oy, ox, o] * x[b, oy*s - p + ky, ox*s - p + kx, group(o)*cpg + cl]
//
// Per group this is a [cpg x M] . [M x cpg*taps] product - 1 / groups of the dense layer's FLOPs (0.9 GFLOP per layer at 32 images) and far too
// thin for an MFMA tile (cpg = 4 ... 64 output columns per group): it is a streaming reduction over pixels, HBM / LDS bound.  One workgroup of
// 256 threads owns OB = 256 / cpg consecutive output channels (whole groups, or a slice of one) and a strided set of output rows (b, oy):
// thread t = (o, cl) keeps its kh*kw sums in registers, the row's dz slice and the kh input rows of the workgroup's channel range are staged in
// LDS as fp32 (zero-padded in x, so a tap is an offset), and every chunk leaves its partial sums in a slab that a second launch folds in a FIXED
// order in fp64 - no atomics: the step stays bit-reproducible.
#include "sp_common.h"

namespace {

constexpr int GW_THREADS = 256;
constexpr int GW_MAX_TAPS = 9;

struct GwArgs {
    const void* x;
    const void* dz;
    float* part;          // [chunks][c * cpg * taps]
    int bf16;
    int batch, in_h, in_w, out_h, out_w, c, cpg, kh, kw, stride, pad;
    int ob, ci;           // output channels per workgroup, input channels it stages (max(ob, cpg))
    int chunks;
};

template <bool BF16>
__device__ __forceinline__ float gw_load(const void* p, long long i) {
    if constexpr (BF16) return (float)reinterpret_cast<const __bf16*>(p)[i];
    else return reinterpret_cast<const float*>(p)[i];
}

template <bool BF16>
__global__ __launch_bounds__(GW_THREADS) void gwgrad_partial_kernel(const GwArgs p) {
    extern __shared__ float gws[];
    const int taps = p.kh * p.kw;
    const int wpad = p.in_w + 2 * p.pad;                     // staged row: `pad` zero pixels on either side
    float* const xs = gws;                                   // [kh][wpad][ci]
    float* const ds = gws + p.kh * wpad * p.ci;              // [out_w][ob]
    const int tid = threadIdx.x;
    const int o0 = blockIdx.y * p.ob;
    const int c0 = (o0 / p.cpg) * p.cpg;                     // first staged input channel (the first group this workgroup touches)
    const int ol = tid / p.cpg, cl = tid - ol * p.cpg;
    const int cin = ((o0 + ol) / p.cpg) * p.cpg + cl - c0;   // this thread's input channel inside the staged range
    float acc[GW_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < GW_MAX_TAPS; ++t) acc[t] = 0.f;
    const int rows = p.batch * p.out_h;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int b = r / p.out_h, oy = r - b * p.out_h;
        __syncthreads();                                     // the previous row's tiles are consumed
        for (int i = tid; i < p.kh * wpad * p.ci; i += GW_THREADS) {
            const int ch = i % p.ci, q = i / p.ci;
            const int px = q % wpad, ky = q / wpad;
            const int iy = oy * p.stride - p.pad + ky, ix = px - p.pad;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w)
                v = gw_load<BF16>(p.x, (((long long)b * p.in_h + iy) * p.in_w + ix) * p.c + c0 + ch);
            xs[i] = v;
        }
        for (int i = tid; i < p.out_w * p.ob; i += GW_THREADS) {
            const int ch = i % p.ob, ox = i / p.ob;
            ds[i] = gw_load<BF16>(p.dz, (((long long)b * p.out_h + oy) * p.out_w + ox) * p.c + o0 + ch);
        }
        __syncthreads();
        if (p.kh == 3 && p.kw == 3 && (p.stride == 1 || p.stride == 2)) {
            // 3x3: consecutive output pixels share two (stride 1) or one (stride 2) of their three input columns - the window lives in
            // registers and only the new column(s) are read: 4 (7) LDS reads per output pixel instead of 10
            const float* x0 = xs + cin;
            const float* x1 = x0 + wpad * p.ci;
            const float* x2 = x1 + wpad * p.ci;
            if (p.stride == 1) {
                float a0 = x0[0], a1 = x0[p.ci], b0 = x1[0], b1 = x1[p.ci], c0v = x2[0], c1 = x2[p.ci];
                for (int ox = 0; ox < p.out_w; ++ox) {
                    const float d = ds[ox * p.ob + ol];
                    const int o2 = (ox + 2) * p.ci;
                    const float a2 = x0[o2], b2 = x1[o2], c2 = x2[o2];
                    acc[0] += d * a0; acc[1] += d * a1; acc[2] += d * a2;
                    acc[3] += d * b0; acc[4] += d * b1; acc[5] += d * b2;
                    acc[6] += d * c0v; acc[7] += d * c1; acc[8] += d * c2;
                    a0 = a1; a1 = a2; b0 = b1; b1 = b2; c0v = c1; c1 = c2;
                }
            } else {
                float a0 = x0[0], b0 = x1[0], c0v = x2[0];
                for (int ox = 0; ox < p.out_w; ++ox) {
                    const float d = ds[ox * p.ob + ol];
                    const int o1 = (2 * ox + 1) * p.ci, o2 = o1 + p.ci;
                    const float a1 = x0[o1], a2 = x0[o2], b1 = x1[o1], b2 = x1[o2], c1 = x2[o1], c2 = x2[o2];
                    acc[0] += d * a0; acc[1] += d * a1; acc[2] += d * a2;
                    acc[3] += d * b0; acc[4] += d * b1; acc[5] += d * b2;
                    acc[6] += d * c0v; acc[7] += d * c1; acc[8] += d * c2;
                    a0 = a2; b0 = b2; c0v = c2;
                }
            }
        } else {
        for (int ox = 0; ox < p.out_w; ++ox) {
            const float d = ds[ox * p.ob + ol];
            const float* xr = xs + (ox * p.stride) * p.ci + cin;         // tap (ky, kx) = xr[(ky * wpad + kx) * ci]
#pragma unroll
            for (int t = 0; t < GW_MAX_TAPS; ++t) {
                if (t < taps) {
                    const int ky = t / p.kw, kx = t - ky * p.kw;
                    acc[t] += d * xr[(ky * wpad + kx) * p.ci];
                }
            }
        }
        }
    }
    float* out = p.part + (long long)blockIdx.x * p.c * p.cpg * taps + ((long long)(o0 + ol) * p.cpg + cl) * taps;
#pragma unroll
    for (int t = 0; t < GW_MAX_TAPS; ++t)
        if (t < taps) out[t] = acc[t];
}

__global__ void gwgrad_fold_kernel(const float* __restrict__ part, int chunks, long long n, float* __restrict__ dw) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < chunks; ++k) s += (double)part[(long long)k * n + i];     // fixed order
        dw[i] = (float)s;
    }
}

int gw_plan(int batch, int out_h, int c, int groups, int kh, int kw, int* ob, int* ci, int* chunks) {
    SP_REQUIRE(batch > 0 && out_h > 0 && c > 0 && groups > 1 && c % groups == 0 && kh > 0 && kw > 0 && kh * kw <= GW_MAX_TAPS,
               "sp_conv2d_wgrad_grouped: bad shape (batch %d, out_h %d, c %d, groups %d, %dx%d taps; at most %d taps)", batch, out_h, c, groups, kh, kw, GW_MAX_TAPS);
    const int cpg = c / groups;
    SP_REQUIRE(cpg <= GW_THREADS && GW_THREADS % cpg == 0 && c % (GW_THREADS / cpg) == 0,
               "sp_conv2d_wgrad_grouped: group width %d must divide %d (and %d / width must divide c = %d)", cpg, GW_THREADS, GW_THREADS, c);
    *ob = GW_THREADS / cpg;
    *ci = *ob > cpg ? *ob : cpg;
    const int by = c / *ob;
    const long long rows = (long long)batch * out_h;
    long long ch = 1024 / by;                                 // ~1,024 workgroups in flight over the chip
    if (ch < 1) ch = 1;
    if (ch > rows) ch = rows;
    *chunks = (int)ch;
    return SP_OK;
}

}  // namespace

extern "C" int sp_conv2d_wgrad_grouped_workspace(int batch, int out_h, int c, int groups, int kh, int kw, int64_t* bytes) {
    SP_REQUIRE(bytes, "sp_conv2d_wgrad_grouped_workspace: null pointer");
    int ob, ci, chunks;
    const int rc = gw_plan(batch, out_h, c, groups, kh, kw, &ob, &ci, &chunks);
    if (rc != SP_OK) return rc;
    *bytes = (int64_t)chunks * c * (c / groups) * kh * kw * 4;
    return SP_OK;
}

extern "C" int sp_conv2d_wgrad_grouped(const void* x, const void* dz, int bf16, int batch, int in_h, int in_w, int out_h, int out_w, int c, int groups,
                                       int kh, int kw, int stride, int pad, float* dw, void* workspace, int64_t workspace_bytes, void* stream) {
    SP_REQUIRE(x && dz && dw && workspace, "sp_conv2d_wgrad_grouped: null pointer");
    SP_REQUIRE(in_h > 0 && in_w > 0 && out_w > 0 && stride > 0 && pad >= 0, "sp_conv2d_wgrad_grouped: bad geometry");
    SP_REQUIRE((in_h + 2 * pad - kh) / stride + 1 == out_h && (in_w + 2 * pad - kw) / stride + 1 == out_w,
               "sp_conv2d_wgrad_grouped: out %dx%d is not the convolution of in %dx%d (k %dx%d, stride %d, pad %d)", out_h, out_w, in_h, in_w, kh, kw, stride, pad);
    GwArgs a;
    int rc = gw_plan(batch, out_h, c, groups, kh, kw, &a.ob, &a.ci, &a.chunks);
    if (rc != SP_OK) return rc;
    const long long n = (long long)c * (c / groups) * kh * kw;
    SP_REQUIRE(workspace_bytes >= (int64_t)a.chunks * n * 4, "sp_conv2d_wgrad_grouped: workspace of %lld bytes, %lld needed (sp_conv2d_wgrad_grouped_workspace)",
               (long long)workspace_bytes, (long long)a.chunks * n * 4);
    SP_REQUIRE((long long)batch * in_h * in_w * c < (1ll << 40), "sp_conv2d_wgrad_grouped: tensor too large");
    a.x = x; a.dz = dz; a.part = reinterpret_cast<float*>(workspace); a.bf16 = bf16 ? 1 : 0;
    a.batch = batch; a.in_h = in_h; a.in_w = in_w; a.out_h = out_h; a.out_w = out_w; a.c = c; a.cpg = c / groups;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    const size_t lds = ((size_t)kh * (in_w + 2 * pad) * a.ci + (size_t)out_w * a.ob) * sizeof(float);
    SP_REQUIRE(lds <= 160 * 1024, "sp_conv2d_wgrad_grouped: a row of %d pixels x %d channels does not fit the staging tile (%zu bytes of LDS)", in_w, a.ci, lds);
    const void* fn = bf16 ? reinterpret_cast<const void*>(&gwgrad_partial_kernel<true>) : reinterpret_cast<const void*>(&gwgrad_partial_kernel<false>);
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { sp_set_error("sp_conv2d_wgrad_grouped: hipFuncSetAttribute(max dynamic LDS = %zu) failed: %s", lds, hipGetErrorString(e)); return SP_ELAUNCH; }
    const dim3 grid(a.chunks, c / a.ob);
    if (bf16) hipLaunchKernelGGL(gwgrad_partial_kernel<true>, grid, dim3(GW_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gwgrad_partial_kernel<false>, grid, dim3(GW_THREADS), lds, (hipStream_t)stream, a);
    rc = sp_check_launch("gwgrad_partial_kernel");
    if (rc != SP_OK) return rc;
    const int fb = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gwgrad_fold_kernel, dim3(fb), dim3(256), 0, (hipStream_t)stream, a.part, a.chunks, n, dw);
    return sp_check_launch("gwgrad_fold_kernel");
}
