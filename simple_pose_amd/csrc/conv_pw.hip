// conv_pw.hip - fp32 1x1 convolutions with a SHORT reduction (K = c_in = 64 or 128) and a wide output (c_out a multiple of 256): the
// bottleneck's expanding conv3 (+ bn3 + residual + relu) and the projection shortcut of layer1 / layer2
// (nets/pose_resnet_dconv.py:112-133, :99-103), sp_conv_desc.kernel = SP_CONV_KERNEL_PW.
//
// These layers sit on BOTH roofs at once: per 64 output rows the MFMAs need ~4-8 us of a CU and the bytes (A once, residual in,
// result out: 144-160 KB) ~8 us of its share of HBM.  The tiled implicit GEMM does load -> 2-4 K tiles -> epilogue per workgroup and
// leaves the overlap to whatever two or three co-resident workgroups happen to be doing: 66-76 % of the floor (239 us vs 181 on
// layer1.*.conv3, 142 vs 95 on layer2.*.conv3 at bs = 128).  Here ONE persistent workgroup per CU streams row tiles:
//   * B (the whole K x 256 weight slice of the workgroup) lives in registers as MFMA fragments for the life of the kernel;
//   * the A tile of row tile t + 1 and the residual of row tile t are requested before the MFMAs of tile t and land behind them
//     (A in a second LDS buffer, the residual in registers), the stores of tile t - 1 drain meanwhile: HBM never waits for math;
//   * the accumulators go through a wave-private LDS transpose so that every residual load / store is 16 bytes of one pixel.
// Bits: identical to the tiled kernel - v_mfma_f32_32x32x2_f32 on the same k pairs in the same order (per 8-k group g: k = 8g+s with
// 8g+4+s, s = 0..3), epilogue acc * scale + shift, + residual, relu in that order.
#include "sp_common.h"

namespace {

struct PwArgs {
    const float* x;          // [M][K]
    const float* w;          // packed [N][K]
    const float* scale;      // [N] or null
    const float* shift;      // [N] or null
    const float* res;        // [M][N] or null
    float* y;                // [M][N]
    int M, N, nchunks, tiles_m, relu;
    unsigned x_bytes, y_bytes;
};

constexpr int BM = 64;
constexpr int TRS = 68;      // transpose row stride in floats (16-byte aligned rows, 4-bank skew)
constexpr unsigned OOB = 0x80000000u;

template <int K>
__global__ __launch_bounds__(256, 1) void conv_pw_kernel(const PwArgs p) {
    constexpr int CH = K / 4;                     // 16-byte chunks per A row
    constexpr int NG = K / 8;                     // 8-k groups
    constexpr int APT = BM * CH / 256;            // A chunks per thread per tile
    extern __shared__ __align__(16) float smem[];
    float* const As = smem;                       // [2][BM * K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const tr = smem + 2 * BM * K + wave * (BM * TRS);
    const int fr = lane & 31, fh = lane >> 5;
    const int nc = blockIdx.x % p.nchunks, slot = blockIdx.x / p.nchunks, nslots = gridDim.x / p.nchunks;
    const int nbase = nc * 256 + wave * 64;       // this wave's 64 output channels

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), (short)0, p.y_bytes, 0x00020000);

    // ---- B fragments: W[n][8g + 4fh + s] for this lane's column of both 32-column tiles ----
    float fb[NG][2][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p.w + (size_t)(nbase + n * 32 + fr) * K + 8 * g + 4 * fh);
            fb[g][n][0] = t[0]; fb[g][n][1] = t[1]; fb[g][n][2] = t[2]; fb[g][n][3] = t[3];
        }
    // ---- epilogue geometry: lane = (16-byte chunk c of the wave's 64 columns, row rsub of every group of 4 rows) ----
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc4 = *reinterpret_cast<const f32x4*>(p.scale + nbase + c4);
    if (p.shift) sh4 = *reinterpret_cast<const f32x4*>(p.shift + nbase + c4);

    // A staging: thread -> chunks q = tid + 256 i of the tile (the tile is one contiguous 64 x K block of x)
    auto swz = [](int row, int chunk) { return row * K + ((chunk ^ (row & (CH - 1))) << 2); };
    u32x4 sa[APT];
    auto load_a = [&](int mt) {
        const unsigned base = (unsigned)mt * (BM * K * 4);
#pragma unroll
        for (int i = 0; i < APT; ++i) sa[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, base + (unsigned)((tid + 256 * i) * 16), 0, 0);
    };
    auto park_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int q = tid + 256 * i;
            *reinterpret_cast<u32x4*>(As + buf * (BM * K) + swz(q / CH, q % CH)) = sa[i];
        }
    };

    int mt = slot;
    if (mt >= p.tiles_m) return;
    load_a(mt);
    park_a(0);
    __syncthreads();
    int cur = 0;
    for (; mt < p.tiles_m; mt += nslots, cur ^= 1) {
        const int m0 = mt * BM;
        const bool more = mt + nslots < p.tiles_m;
        if (more) load_a(mt + nslots);
        // residual of this tile: 16 x 16 bytes per lane, in flight behind the MFMAs
        unsigned off[16];
        u32x4 rv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = m0 + it * 4 + rsub;
            off[it] = row < p.M ? (unsigned)(((size_t)row * p.N + nbase + c4) * 4) : OOB;
        }
        if (p.res) {
#pragma unroll
            for (int it = 0; it < 16; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, off[it], 0, 0);
        }

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
        const float* a = As + cur * (BM * K);
        f32x4 fa[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a + swz(i * 32 + fr, fh));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a + swz(i * 32 + fr, 2 * (g + 1) + fh));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][s], fb[g][n][s], acc[i][n], 0, 0, 0);
        }

        // ---- epilogue: transpose through the wave's private LDS slice, then 16 bytes of one pixel per lane ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) tr[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * TRS + n * 32 + fr] = acc[i][n][r];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tr + (it * 4 + rsub) * TRS + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc4[e] + sh4[e];
            if (p.res) {
                const f32x4 r4 = __builtin_bit_cast(f32x4, rv[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r4[e];
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, off[it], 0, 0);
        }
        if (more) park_a(cur ^ 1);
        __syncthreads();                          // tile t + 1 is in LDS; every wave has left buffer `cur`
    }
}

// ---- the tail of a stage-opening Bottleneck as ONE launch (round 6; fp32 twin of conv_dual.hip):
//          y = relu( bn3(conv1x1(t)) + bn_d(conv1x1(x)) ),   t and x [M][K], both weight matrices [256 n][K]      (nets/pose_resnet_dconv.py:99-103,120-131)
// The projection shortcut's tensor (402 MB at bs=128 on layer1.0) is neither written nor read back as conv3's residual: 1,406 -> 602 MB for the
// pair.  Same structure as conv_pw_kernel - both B slices in registers for the life of the kernel, both A tiles double-buffered in LDS, the
// two accumulators combined in the accumulator layout (per-lane channel constants) and transposed ONCE - and the same bits as the two
// launches: each product is the tiled kernel's MFMA chain, the shortcut value acc * scale + shift is the fp32 number the two-launch program
// stores, and the sum / ReLU are conv3's epilogue element by element.
struct PwDualArgs {
    const float* x_main;     // t [M][K]
    const float* x_short;    // x [M][K]
    const float* w_main;     // packed [N][K]
    const float* w_short;
    const float *s_main, *h_main, *s_short, *h_short;
    float* y;                // [M][N]
    int M, N, nchunks, tiles_m, relu;
    unsigned x_bytes, y_bytes;
};

template <int K>
__global__ __launch_bounds__(256, 1) void conv_pw_dual_kernel(const PwDualArgs p) {
    constexpr int CH = K / 4;
    constexpr int NG = K / 8;
    constexpr int APT = BM * CH / 256;
    extern __shared__ __align__(16) float smem[];
    float* const As = smem;                       // [2 buffers][2 products][BM * K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* const tr = smem + 4 * BM * K + wave * (BM * TRS);
    const int fr = lane & 31, fh = lane >> 5;
    const int nc = blockIdx.x % p.nchunks, slot = blockIdx.x / p.nchunks, nslots = gridDim.x / p.nchunks;
    const int nbase = nc * 256 + wave * 64;

    const __amdgpu_buffer_rsrc_t xr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x_main), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x_short), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);

    float fb[2][NG][2][4];                        // [product][8-k group][column tile][s]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 t = *reinterpret_cast<const f32x4*>((q ? p.w_short : p.w_main) + (size_t)(nbase + n * 32 + fr) * K + 8 * g + 4 * fh);
                fb[q][g][n][0] = t[0]; fb[q][g][n][1] = t[1]; fb[q][g][n][2] = t[2]; fb[q][g][n][3] = t[3];
            }
    // folded BatchNorms in the ACCUMULATOR layout: this lane's channel of column tile n is nbase + 32 n + fr
    float s3v[2], h3v[2], sdv[2], hdv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ch = nbase + n * 32 + fr;
        s3v[n] = p.s_main ? p.s_main[ch] : 1.f;  h3v[n] = p.h_main ? p.h_main[ch] : 0.f;
        sdv[n] = p.s_short ? p.s_short[ch] : 1.f; hdv[n] = p.h_short ? p.h_short[ch] : 0.f;
    }
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;

    auto swz = [](int row, int chunk) { return row * K + ((chunk ^ (row & (CH - 1))) << 2); };
    u32x4 sa[2][APT];
    auto load_a = [&](int mt) {
        const unsigned base = (unsigned)mt * (BM * K * 4);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            sa[0][i] = __builtin_amdgcn_raw_buffer_load_b128(xr0, base + (unsigned)((tid + 256 * i) * 16), 0, 0);
            sa[1][i] = __builtin_amdgcn_raw_buffer_load_b128(xr1, base + (unsigned)((tid + 256 * i) * 16), 0, 0);
        }
    };
    auto park_a = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < APT; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<u32x4*>(As + (buf * 2 + q) * (BM * K) + swz(e / CH, e % CH)) = sa[q][i];
            }
    };

    int mt = slot;
    if (mt >= p.tiles_m) return;
    load_a(mt);
    park_a(0);
    __syncthreads();
    int cur = 0;
    for (; mt < p.tiles_m; mt += nslots, cur ^= 1) {
        const int m0 = mt * BM;
        const bool more = mt + nslots < p.tiles_m;
        if (more) load_a(mt + nslots);
        f32x16 acc[2][2][2];                      // [product][row tile][column tile]
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][i][n][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* a = As + (cur * 2 + q) * (BM * K);
            f32x4 fa[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a + swz(i * 32 + fr, fh));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(a + swz(i * 32 + fr, 2 * (g + 1) + fh));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[q][i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][s], fb[q][g][n][s], acc[q][i][n], 0, 0, 0);
            }
        }
        // ---- epilogue: combine in the accumulator layout, transpose once, 16 bytes of one pixel per lane ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vd = acc[1][i][n][r] * sdv[n] + hdv[n];      // the shortcut's epilogue: the fp32 value the two-launch program stores
                    float v = acc[0][i][n][r] * s3v[n] + h3v[n];             // conv3's: scale / shift, + residual, ReLU
                    v += vd;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    tr[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * TRS + n * 32 + fr] = v;
                }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = m0 + it * 4 + rsub;
            const unsigned off = row < p.M ? (unsigned)(((size_t)row * p.N + nbase + c4) * 4) : OOB;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tr + (it * 4 + rsub) * TRS + c4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, off, 0, 0);
        }
        if (more) park_a(cur ^ 1);
        __syncthreads();
    }
}

template <int K>
int launch_pw_dual(const PwDualArgs& a, hipStream_t stream) {
    constexpr int lds = (4 * BM * K + 4 * BM * TRS) * 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_dual_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        sp_set_error("conv_pw_dual: hipFuncSetAttribute(max dynamic LDS = %d) failed", lds);
        return SP_ELAUNCH;
    }
    int slots = cus / a.nchunks;
    if (slots < 1) slots = 1;
    if (slots > a.tiles_m) slots = a.tiles_m;
    const int rounds = (a.tiles_m + slots - 1) / slots;
    slots = (a.tiles_m + rounds - 1) / rounds;
    hipLaunchKernelGGL(conv_pw_dual_kernel<K>, dim3(slots * a.nchunks), dim3(256), lds, stream, a);
    return sp_check_launch("conv_pw_dual_kernel");
}

template <int K>
int launch_pw(const PwArgs& a, hipStream_t stream) {
    constexpr int lds = (2 * BM * K + 4 * BM * TRS) * 4;
    static bool opted[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            sp_set_error("conv_pw: hipFuncSetAttribute(max dynamic LDS = %d) failed on device %d", lds, dev);
            return SP_ELAUNCH;
        }
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        opted[dev] = true;
    }
    // one workgroup per CU, a multiple of the channel-chunk count, whole rounds of row tiles per workgroup where possible
    int slots = cus[dev] / a.nchunks;
    if (slots < 1) slots = 1;
    if (slots > a.tiles_m) slots = a.tiles_m;
    const int rounds = (a.tiles_m + slots - 1) / slots;
    slots = (a.tiles_m + rounds - 1) / rounds;
    hipLaunchKernelGGL(conv_pw_kernel<K>, dim3(slots * a.nchunks), dim3(256), lds, stream, a);
    return sp_check_launch("conv_pw_kernel");
}

}  // namespace

extern "C" int sp_conv2d_pw_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return 0;
    if (!d) return 0;
    if (d->flags & (SP_CONV_BF16 | SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) return 0;
    if (d->taps_h != 1 || d->taps_w != 1 || d->stride != 1 || (d->stride_x != 0 && d->stride_x != 1)) return 0;
    if (d->phases_y != 1 || d->phases_x != 1 || d->dy0 != 0 || d->dx0 != 0) return 0;
    if (d->grid_h != d->in_h || d->grid_w != d->in_w || d->out_h != d->in_h || d->out_w != d->in_w) return 0;
    if (d->oy_mul != 1 || d->ox_mul != 1 || d->oy_add != 0 || d->ox_add != 0) return 0;
    if (!(d->c_in == 64 || d->c_in == 128) || d->k_pad != d->c_in) return 0;
    if (d->c_out % 256 != 0 || d->n_pad != d->c_out || d->out_c != d->c_out) return 0;
    return 1;
}

// called by sp_conv2d_fwd for sp_conv_desc.kernel == SP_CONV_KERNEL_PW (the descriptor has passed its checks there)
int sp_conv_pw_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual,
                      void* y, void* stream) {
    SP_REQUIRE(sp_conv2d_pw_ok(d), "sp_conv2d_fwd: kernel = SP_CONV_KERNEL_PW needs an fp32 1x1 stride-1 NHWC layer with c_in 64 / 128 and c_out %% 256 == 0");
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    PwArgs a;
    a.x = reinterpret_cast<const float*>(x); a.w = reinterpret_cast<const float*>(w_packed); a.scale = scale; a.shift = shift;
    a.res = reinterpret_cast<const float*>(residual); a.y = reinterpret_cast<float*>(y);
    a.M = (int)M; a.N = d->c_out; a.nchunks = d->c_out / 256; a.tiles_m = (int)((M + BM - 1) / BM);
    a.relu = (d->flags & SP_CONV_RELU) ? 1 : 0;
    a.x_bytes = (unsigned)(M * d->c_in * 4); a.y_bytes = (unsigned)(M * d->c_out * 4);
    if (sp_name_query_active()) {
        sp_name_query_set("conv_pw_kernel<%d>", d->c_in);
        return SP_OK;
    }
    return d->c_in == 64 ? launch_pw<64>(a, (hipStream_t)stream) : launch_pw<128>(a, (hipStream_t)stream);
}


extern "C" int sp_dual_pw_f32_ok(int64_t rows, int c_main, int c_short, int c_out) {
    return rows > 0 && rows * (int64_t)c_out < (1ll << 30) && c_main == 64 && c_short == 64 && c_out > 0 && c_out % 256 == 0;
}

extern "C" int sp_dual_pw_f32(const float* a_main, const float* w_main_packed, const float* scale_main, const float* shift_main, const float* a_short,
                              const float* w_short_packed, const float* scale_short, const float* shift_short, float* y, int64_t rows, int c_main,
                              int c_short, int c_out, int relu, void* stream) {
    SP_REQUIRE(a_main && w_main_packed && a_short && w_short_packed && y, "sp_dual_pw_f32: null pointer");
    SP_REQUIRE(sp_dual_pw_f32_ok(rows, c_main, c_short, c_out), "sp_dual_pw_f32: two fp32 1x1 stride-1 products of 64 channels each into a multiple of 256 (got %d + %d -> %d, %lld rows)",
               c_main, c_short, c_out, (long long)rows);
    SP_REQUIRE(y != a_main && y != a_short, "sp_dual_pw_f32: y must not alias an input");
    if (sp_name_query_active()) { sp_name_query_set("conv_pw_dual_kernel<64>"); return SP_OK; }
    PwDualArgs a;
    a.x_main = a_main; a.x_short = a_short; a.w_main = w_main_packed; a.w_short = w_short_packed;
    a.s_main = scale_main; a.h_main = shift_main; a.s_short = scale_short; a.h_short = shift_short; a.y = y;
    a.M = (int)rows; a.N = c_out; a.nchunks = c_out / 256; a.tiles_m = (int)((rows + BM - 1) / BM); a.relu = relu ? 1 : 0;
    a.x_bytes = (unsigned)(rows * 64 * 4); a.y_bytes = (unsigned)(rows * c_out * 4);
    return launch_pw_dual<64>(a, (hipStream_t)stream);
}
