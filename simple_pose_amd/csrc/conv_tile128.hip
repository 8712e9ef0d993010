// conv_tile128.hip - 3x3 stride-1 convolution 128 -> 128 channels (bf16) as a tile kernel: HRNet's third branch (nets/pose_hrnet.py BasicBlock
// convs at 16x12 per 256x192 image: 59 of the 293 convs).  Through the LDS-DMA ring (implicit GEMM, 128x128 tiles) such a layer is 192 tiles of
// 18 K tiles on 256 CUs: 20.5 us at bs=128 for 7.2 GFLOP and 21 MB.  Here - the structure of conv_direct.hip's 64-channel kernel and of
// conv_trans.hip - a workgroup of 8 waves owns a 16 x 12 pixel tile (the whole map of a 256x192 image): the 18 x 14 halo sits in LDS once (linear
// 272-byte pixel rows = 256 B + 16 B of padding, row stride 64 mod 256 bytes: the 4 x 4 pixels of a 16-lane read beat land on 16 distinct bank
// groups, a tap is a compile-time offset), the filter streams per TAP (128 rows x 256 B = 32 KB) through a double-buffered LDS stage, requested
// three taps ahead into a ring of three register sets (the stream runs on across tiles).  Wave (mg, nt) multiplies three row tiles of 8 x 4
// pixels (rows 8 mg .. 8 mg + 7, columns 4 i .. 4 i + 3) with channels 32 nt .. 32 nt + 31: one B fragment feeds three MFMAs.
// Reduction order (tap, channel) with one MFMA chain per output = the implicit GEMM's: bit-identical results
// (tests/test_gpu_parity.py::test_direct_3x3_kernels_are_bit_identical_to_the_implicit_gemm).
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C9 = 128;
constexpr int T9R = 16, T9C = 12;
constexpr int H9R = T9R + 2, H9C = T9C + 2;               // 18 x 14 halo
constexpr int PX9 = 272;                                   // bytes per halo pixel (256 + 16)
constexpr int RS9 = H9C * PX9 + 96;                        // 3,904 = 64 (mod 256) bytes per halo row
constexpr int X9_BYTES = H9R * RS9;                        // 70,272
constexpr int W9_STAGE = C9 * 256;                         // one tap of the filter: 32,768 B
constexpr int LDS9 = X9_BYTES + 2 * W9_STAGE;              // 135,808
constexpr int NH9 = (H9R * H9C * 16 + 511) / 512;          // 16-byte halo pieces per thread (8)
constexpr unsigned OOB = 0x80000000u;

struct Tile128Args {
    const void* x;        // NHWC bf16 [B,H,W,128]
    const void* w;        // packed [n_pad >= 128][k_pad = 1152] bf16, K = (tap, channel)
    const float* scale;
    const float* shift;
    const void* res;      // NHWC bf16 [B,H,W,128] or null
    void* y;
    int H, W, k_pad, batch;
    int tiles_x, tiles_y;
    int relu;
    int x_bytes, w_bytes;
};

__global__ __launch_bounds__(512, 2) void conv3x3_c128_tile_kernel(const Tile128Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem9[];
    unsigned char* const Xs = smem9;
    unsigned char* const Ws = smem9 + X9_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int per_img = p.tiles_x * p.tiles_y;
    const int ntiles = per_img * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.x_bytes, 0x00020000);

    // ---- halo: piece q = tid + 512 i -> pixel q >> 4, 16-byte piece q & 15 (the same piece index for every i) ----
    const int h_pc = tid & 15;
    u32x4 hv[NH9];
    auto req_halo = [&](int tile) {
        const int b = tile / per_img, rem = tile - b * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int i = 0; i < NH9; ++i) {
            const int q = tid + 512 * i, P = q >> 4;
            const int hy = (P * 4682) >> 16, hx = P - hy * H9C;          // P / 14 for P < 256
            const int iy = ty * T9R - 1 + hy, ix = tx * T9C - 1 + hx;
            const bool ok = tile < ntiles && q < H9R * H9C * 16 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * C9 + h_pc * 8) * 2) : OOB, 0, 0);
        }
    };
    auto put_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NH9; ++i) {
            const int q = tid + 512 * i, P = q >> 4;
            const int hy = (P * 4682) >> 16, hx = P - hy * H9C;
            if (q < H9R * H9C * 16) *reinterpret_cast<u32x4*>(Xs + hy * RS9 + hx * PX9 + (h_pc << 4)) = hv[i];
        }
    };
    // ---- filter stage = one tap: 128 rows x 16 pieces; thread t: piece t & 15 of rows (t >> 4) + 32 i.  Row n keeps piece pc at pc ^ (n & 15):
    //      the 16 consecutive rows of a B-fragment read beat then hit 16 distinct 16-byte slots of the 256-byte bank row ----
    const int w_pc = tid & 15, w_n0 = tid >> 4;
    u32x4 wv[3][4];
    auto req_w = [&](int tap, u32x4* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dst[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)(((w_n0 + 32 * i) * p.k_pad + tap * C9 + w_pc * 8) * 2), 0, 0);
    };
    auto put_w = [&](int buf, const u32x4* src) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = w_n0 + 32 * i;
            *reinterpret_cast<u32x4*>(Ws + buf * W9_STAGE + n * 256 + ((w_pc ^ (n & 15)) << 4)) = src[i];
        }
    };
    // ---- fragments ----
    const int mg = wave >> 2, nt = wave & 3;
    const int a_r = 8 * mg + (fr >> 2), a_c = fr & 3;               // this lane's pixel of row tile i: tile row a_r, column 4 i + a_c
    const int x_a = a_r * RS9 + a_c * PX9 + (fh << 4);              // + tap offset + 4 i pixels + 32 j
    const int w_n = 32 * nt + fr;
    int w_f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w_f[j] = w_n * 256 + (((2 * j + fh) ^ (w_n & 15)) << 4);
    float sc[8], sh[8];
    const int chunk = lane & 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = p.scale ? p.scale[32 * nt + chunk * 8 + e] : 1.f; sh[e] = p.shift ? p.shift[32 * nt + chunk * 8 + e] : 0.f; }
    float* const tr = reinterpret_cast<float*>(smem9) + wave * 1024;   // epilogue transpose: 4 KB per wave inside the (then idle) halo

    int tile = blockIdx.x;
    req_halo(tile);
    req_w(0, wv[0]); req_w(1, wv[1]); req_w(2, wv[2]);
    put_w(0, wv[0]);
    req_w(3, wv[0]);
    int gs = 0;                                                    // stages done so far: LDS buffer of stage s is (gs & 1) at run time
    for (; tile < ntiles; tile += gridDim.x) {
        put_halo();                                                // (every wave has passed the barrier that ended the halo's last use)
        req_halo(tile + gridDim.x);
        const int b = tile / per_img, rem = tile - b * per_img;
        const int tyy = rem / p.tiles_x, txx = rem - tyy * p.tiles_x;
        unsigned ooff[3][2];
        u32x4 rv[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2);                  // pixel of the 32-pixel row tile: (row >> 2, row & 3)
                const int oy = tyy * T9R + 8 * mg + (row >> 2), ox = txx * T9C + 4 * i + (row & 3);
                ooff[i][it] = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C9 + 32 * nt + chunk * 8) * 2) : OOB;
                rv[i][it] = u32x4{0u, 0u, 0u, 0u};
                if (p.res) rv[i][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, ooff[i][it], 0, 0);
            }
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // tap t + 1 -> the other LDS buffer (its last readers finished before the barrier that ended stage t - 1); its register set then
            // takes tap t + 4 (of the next tile from t = 5 on)
            put_w((gs + 1) & 1, wv[(t + 1) % 3]);
            req_w((t + 4) % 9, wv[(t + 1) % 3]);
            if (t == 0) __syncthreads();                           // the new halo is visible
            const unsigned char* Wb = Ws + (gs & 1) * W9_STAGE;
            const int toff = (t / 3) * RS9 + (t % 3) * PX9;         // compile-time
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 wb = *reinterpret_cast<const u32x4*>(Wb + w_f[j]);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const u32x4 xa = *reinterpret_cast<const u32x4*>(Xs + x_a + toff + i * 4 * PX9 + j * 32);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, wb), acc[i], 0, 0, 0);
                }
            }
            __syncthreads();                                       // stage done everywhere: its buffer (and after t = 8 the halo) may be overwritten
            ++gs;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[i][r];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2);
                float v[8];
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4 * e4);
                    v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                if (p.res) {
                    const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[i][it]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff[i][it], 0, 0);
            }
        }
        __syncthreads();                                           // the transposes are done before the next halo lands
    }
}

}  // namespace

bool sp_tile128_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;
    return d && d->c_in == C9 && (d->flags & SP_CONV_BF16) && !(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) &&
           d->c_out == C9 && d->out_c == C9 && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) &&
           d->dy0 == -1 && d->dx0 == -1 && d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == 9 * C9 &&
           d->n_pad >= C9 && d->grid_h == d->in_h && d->grid_w == d->in_w && d->out_h == d->in_h && d->out_w == d->in_w && d->oy_mul == 1 &&
           d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0;
}

int sp_tile128_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual,
                      void* y, void* stream) {
    const long long elems = (long long)d->batch * d->in_h * d->in_w * C9;
    SP_REQUIRE(elems < (1ll << 29), "sp_conv3x3_direct: tensor too large");
    if (sp_name_query_active()) { sp_name_query_set("conv3x3_c128_tile_kernel"); return SP_OK; }
    Tile128Args a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.k_pad = d->k_pad; a.batch = d->batch;
    a.relu = (d->flags & SP_CONV_RELU) ? 1 : 0;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    a.tiles_x = (d->in_w + T9C - 1) / T9C; a.tiles_y = (d->in_h + T9R - 1) / T9R;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_conv3x3_direct: too many tiles");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c128_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS9);
    if (e != hipSuccess) { sp_set_error("sp_conv3x3_direct: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDS9, hipGetErrorString(e)); return SP_ELAUNCH; }
    const long long grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL(conv3x3_c128_tile_kernel, dim3((unsigned)grid), dim3(512), LDS9, (hipStream_t)stream, a);
    return sp_check_launch("conv3x3_c128_tile_kernel");
}
