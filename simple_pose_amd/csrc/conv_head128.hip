// conv_head128.hip - the DUC head's last layer as a tile kernel (bf16 in, fp32 NCHW heat maps out): nn.Conv2d(128, J <= 32, 3, padding=1) + bias on the
// 64x48 map (nets/pose_resnet_duc.py:172-177, J = 17 joints).  Through the implicit GEMM (128x32 tiles) it gathers every input pixel nine times
// through L2 for 17 output channels: 82.8 us at bs=128 (profiles/r05_duc_bf16_kernel_stats.csv) against ~30 us of HBM time for 100 MB in and 27 MB
// out.  Here - conv_tile128.hip's structure - a workgroup of 6 waves owns a 16 x 12 pixel tile: the 18 x 14 halo sits in LDS once (linear 272-byte
// pixel rows, a tap is a compile-time offset), the whole filter [tap][32 rows][256 B] (72 KB) is resident for the persistent workgroup, wave w
// multiplies ONE row tile of 8 x 4 pixels with the 32 (padded) output channels; the accumulator's four consecutive registers are four consecutive
// x positions of one channel, i.e. one 16-byte NCHW store.  Reduction order (tap, channel) with one MFMA chain per output = the implicit GEMM's:
// bit-identical results (tests/test_gpu_parity.py::test_direct_3x3_kernels_are_bit_identical_to_the_implicit_gemm).
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CH = 128, NH = 32;
constexpr int THR = 16, THC = 12;
constexpr int HHR = THR + 2, HHC = THC + 2;               // 18 x 14 halo
constexpr int PXH = 272;                                   // bytes per halo pixel (256 + 16)
constexpr int RSH = HHC * PXH + 96;                        // 3,904 = 64 (mod 256) bytes per halo row
constexpr int XH_BYTES = HHR * RSH;                        // 70,272
constexpr int WH_BYTES = 9 * NH * 256;                     // 73,728
constexpr int LDSH = XH_BYTES + WH_BYTES;                  // 144,000
constexpr int THREADS_H = 384;
constexpr int NHP = (HHR * HHC * 16 + THREADS_H - 1) / THREADS_H;   // 16-byte halo pieces per thread (11)
constexpr unsigned OOB = 0x80000000u;

struct HeadArgs {
    const void* x;        // NHWC bf16 [B,H,W,128]
    const void* w;        // packed [32][1152] bf16, K = (tap, channel)
    const float* scale;
    const float* shift;
    float* y;             // NCHW fp32 [B,c_out,H,W]
    int H, W, batch, c_out;
    int tiles_x, tiles_y;
    int relu;
    int x_bytes, w_bytes, y_bytes;
};

__global__ __launch_bounds__(THREADS_H, 1) void conv3x3_c128_head_kernel(const HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemh[];
    unsigned char* const Xs = smemh;
    unsigned char* const Ws = smemh + XH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int per_img = p.tiles_x * p.tiles_y;
    const int ntiles = per_img * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);

    u32x4 hv[NHP];
    auto req_halo = [&](int tile) {
        const int b = tile / per_img, rem = tile - b * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const int q = tid + THREADS_H * i, P = q >> 4, pc = q & 15;
            const int hy = (P * 4682) >> 16, hx = P - hy * HHC;          // P / 14 for P < 256
            const int iy = ty * THR - 1 + hy, ix = tx * THC - 1 + hx;
            const bool ok = tile < ntiles && q < HHR * HHC * 16 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * CH + pc * 8) * 2) : OOB, 0, 0);
        }
    };
    auto put_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const int q = tid + THREADS_H * i, P = q >> 4, pc = q & 15;
            const int hy = (P * 4682) >> 16, hx = P - hy * HHC;
            if (q < HHR * HHC * 16) *reinterpret_cast<u32x4*>(Xs + hy * RSH + hx * PXH + (pc << 4)) = hv[i];
        }
    };
    int tile = blockIdx.x;
    req_halo(tile);
    // ---- the whole filter -> LDS once: piece (tap, n, pc) <- W[n][tap * 128 + pc * 8 .. + 8]; row n keeps piece pc at pc ^ (n & 15) ----
    for (int q = tid; q < 9 * NH * 16; q += THREADS_H) {
        const int pc = q & 15, n = (q >> 4) & 31, tap = q >> 9;
        *reinterpret_cast<u32x4*>(Ws + tap * (NH * 256) + n * 256 + ((pc ^ (n & 15)) << 4)) =
            __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)((n * (9 * CH) + tap * CH + pc * 8) * 2), 0, 0);
    }
    // ---- fragments: wave w = row tile (mg = w / 3: rows 8 mg .. 8 mg + 7, i = w % 3: columns 4 i .. 4 i + 3) x the 32 channels ----
    const int mg = wave / 3, ci = wave - 3 * mg;
    const int a_r = 8 * mg + (fr >> 2), a_c = 4 * ci + (fr & 3);
    const int x_a = a_r * RSH + a_c * PXH + (fh << 4);              // + tap offset + 32 j
    int w_f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w_f[j] = fr * 256 + (((2 * j + fh) ^ (fr & 15)) << 4);
    const float sc = (p.scale && fr < p.c_out) ? p.scale[fr] : 1.f, sh = (p.shift && fr < p.c_out) ? p.shift[fr] : 0.f;

    for (; tile < ntiles; tile += gridDim.x) {
        put_halo();
        __syncthreads();                                          // halo (and, first time round, the filter) visible
        req_halo(tile + gridDim.x);
        const int b = tile / per_img, rem = tile - b * per_img;
        const int tyy = rem / p.tiles_x, txx = rem - tyy * p.tiles_x;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3) * RSH + (t % 3) * PXH;         // compile-time
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 wb = *reinterpret_cast<const u32x4*>(Ws + t * (NH * 256) + w_f[j]);
                const u32x4 xa = *reinterpret_cast<const u32x4*>(Xs + x_a + toff + j * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, wb), acc, 0, 0, 0);
            }
        }
        // register r = 4 q + s of lane (fr, fh): channel fr, pixel row (2 q + fh) of the 8, column s of the 4 -> four consecutive x: one 16-byte store
        const int ox0 = txx * THC + 4 * ci;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = tyy * THR + 8 * mg + 2 * q + fh;
            float ts[4];                                  // (scalars: hipcc's bit_cast of an ext_vector ELEMENT reads element 0)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float t = acc[4 * q + s] * sc + sh;
                if (p.relu) t = t > 0.f ? t : 0.f;
                ts[s] = t;
            }
            const f32x4 v = {ts[0], ts[1], ts[2], ts[3]};
            const bool row_ok = fr < p.c_out && oy < p.H;
            const long long base = (((long long)b * p.c_out + fr) * p.H + oy) * p.W + ox0;
            if (row_ok && ox0 + 3 < p.W && (p.W & 3) == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, (unsigned)(base * 4), 0, 0);
            } else if (row_ok) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (ox0 + s < p.W) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ts[s]), yr, (unsigned)((base + s) * 4), 0, 0);
            }
        }
        __syncthreads();                                          // every wave is done with the halo before the next one lands
    }
}

}  // namespace

bool sp_head128_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;
    const unsigned need = SP_CONV_BF16 | SP_CONV_OUT_NCHW;
    return d && d->c_in == CH && (d->flags & need) == need && !(d->flags & SP_CONV_PIXEL_SHUFFLE) && d->c_out > 0 && d->c_out <= NH && d->n_pad == NH &&
           d->out_c == d->c_out && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) && d->dy0 == -1 && d->dx0 == -1 &&
           d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == 9 * CH && d->grid_h == d->in_h && d->grid_w == d->in_w &&
           d->out_h == d->in_h && d->out_w == d->in_w && d->oy_mul == 1 && d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0;
}

int sp_head128_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                      void* stream) {
    SP_REQUIRE(!residual, "sp_conv3x3_direct: the 128 -> J head kernel takes no residual");
    const long long elems = (long long)d->batch * d->in_h * d->in_w * CH;
    const long long out_elems = (long long)d->batch * d->c_out * d->in_h * d->in_w;
    SP_REQUIRE(elems < (1ll << 29) && out_elems < (1ll << 29), "sp_conv3x3_direct: tensor too large");
    if (sp_name_query_active()) { sp_name_query_set("conv3x3_c128_head_kernel"); return SP_OK; }
    HeadArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = reinterpret_cast<float*>(y);
    a.H = d->in_h; a.W = d->in_w; a.batch = d->batch; a.c_out = d->c_out;
    a.relu = (d->flags & SP_CONV_RELU) ? 1 : 0;
    a.x_bytes = (int)(elems * 2); a.w_bytes = NH * 9 * CH * 2; a.y_bytes = (int)(out_elems * 4);
    a.tiles_x = (d->in_w + THC - 1) / THC; a.tiles_y = (d->in_h + THR - 1) / THR;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_conv3x3_direct: too many tiles");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c128_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSH);
    if (e != hipSuccess) { sp_set_error("sp_conv3x3_direct: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDSH, hipGetErrorString(e)); return SP_ELAUNCH; }
    const long long grid = tiles < cus ? tiles : cus;
    hipLaunchKernelGGL(conv3x3_c128_head_kernel, dim3((unsigned)grid), dim3(THREADS_H), LDSH, (hipStream_t)stream, a);
    return sp_check_launch("conv3x3_c128_head_kernel");
}
