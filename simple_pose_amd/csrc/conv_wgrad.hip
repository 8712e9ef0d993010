// conv_wgrad.hip - weight gradients of the conv / transposed-conv family on the fp32 matrix cores.
//
//   dW[n][k] = sum_m G[m][n] * A[m][k]          (reduction over the pixels m of one launch grid)
//
// G is a plain NHWC tensor [M][N] (conv: dz, the gradient at the conv output; transposed conv: the layer INPUT x),
// A is the implicit im2col of the other tensor, gathered exactly like the forward kernel gathers its A operand
// (conv: x with the layer's taps/stride/padding; transposed conv k4s2p1: dy with 4x4 taps, stride 2, pad 1).
// Both operands have the reduction index m as their slow (row) index, so LDS tiles are [32 m][128] and the MFMA
// operands are read one float per lane (ds_read2_b32: k-pairs (m0+h, m0+2+h) of two 32-wide tiles per instruction).
// The pixel range is split over blockIdx.y; every split writes its partial [n_pad][k_pad] slab with plain coalesced
// stores, and wgrad_reduce_kernel sums the slabs in a fixed order and scatters into the reference's weight layout
// (Conv2d [O,I,kh,kw], ConvTranspose2d [I,O,kh,kw]) - deterministic, no float atomics.
#include "sp_common.h"

namespace {

struct WgradArgs {
    const void* g;    // [M][n_ld]  fp32 (or bf16: conv_wgrad_bf16_kernel)
    const void* a;    // NHWC [B][in_h][in_w][c_in], same dtype
    float* slab;      // [splits][n_rows][k_pad]
    int M, n_rows, n_ld;   // n_rows = rows of dW computed (multiple of 128 via padding of the slab), n_ld = row stride of g
    int n_valid;           // real N (columns of g beyond it are not read)
    int in_h, in_w, c_in;
    int grid_h, grid_w;
    int taps_h, taps_w, k_pad;
    int stride, dy0, dy_step, dx0, dx_step;
    int rows_per_split;    // multiple of 32
    int g_bytes, a_bytes;
};

constexpr int WB = 128;   // tile of dW: 128 (n) x 128 (k)
constexpr int WBK = 32;   // pixels per step

__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Gs = smem;                  // [2][32][128]
    float* As = smem + 2 * WBK * WB;   // [2][32][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_k = p.k_pad / WB;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * WB, k0 = tk * WB;
    const int m_begin = blockIdx.y * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), (short)0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), (short)0, p.a_bytes, 0x00020000);

    // staging: thread -> 16-byte chunk q = tid % 32 of rows tid/32 + 8 i (i = 0..3) of both tiles
    const int q = tid & 31, srow = tid >> 5;
    // this thread's k chunk is fixed for the whole launch: decode its tap / channel once
    const int kk = k0 + q * 4;
    const int tap = kk / p.c_in, c_off = kk - tap * p.c_in;
    const int ty = tap / p.taps_w, tx = tap - ty * p.taps_w;
    const bool tap_ok = ty < p.taps_h;
    const int ddy = ty * p.dy_step + p.dy0, ddx = tx * p.dx_step + p.dx0;
    const bool gcol_ok = (n0 + q * 4) < p.n_valid;
    const int ghw = p.grid_h * p.grid_w;

    u32x4 sg[4], sa[4];
    auto load_tiles = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + srow + 8 * i;
            const bool row_ok = m < m_end;
            sg[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, (row_ok && gcol_ok) ? (unsigned)((m * p.n_ld + n0 + q * 4) * 4) : OOB, 0, 0);
            const int b = m / ghw, rem = m - b * ghw;
            const int gy = rem / p.grid_w, gx = rem - gy * p.grid_w;
            const int iy = gy * p.stride + ddy, ix = gx * p.stride + ddx;
            const bool ok = row_ok && tap_ok && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            sa[i] = __builtin_amdgcn_raw_buffer_load_b128(ar, ok ? (unsigned)((((b * p.in_h + iy) * p.in_w + ix) * p.c_in + c_off) * 4) : OOB, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(Gs + (buf * WBK + srow + 8 * i) * WB + q * 4) = sg[i];
            *reinterpret_cast<u32x4*>(As + (buf * WBK + srow + 8 * i) * WB + q * 4) = sa[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += WBK) {
            const bool more = m0 + WBK < m_end;
            if (more) load_tiles(m0 + WBK);
            const float* gs = Gs + cur * WBK * WB + wr * 64 + fr;
            const float* as = As + cur * WBK * WB + wc * 64 + fr;
#pragma unroll
            for (int kp = 0; kp < WBK / 2; ++kp) {  // MFMA k = 2 pixels: lane half fh takes pixel 2 kp + fh
                const int row = 2 * kp + fh;
                const float g0 = gs[row * WB], g1 = gs[row * WB + 32];
                const float a0 = as[row * WB], a1 = as[row * WB + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, a0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, a1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, a0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, a1, acc[1][1], 0, 0, 0);
            }
            if (more) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    // partial slab: rows n (accumulator rows), columns k; C/D map col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)
    float* out = p.slab + ((size_t)blockIdx.y * p.n_rows + n0 + wr * 64) * p.k_pad + k0 + wc * 64 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                out[(size_t)row * p.k_pad + n * 32] = acc[i][n][r];
            }
}

// ---- bf16 operands -------------------------------------------------------------------------------------------------------
// Same decomposition with v_mfma_f32_32x32x16_bf16 (k = 16 pixels per MFMA).  Both operands are pixel-major in memory
// ([m][n] and [m][k]) while the MFMA wants 8 consecutive k (= pixels) per lane for ONE column, i.e. a transposed read:
// ds_read_b64_tr_b16 delivers, per 16-lane group, a 4-row x 16-column block column-major - lane i gets column i of the
// 4 rows - so two of them build the 8-pixel operand of a lane straight from the row-major LDS tile (no transposing stores).
// LDS rows are padded to 320 B so that the four rows a 16-lane group addresses and the two column halves of a 32-lane
// half fall on disjoint banks.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WROW = 160;  // bf16 elements per LDS tile row (128 + 32 pad) = 320 B

__device__ __forceinline__ bf16x8 tr_operand(const __bf16* tile, int row0, int col0, int lane) {
    // lane -> (16-lane group g: column half g&1, k half g>>1), inside the group lane 4q+p addresses row q, columns 4p..4p+3
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const __bf16* base = tile + (row0 + 8 * (g >> 1) + q) * WROW + col0 + 16 * (g & 1) + 4 * pp;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * WROW));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Gs = reinterpret_cast<__bf16*>(smem);   // [2][32][WROW]
    __bf16* As = Gs + 2 * WBK * WROW;               // [2][32][WROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_k = p.k_pad / WB;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
    const int n0 = tn * WB, k0 = tk * WB;
    const int m_begin = blockIdx.y * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.g), (short)0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), (short)0, p.a_bytes, 0x00020000);

    // staging: 32 rows x 16 chunks (8 bf16) per operand; thread -> chunk q = tid % 16 of rows tid/16 + 16 i (i = 0, 1)
    const int q = tid & 15, srow = tid >> 4;
    const int kk = k0 + q * 8;
    const int tap = kk / p.c_in, c_off = kk - tap * p.c_in;
    const int ty = tap / p.taps_w, tx = tap - ty * p.taps_w;
    const bool tap_ok = ty < p.taps_h;
    const int ddy = ty * p.dy_step + p.dy0, ddx = tx * p.dx_step + p.dx0;
    const bool gcol_ok = (n0 + q * 8) < p.n_valid;
    const int ghw = p.grid_h * p.grid_w;

    u32x4 sg[2], sa[2];
    auto load_tiles = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + srow + 16 * i;
            const bool row_ok = m < m_end;
            sg[i] = __builtin_amdgcn_raw_buffer_load_b128(gr, (row_ok && gcol_ok) ? (unsigned)((m * p.n_ld + n0 + q * 8) * 2) : OOB, 0, 0);
            const int b = m / ghw, rem = m - b * ghw;
            const int gy = rem / p.grid_w, gx = rem - gy * p.grid_w;
            const int iy = gy * p.stride + ddy, ix = gx * p.stride + ddx;
            const bool ok = row_ok && tap_ok && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            sa[i] = __builtin_amdgcn_raw_buffer_load_b128(ar, ok ? (unsigned)((((b * p.in_h + iy) * p.in_w + ix) * p.c_in + c_off) * 2) : OOB, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(Gs + (buf * WBK + srow + 16 * i) * WROW + q * 8) = sg[i];
            *reinterpret_cast<u32x4*>(As + (buf * WBK + srow + 16 * i) * WROW + q * 8) = sa[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    if (m_begin < m_end) {
        load_tiles(m_begin);
        store_tiles(0);
        __syncthreads();
        int cur = 0;
        for (int m0 = m_begin; m0 < m_end; m0 += WBK) {
            const bool more = m0 + WBK < m_end;
            if (more) load_tiles(m0 + WBK);
            const __bf16* gs = Gs + cur * WBK * WROW;
            const __bf16* as = As + cur * WBK * WROW;
#pragma unroll
            for (int ks = 0; ks < WBK / 16; ++ks) {   // MFMA k = 16 pixels
                const bf16x8 g0 = tr_operand(gs, 16 * ks, wr * 64, lane), g1 = tr_operand(gs, 16 * ks, wr * 64 + 32, lane);
                const bf16x8 a0 = tr_operand(as, 16 * ks, wc * 64, lane), a1 = tr_operand(as, 16 * ks, wc * 64 + 32, lane);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, a0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, a1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, a0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, a1, acc[1][1], 0, 0, 0);
            }
            if (more) store_tiles(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    float* out = p.slab + ((size_t)blockIdx.y * p.n_rows + n0 + wr * 64) * p.k_pad + k0 + wc * 64 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                out[(size_t)row * p.k_pad + n * 32] = acc[i][n][r];
            }
}

// dst[n*s_n + c*s_c + (ty*kw + tx)] (+)= sum_s slab[s][n][(ty*taps_w + tx)*c_in + c]   for n < n_valid, c < c_valid, tx < kw
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int splits, int n_rows, int k_pad, int n_valid, int c_in, int c_valid,
                                    int taps_h, int taps_w, int kw, long long s_n, long long s_c, float* __restrict__ dst, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % k_pad);
        const int n = (int)(i / k_pad);
        const int tap = k / c_in, c = k - tap * c_in;
        const int ty = tap / taps_w, tx = tap - ty * taps_w;
        if (n >= n_valid || c >= c_valid || ty >= taps_h || tx >= kw) continue;
        // fixed summation order (deterministic); 4 independent partial chains keep 4 loads in flight
        const size_t stride = (size_t)n_rows * k_pad;
        const float* src = slab + (size_t)n * k_pad + k;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = 0;
        for (; sp + 4 <= splits; sp += 4) {
            s0 += src[(size_t)sp * stride]; s1 += src[(size_t)(sp + 1) * stride];
            s2 += src[(size_t)(sp + 2) * stride]; s3 += src[(size_t)(sp + 3) * stride];
        }
        for (; sp < splits; ++sp) s0 += src[(size_t)sp * stride];
        dst[n * s_n + c * s_c + ty * kw + tx] = (s0 + s1) + (s2 + s3);
    }
}

}  // namespace

extern "C" int sp_conv2d_wgrad(const sp_conv_desc* d, const void* g, int g_channels, const void* a, int n_valid, int c_valid, int kw_valid,
                               int64_t dst_stride_n, int64_t dst_stride_c, float* dw, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    SP_REQUIRE(d && (d->stride_x == 0 || d->stride_x == d->stride), "sp_conv2d_wgrad: separate x / y strides are a forward-only feature");
    SP_REQUIRE(d && g && a && dw && workspace, "sp_conv2d_wgrad: null pointer");
    const bool bf16 = d->flags & SP_CONV_BF16;
    const int es = bf16 ? 2 : 4, epc = 16 / es;
    SP_REQUIRE(d->c_in > 0 && d->c_in % epc == 0 && d->k_pad % 32 == 0 && d->k_pad >= d->taps_h * d->taps_w * d->c_in,
               "sp_conv2d_wgrad: bad c_in / k_pad");
    SP_REQUIRE(g_channels > 0 && g_channels % epc == 0 && n_valid > 0 && n_valid <= g_channels, "sp_conv2d_wgrad: bad g_channels/n_valid");
    SP_REQUIRE(c_valid > 0 && c_valid <= d->c_in && kw_valid > 0 && kw_valid <= d->taps_w, "sp_conv2d_wgrad: bad c_valid/kw_valid");
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    const long long a_elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    SP_REQUIRE(M > 0 && M * g_channels < (1ll << 29) && a_elems < (1ll << 29), "sp_conv2d_wgrad: tensor too large");
    const int k_pad128 = (d->k_pad + 127) / 128 * 128;
    const int n_rows = (n_valid + 127) / 128 * 128;
    const int tiles = (n_rows / 128) * (k_pad128 / 128);
    // split the pixel range so that ~768 (fp32) / ~384 (bf16: the MFMA part is 4x shorter, slab traffic dominates) workgroups
    // exist; each split a multiple of 32 pixels
    const int target = bf16 ? 384 : 768;
    long long splits = (target + tiles - 1) / tiles;
    const long long max_splits = (M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long rows_per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
    splits = (M + rows_per_split - 1) / rows_per_split;
    const long long need = splits * n_rows * (long long)k_pad128 * 4;
    SP_REQUIRE(need <= workspace_bytes, "sp_conv2d_wgrad: workspace too small (%lld B needed, %lld given)", need, (long long)workspace_bytes);

    WgradArgs p;
    p.g = g; p.a = a; p.slab = reinterpret_cast<float*>(workspace);
    p.M = (int)M; p.n_rows = n_rows; p.n_ld = g_channels; p.n_valid = n_valid;
    p.in_h = d->in_h; p.in_w = d->in_w; p.c_in = d->c_in; p.grid_h = d->grid_h; p.grid_w = d->grid_w;
    p.taps_h = d->taps_h; p.taps_w = d->taps_w; p.k_pad = k_pad128;
    p.stride = d->stride; p.dy0 = d->dy0; p.dy_step = d->dy_step; p.dx0 = d->dx0; p.dx_step = d->dx_step;
    p.rows_per_split = (int)rows_per_split;
    p.g_bytes = (int)(M * g_channels * es); p.a_bytes = (int)(a_elems * es);
    hipStream_t s = (hipStream_t)stream;
    if (bf16) {
        const size_t lds = (size_t)4 * WBK * WROW * 2;
        hipLaunchKernelGGL(conv_wgrad_bf16_kernel, dim3(tiles, (unsigned)splits), dim3(256), lds, s, p);
    } else {
        const size_t lds = (size_t)4 * WBK * WB * sizeof(float);
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles, (unsigned)splits), dim3(256), lds, s, p);
    }
    const long long total = (long long)n_rows * k_pad128;
    long long gsz = (total + 255) / 256;
    if (gsz > 2048) gsz = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)gsz), dim3(256), 0, s, p.slab, (int)splits, n_rows, k_pad128, n_valid, d->c_in, c_valid,
                       d->taps_h, d->taps_w, kw_valid, dst_stride_n, dst_stride_c, dw, total);
    return sp_check_launch("conv_wgrad");
}
