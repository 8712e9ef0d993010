// conv_direct.hip - 3x3 stride-1 convolutions for SMALL channel counts (bf16; 32 -> 32 and 64 -> 64 channels: HRNet's two high-resolution
// branches, ResNet's layer1 conv2).  The implicit GEMM of conv_igemm.hip gathers every input pixel once per tap, i.e. it moves 9x the input
// through L2 -> CU; with 32 / 64 output channels there is too little MFMA work per byte to hide that (measured: 43 us per 32-channel layer at
// bs=128 against a 15 us HBM bound).  Here a workgroup stages the halo of its 16 x 24-pixel output tile ONCE in LDS (linear, padded pixel rows:
// a tap is a compile-time offset) and forms the nine taps from there, with the filter staged once per persistent workgroup.  Same reduction
// order as the implicit GEMM (tap-major, channel-minor, one v_mfma_f32_32x32x16_bf16 chain per output) -> bit-identical results, which is how
// they are tested.  Replaces the conv3x3 + BN [+ residual] + ReLU of nets/pose_hrnet.py BasicBlocks and nets/pose_resnet_dconv.py:112-120.
#include "sp_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CI = 32, CO = 32;
constexpr unsigned OOB = 0x80000000u;

struct DirectArgs {
    const void* x;        // NHWC bf16 [B,H,W,32]
    const void* w;        // packed [n_pad >= 32][k_pad = 320] bf16, K = (tap, channel)
    const float* scale;
    const float* shift;
    const void* res;      // NHWC bf16 [B,H,W,32] or null
    void* y;              // NHWC bf16 [B,H,W,32]
    int H, W, k_pad, batch;
    int tiles_x, tiles_y;
    int relu;
    int x_bytes, w_bytes;
};

// ---- 64 -> 64 channels (HRNet's second branch: 64 of its 293 convs at 32x24; ResNet-50 layer1.*.conv2 at 64x48) -------------------------------
// Round 4 structure: 8 waves, 16 x 24 pixel tiles.  (The round-2 kernel - 16 x 8 tiles, ONE wave per SIMD: 72 KB filter + two halo buffers = one
// 4-wave workgroup per CU, 300 registers, 1.5 LDS reads per MFMA - was LDS- and latency-bound: 26 us per HRNet layer at bs=128 against an 8 us
// HBM floor; this one measures 19.8 us, same bits.)  The implicit GEMM spends 26 us on such a layer (9 short K tiles per workgroup, every
// input pixel gathered nine times).  A workgroup of 8 waves owns 16 rows x 24
// columns (HRNet's second branch is 32 x 24 per image: two tiles; ResNet's layer1 64 x 48: eight): wave (g, h) multiplies rows 4g..4g+3 - three
// row tiles of 4 x 8 pixels - with channels 32h..32h+31: one B fragment feeds three MFMAs, 1.33 reads per MFMA at two waves per SIMD.  The halo
// (18 x 26 pixels) sits in LDS as LINEAR 144-byte pixel rows (a tap = a compile-time offset; 128 B + 16 B of padding spreads 8 consecutive
// pixels over the banks) with the row stride padded to 128 mod 256 bytes so that the two pixel rows of a 16-lane read beat interleave; the
// whole filter [tap][n][128 B] (XOR-swizzled 16-byte pieces, 72 KB) is loaded once per persistent workgroup; the next tile's halo waits in
// registers.  Same reduction order as the implicit GEMM (one MFMA chain per output, tap-major, channel-minor): bit-identical results.
constexpr int C6 = 64;
constexpr int T7R = 16, T7C = 24;
constexpr int H7R = T7R + 2, H7C = T7C + 2;              // 18 x 26 halo
constexpr int PX7 = 144;                                  // bytes per halo pixel
constexpr int RS7 = H7C * PX7 + 224;                      // 3,968 = 128 (mod 256) bytes per halo row
constexpr int X7_BYTES = H7R * RS7;                       // 71,424
constexpr int W7_BYTES = 9 * 64 * 128;                    // 73,728
constexpr int LDS7 = X7_BYTES + W7_BYTES;                 // 145,152
constexpr int NH7 = (H7R * H7C * 8 + 511) / 512;          // 16-byte halo pieces per thread (8)
__device__ __forceinline__ int w7off(int n, int pc) { return (n << 7) + ((pc ^ ((n >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512, 2) void conv3x3_c64_tile_kernel(const DirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    unsigned char* const Xs = smem7;
    unsigned char* const Ws = smem7 + X7_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int per_img = p.tiles_x * p.tiles_y;
    const int ntiles = per_img * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.x_bytes, 0x00020000);

    const int h_pc = tid & 7;
    u32x4 hv[NH7];
    auto req_halo = [&](int tile) {
        const int b = tile / per_img, rem = tile - b * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int i = 0; i < NH7; ++i) {
            const int q = tid + 512 * i, P = q >> 3;
            const int hy = (P * 2521) >> 16, hx = P - hy * H7C;          // P / 26 for P < 512
            const int iy = ty * T7R - 1 + hy, ix = tx * T7C - 1 + hx;
            const bool ok = tile < ntiles && q < H7R * H7C * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * C6 + h_pc * 8) * 2) : OOB, 0, 0);
        }
    };
    auto put_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NH7; ++i) {
            const int q = tid + 512 * i, P = q >> 3;
            const int hy = (P * 2521) >> 16, hx = P - hy * H7C;
            if (q < H7R * H7C * 8) *reinterpret_cast<u32x4*>(Xs + hy * RS7 + hx * PX7 + (h_pc << 4)) = hv[i];
        }
    };
    int tile = blockIdx.x;
    req_halo(tile);
    // ---- the whole filter -> LDS, once per workgroup: piece q = (tap, n, pc) <- W[n][tap * 64 + pc * 8 .. + 8] ----
    {   // (512 threads x 9 = the 4,608 pieces: thread t takes piece (pc, n) = (t & 7, t >> 3) of every tap; all nine loads in flight at once -
        //  a load -> store loop paid nine HBM / L2 round trips one after the other, and with one tile per workgroup at bs=128 nothing hid them)
        const int pc = tid & 7, n = tid >> 3;
        u32x4 wv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wv[tap] = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)((n * p.k_pad + tap * 64 + pc * 8) * 2), 0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<u32x4*>(Ws + tap * 8192 + w7off(n, pc)) = wv[tap];
    }
    const int g = wave >> 1, h = wave & 1;
    const int a_r = 4 * g + (fr >> 3), a_c = fr & 7;              // this lane's pixel of a row tile: tile row a_r, column 8 m + a_c
    const int x_a = a_r * RS7 + a_c * PX7 + (fh << 4);             // + tap offset + 8 m pixels + 32 j
    int w_f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w_f[j] = w7off(32 * h + fr, 2 * j + fh);
    float sc[8], sh[8];                                            // this lane's 8 output channels in the epilogue: 32 h + 8 chunk + e
    const int chunk = lane & 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = p.scale ? p.scale[32 * h + chunk * 8 + e] : 1.f; sh[e] = p.shift ? p.shift[32 * h + chunk * 8 + e] : 0.f; }
    float* const tr = reinterpret_cast<float*>(smem7) + wave * 1024;   // epilogue transpose: 4 KB per wave inside the (then idle) halo

    for (; tile < ntiles; tile += gridDim.x) {
        put_halo();
        __syncthreads();                                          // halo (and, first time round, the filter) visible; last tile's transposes done
        req_halo(tile + gridDim.x);
        const int b = tile / per_img, rem = tile - b * per_img;
        const int tyy = rem / p.tiles_x, txx = rem - tyy * p.tiles_x;
        // output offsets and residual of this lane's 6 stores (3 row tiles x 2 passes of 16 pixels x 4 chunks), in flight under the MFMAs
        unsigned ooff[3][2];
        u32x4 rv[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2);                  // pixel of the 32-pixel row tile
                const int oy = tyy * T7R + 4 * g + (row >> 3), ox = txx * T7C + 8 * m + (row & 7);
                ooff[m][it] = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C6 + 32 * h + chunk * 8) * 2) : OOB;
                rv[m][it] = u32x4{0u, 0u, 0u, 0u};
                if (p.res) rv[m][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, ooff[m][it], 0, 0);
            }
        f32x16 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * RS7 + (tap % 3) * PX7;           // compile-time
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 wb = *reinterpret_cast<const u32x4*>(Ws + tap * 8192 + w_f[j]);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const u32x4 xa = *reinterpret_cast<const u32x4*>(Xs + x_a + toff + m * 8 * PX7 + j * 32);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, wb), acc[m], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                          // every wave is done with the halo: it becomes the transpose scratch
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[m][r];
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
                    const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[m][it]);
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff[m][it], 0, 0);
            }
        }
        __syncthreads();                                          // the transposes are done before the next halo lands
    }
}

// ---- 32 -> 32 channels (HRNet's high-resolution branch: 64 of its 293 convs) in the tile structure of the 64-channel kernel above (round 4) ------
// (The round-1 kernel - 8 x 16 tiles, the filter in 72 registers of every lane, an XOR-swizzled 64-byte pixel image - ran at 20.1-20.5 us per
// layer at bs=128 against a 15 us HBM floor, bound by its own instruction stream; this one measures 18.2 us, same bits.)
// 4 waves, 16 x 24 pixel tiles, wave g = rows 4g..4g+3 as three 4 x 8-pixel row tiles (one B fragment, read from LDS, feeds three MFMAs); halo as
// linear 80-byte pixel rows (64 B + 16 B padding), row stride 128 mod 256 bytes; filter [tap][n][80 B] in LDS; two workgroups per CU.
constexpr int PX8 = 80;
constexpr int RS8 = H7C * PX8 + 96;                       // 2,176 = 128 (mod 256)
constexpr int X8_BYTES = H7R * RS8;                       // 39,168
constexpr int W8_BYTES = 9 * 32 * 80;                     // 23,040
constexpr int LDS8 = X8_BYTES + W8_BYTES;                 // 62,208
constexpr int NH8 = (H7R * H7C * 4 + 255) / 256;          // 16-byte halo pieces per thread (8)

__global__ __launch_bounds__(256, 2) void conv3x3_c32_tile_kernel(const DirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    unsigned char* const Xs = smem8;
    unsigned char* const Ws = smem8 + X8_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int per_img = p.tiles_x * p.tiles_y;
    const int ntiles = per_img * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.x_bytes, 0x00020000);

    const int h_pc = tid & 3;
    u32x4 hv[NH8];
    auto req_halo = [&](int tile) {
        const int b = tile / per_img, rem = tile - b * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int i = 0; i < NH8; ++i) {
            const int q = tid + 256 * i, P = q >> 2;
            const int hy = (P * 2521) >> 16, hx = P - hy * H7C;          // P / 26 for P < 512
            const int iy = ty * T7R - 1 + hy, ix = tx * T7C - 1 + hx;
            const bool ok = tile < ntiles && q < H7R * H7C * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * CI + h_pc * 8) * 2) : OOB, 0, 0);
        }
    };
    auto put_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NH8; ++i) {
            const int q = tid + 256 * i, P = q >> 2;
            const int hy = (P * 2521) >> 16, hx = P - hy * H7C;
            if (q < H7R * H7C * 4) *reinterpret_cast<u32x4*>(Xs + hy * RS8 + hx * PX8 + (h_pc << 4)) = hv[i];
        }
    };
    int tile = blockIdx.x;
    req_halo(tile);
    {   // filter -> LDS: 9 taps x 32 rows x 4 pieces = 1,152 pieces; thread t: piece (t & 3) of row (t >> 2) & 31 of taps (t >> 7) + 2 i
        const int pc = tid & 3, n = (tid >> 2) & 31, t0 = tid >> 7;
        u32x4 wv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tap = t0 + 2 * i;
            wv[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, tap < 9 ? (unsigned)((n * p.k_pad + tap * 32 + pc * 8) * 2) : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tap = t0 + 2 * i;
            if (tap < 9) *reinterpret_cast<u32x4*>(Ws + (tap * 32 + n) * 80 + (pc << 4)) = wv[i];
        }
    }
    const int g = wave;
    const int a_r = 4 * g + (fr >> 3), a_c = fr & 7;
    const int x_a = a_r * RS8 + a_c * PX8 + (fh << 4);             // + tap offset + 8 m pixels + 32 j
    const int w_b = fr * 80 + (fh << 4);                           // + tap * 2560 + 32 j
    // Epilogue in registers (round 5).  The MFMA runs with the FILTER as its first operand, so an accumulator holds 16 output channels of ONE
    // pixel (column fr = pixel, rows (r & 3) + 8 (r >> 2) + 4 fh = channels); lanes l and l + 32 hold complementary channel quads of the same
    // pixel, and eight v_permlane32_swap turn them into two runs of 8 consecutive channels per lane (fh = 0: channels 0-7 and 16-23, fh = 1:
    // 8-15 and 24-31): 16-byte residual loads and stores straight from registers.  The round-4 epilogue went through LDS (16 ds_write_b32 +
    // 4 ds_read_b128 per 32 x 32 tile and a workgroup barrier per pixel tile); operand order does not change an MFMA's sum (same products,
    // same k order), so the bits are the implicit GEMM's as before (`test_direct_3x3_kernels_are_bit_identical_to_the_implicit_gemm`).
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = 16 * q + 8 * fh + e;
            sc[q][e] = p.scale ? p.scale[ch] : 1.f;
            sh[q][e] = p.shift ? p.shift[ch] : 0.f;
        }

    for (; tile < ntiles; tile += gridDim.x) {
        put_halo();
        __syncthreads();
        req_halo(tile + gridDim.x);
        const int b = tile / per_img, rem = tile - b * per_img;
        const int tyy = rem / p.tiles_x, txx = rem - tyy * p.tiles_x;
        unsigned ooff[3];
        u32x4 rv[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int oy = tyy * T7R + 4 * g + (fr >> 3), ox = txx * T7C + 8 * m + (fr & 7);
            ooff[m] = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * CO + 8 * fh) * 2) : OOB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rv[m][q] = u32x4{0u, 0u, 0u, 0u};
                if (p.res) rv[m][q] = __builtin_amdgcn_raw_buffer_load_b128(rr, ooff[m] + 32u * q, 0, 0);
            }
        }
        f32x16 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * RS8 + (tap % 3) * PX8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 wb = *reinterpret_cast<const u32x4*>(Ws + tap * 2560 + w_b + j * 32);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const u32x4 xa = *reinterpret_cast<const u32x4*>(Xs + x_a + toff + m * 8 * PX8 + j * 32);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb), __builtin_bit_cast(bf16x8, xa), acc[m], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // upper 32 lanes of the first register <-> lower 32 lanes of the second.  (The elements go through scalar temporaries:
                    // `__builtin_bit_cast(unsigned, vec[k])` on an ext_vector ELEMENT reads element 0 with this hipcc - ROCm 7.2, found here.)
                    const float lo = acc[m][8 * h + i], hi = acc[m][8 * h + 4 + i];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                    acc[m][8 * h + i] = __uint_as_float(sw[0]);
                    acc[m][8 * h + 4 + i] = __uint_as_float(sw[1]);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[m][8 * q + e] * sc[q][e] + sh[q][e];
                if (p.res) {
                    const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[m][q]);
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
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff[m] + 32u * q, 0, 0);
            }
        }
        __syncthreads();                                        // every wave is done with this tile's halo before the next one is put down
    }
}

}  // namespace

// Eligibility: what the engine checks before it offers this kernel to the tuner (sp_conv3x3_direct_ok) and what the launch re-checks.
bool sp_tile128_ok(const sp_conv_desc* d);               // conv_tile128.hip: the 128-channel tile kernel behind the same entry point
int sp_tile128_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual,
                      void* y, void* stream);
bool sp_head128_ok(const sp_conv_desc* d);               // conv_head128.hip: 128 -> J <= 32 channels, fp32 NCHW out (the DUC head's last layer)
int sp_head128_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual,
                      void* y, void* stream);

static bool direct_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;            // grouped convolutions run on the implicit GEMM only
    if (d && d->c_in == 128) return sp_tile128_ok(d) || sp_head128_ok(d);
    if (!d || (d->c_in != 32 && d->c_in != 64)) return false;
    const int c = d->c_in, kp = c == 32 ? 320 : 576;          // k_pad: 9 taps x c, rounded to whole 64-element K tiles
    return (d->flags & SP_CONV_BF16) && !(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) &&
           d->c_out == c && d->out_c == c && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) &&
           d->dy0 == -1 && d->dx0 == -1 && d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == kp &&
           d->n_pad >= c && d->grid_h == d->in_h && d->grid_w == d->in_w && d->out_h == d->in_h && d->out_w == d->in_w && d->oy_mul == 1 &&
           d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0;
}

extern "C" int sp_conv3x3_direct_ok(const sp_conv_desc* d) { return direct_ok(d) ? 1 : 0; }

extern "C" int sp_conv3x3_direct(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                                 const void* residual, void* y, void* stream) {
    SP_REQUIRE(d && x && w_packed && y, "sp_conv3x3_direct: null pointer");
    SP_REQUIRE(direct_ok(d), "sp_conv3x3_direct: needs a bf16 3x3 stride-1 pad-1 convolution with 32 -> 32, 64 -> 64 or 128 -> 128 channels (NHWC bf16 out), or "
               "128 -> at most 32 channels with fp32 NCHW output");
    SP_REQUIRE(d->batch > 0, "sp_conv3x3_direct: bad batch");
    if (d->c_in == 128 && sp_head128_ok(d)) return sp_head128_launch(d, x, w_packed, scale, shift, residual, y, stream);
    if (d->c_in == 128) return sp_tile128_launch(d, x, w_packed, scale, shift, residual, y, stream);
    const long long elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    SP_REQUIRE(elems < (1ll << 29), "sp_conv3x3_direct: tensor too large");
    DirectArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.k_pad = d->k_pad; a.batch = d->batch;
    a.relu = (d->flags & SP_CONV_RELU) ? 1 : 0;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    if (d->c_in == 64) {
        if (sp_name_query_active()) { sp_name_query_set("conv3x3_c64_tile_kernel"); return SP_OK; }
        a.tiles_x = (d->in_w + T7C - 1) / T7C; a.tiles_y = (d->in_h + T7R - 1) / T7R;
        const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
        SP_REQUIRE(tiles < (1ll << 31), "sp_conv3x3_direct: too many tiles");
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS7);
        if (e != hipSuccess) { sp_set_error("sp_conv3x3_direct: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDS7, hipGetErrorString(e)); return SP_ELAUNCH; }
        const long long grid = tiles < cus ? tiles : cus;
        hipLaunchKernelGGL(conv3x3_c64_tile_kernel, dim3((unsigned)grid), dim3(512), LDS7, (hipStream_t)stream, a);
        return sp_check_launch("conv3x3_c64_tile_kernel");
    }
    if (sp_name_query_active()) { sp_name_query_set("conv3x3_c32_tile_kernel"); return SP_OK; }
    a.tiles_x = (d->in_w + T7C - 1) / T7C; a.tiles_y = (d->in_h + T7R - 1) / T7R;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_conv3x3_direct: too many tiles");
    // persistent workgroups, two per CU (62 KB of LDS, 190 VGPRs): the filter is staged once per workgroup, the next tile's halo waits in registers
    const long long grid = tiles < 512 ? tiles : 512;
    hipLaunchKernelGGL(conv3x3_c32_tile_kernel, dim3((unsigned)grid), dim3(256), LDS8, (hipStream_t)stream, a);
    return sp_check_launch("conv3x3_c32_tile_kernel");
}
