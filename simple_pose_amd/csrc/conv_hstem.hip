// conv_hstem.hip - HRNet's stem as ONE launch, bf16 compute: fp32 NCHW image -> conv1 3x3 s2 p1 (3 -> 64) + bn1 + relu -> conv2 3x3 s2 p1
// (64 -> 64) + bn2 + relu -> NHWC bf16 at a quarter of the resolution (nets/pose_hrnet.py:419-425 `x = relu(bn1(conv1(x))); x = relu(bn2(conv2(x)))`).
//
// As three launches (layout change, two implicit GEMMs: 30 + 76 + 52 us at bs = 128) conv1's 128 x 96 x 64 map is written to HBM and read
// back (201 MB each way).  Here a persistent 4-wave workgroup owns 8 x 8 output pixels (136 us per launch: the conv2 weights in registers
// cost one wave per SIMD, so nothing hides the LDS round trips - a modest win, kept because it is bit-identical and removes 0.4 GB of traffic):
//   * the 35 x 36 image patch behind them goes to LDS once as NHWC4 bf16 (fetched from the fp32 planes; the next tile's patch is requested
//     before the current tile's MFMAs);
//   * conv1 on the 17 x 17 pixels conv2 needs (GEMM columns m = cy * 18 + cx, 10 column tiles; positions outside conv1's output are
//     conv2's zero padding and are stored as 0) with the patch as the B operand and the packed conv1 weights as the A operand of
//     v_mfma_f32_32x32x16_bf16: the accumulator then holds 4 consecutive CHANNELS of one pixel per register group, so BatchNorm + ReLU +
//     bf16 go out as 8-byte stores into a pixel-major LDS tile - exactly the layout conv2's fragments want (16 bytes = 8 channels of a tap);
//   * conv2 from that tile: wave (wm, wn) = 32 output pixels x 32 channels, its 36 weight fragments in registers for the life of the
//     workgroup, one ds_read_b128 per MFMA; result through a small pixel-major staging tile, then 16-byte coalesced stores.
// LDS pitches (34 dwords per pixel, 624 per row) make both the 8-byte conv1 stores and the stride-2 16-byte conv2 reads conflict-free.
//
// Bits: identical to sp_nchw_to_nhwc4_bf16 -> sp_conv2d_fwd(conv1) -> sp_conv2d_fwd(conv2).  Same packed weights, same instruction, same k
// positions in the same order (conv1: k-step ky = 4 x slots x 4 channels of the "pixel pair" view, x slot q = pixel 2 ox - 2 + q, the all-zero
// fourth step dropped; conv2: k-step j = tap j / 4, channels 16 (j % 4) ..); swapping the MFMA operands transposes the accumulator, not the sums.
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct HStemArgs {
    const float* x;                 // [B][3][H][W]
    const __bf16* w1;               // packed conv1 [64][k1_pad] (K order: ky, pair 0..1, sub 0..1, channel 0..3)
    const __bf16* w2;               // packed conv2 [64][576]   (K order: ty, tx, channel)
    const float *sc1, *sh1, *sc2, *sh2;
    __bf16* y;                      // [B][H2][W2][64]
    int batch, H, W, H1, W1, H2, W2;
    int tiles_y, tiles_x, n_tiles, k1_pad;
    unsigned x_bytes;
};

constexpr int T2 = 8;                       // output tile edge (conv2 pixels)
constexpr int C1 = 2 * T2 + 1;              // conv1 rows / columns behind it (17)
constexpr int C1S = C1 + 1;                 // + one dummy column: 18 GEMM columns per conv1 row
constexpr int M1 = C1 * C1S;                // 306
constexpr int MT1 = (M1 + 31) / 32;         // 10
constexpr int PR = 2 * (C1 - 1) + 3;        // 35 patch rows
constexpr int PC = 2 * (C1 - 1) + 4;        // 36 patch columns (the pair view's 4 x slots)
constexpr int NPIX = PR * PC;               // 1260
constexpr int NPF = (NPIX + 255) / 256;     // 5
constexpr int PP = 34;                      // conv1 tile: dwords per pixel (64 bf16 = 32 dwords + 2)
constexpr int RP = 624;                     // conv1 tile: dwords per row of 18 pixels (>= 18 * 34 = 612; 2 * RP = 32 mod 64)
constexpr int PATCH_BYTES = ((NPIX + 8) * 8 + 15) / 16 * 16;      // + slack: the dummy column's fragments read past the last pixel
constexpr int OUT1_BYTES = C1 * RP * 4;
constexpr int STAGE_BYTES = 64 * PP * 4;
constexpr int LDS_BYTES = PATCH_BYTES + OUT1_BYTES + STAGE_BYTES;
constexpr unsigned OOB = 0x80000000u;

__global__ __launch_bounds__(256, 1) void hrnet_stem_kernel(const HStemArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* const patch = smem;
    unsigned* const out1 = reinterpret_cast<unsigned*>(smem + PATCH_BYTES);
    unsigned* const stage = reinterpret_cast<unsigned*>(smem + PATCH_BYTES + OUT1_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const int plane4 = p.H * p.W * 4;

    // ---- weight fragments (the A operands): conv1 for both channel tiles, conv2 for this wave's channel tile ----
    u32x4 fw1[3][2], fw2[36];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int i = 0; i < 2; ++i) fw1[kh][i] = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)(i * 32 + fr) * p.k1_pad + 16 * kh + 8 * fh);
#pragma unroll
    for (int j = 0; j < 36; ++j) fw2[j] = *reinterpret_cast<const u32x4*>(p.w2 + (size_t)(wn * 32 + fr) * 576 + 16 * j + 8 * fh);
    // BatchNorm per accumulator register: channel of register r = (r & 3) + 8 (r >> 2) + 4 fh (+ 32 per channel tile)
    float s1[2][16], h1[2][16], s2[16], h2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
        for (int i = 0; i < 2; ++i) { s1[i][r] = p.sc1[i * 32 + c]; h1[i][r] = p.sh1[i * 32 + c]; }
        s2[r] = p.sc2[wn * 32 + c]; h2[r] = p.sh2[wn * 32 + c];
    }

    // ---- patch prefetch (as conv_stem.hip): this thread's pixels idx = tid + 256 i, three planes each, zero outside the image ----
    int prel[NPF];
    unsigned prc[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / PC, c = idx - r * PC;
        prel[i] = r * p.W + c;
        prc[i] = idx < NPIX ? (unsigned)((r << 8) | c) : 0xffffff00u;
    }
    float pf[NPF][3];
    auto origin = [&](int t, int& b, int& oy0, int& ox0) {
        const int per = p.tiles_y * p.tiles_x;
        b = t / per;
        const int r = t - b * per;
        const int ty = r / p.tiles_x;
        oy0 = ty * T2;
        ox0 = (r - ty * p.tiles_x) * T2;
    };
    auto prefetch = [&](int t) {
        int b, oy0, ox0;
        origin(t, b, oy0, ox0);
        const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 4;
        const int base = (b * 3 * p.H + iy0) * p.W + ix0;
        const bool inside = iy0 >= 0 && ix0 >= 0 && iy0 + PR <= p.H && ix0 + PC <= p.W;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int gy = iy0 + (int)(prc[i] >> 8), gx = ix0 + (int)(prc[i] & 255);
            const bool ok = inside ? ((i + 1) * 256 <= NPIX || tid + 256 * i < NPIX) : ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W);
            const unsigned off = ok ? (unsigned)((base + prel[i]) * 4) : OOB;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) pf[i][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, ch * plane4, 0));
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NPIX) {
                const bf16x4 v = {(__bf16)pf[i][0], (__bf16)pf[i][1], (__bf16)pf[i][2], (__bf16)0.f};
                *reinterpret_cast<u32x2*>(patch + idx * 8) = __builtin_bit_cast(u32x2, v);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < p.n_tiles) prefetch(tile);
    for (; tile < p.n_tiles; tile += gridDim.x) {
        int b, oy0, ox0;
        origin(tile, b, oy0, ox0);
        park();
        __syncthreads();                                   // patch complete; the previous tile's staging reads are done
        const int next = tile + gridDim.x;
        if (next < p.n_tiles) prefetch(next);

        // ---- conv1 on the 17 x 18 pixels behind the tile: column tile mt = 32 pixels, both channel tiles ----
        for (int mt = wave; mt < MT1; mt += 4) {
            const int m = mt * 32 + fr;
            const int mc = m < M1 ? m : 0;
            const int cy = mc / C1S, cx = mc - C1S * cy;
            const unsigned char* bp = patch + ((2 * cy) * PC + 2 * cx + 2 * fh) * 8;
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const u32x4 px = *reinterpret_cast<const u32x4*>(bp + kh * (PC * 8));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw1[kh][i]), __builtin_bit_cast(bf16x8, px), acc[i], 0, 0, 0);
            }
            // conv1 pixel of this lane in the image; outside conv1's output it is conv2's zero padding
            const int r1 = 2 * oy0 - 1 + cy, c1 = 2 * ox0 - 1 + cx;
            const bool valid = m < M1 && cx < C1 && (unsigned)r1 < (unsigned)p.H1 && (unsigned)c1 < (unsigned)p.W1;
            if (m < M1) {
                unsigned* dst = out1 + cy * RP + cx * PP;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[i][4 * g + e] * s1[i][4 * g + e] + h1[i][4 * g + e];
                            v = v > 0.f ? v : 0.f;
                            o[e] = (__bf16)(valid ? v : 0.f);
                        }
                        *reinterpret_cast<u32x2*>(dst + (i * 32 + 8 * g + 4 * fh) / 2) = __builtin_bit_cast(u32x2, o);
                    }
            }
        }
        __syncthreads();                                   // conv1 tile complete

        // ---- conv2: wave (wm, wn) = output pixels 32 wm .. + 31 (4 rows of 8) x channels 32 wn .. + 31 ----
        {
            const int oy = 4 * wm + (fr >> 3), ox = fr & 7;
            const unsigned* ap = out1 + (2 * oy) * RP + (2 * ox) * PP + 4 * fh;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 36; ++j) {
                const int tap = j >> 2, q = j & 3, ty = tap / 3, tx = tap - 3 * ty;
                const u32x4 a = *reinterpret_cast<const u32x4*>(ap + ty * RP + tx * PP + 8 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw2[j]), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
            }
            unsigned* dst = stage + (wm * 32 + fr) * PP + wn * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * g + e] * s2[4 * g + e] + h2[4 * g + e];
                    v = v > 0.f ? v : 0.f;
                    o[e] = (__bf16)v;
                }
                *reinterpret_cast<u32x2*>(dst + (8 * g + 4 * fh) / 2) = __builtin_bit_cast(u32x2, o);
            }
        }
        __syncthreads();                                   // staging complete
        // ---- 64 pixels x 128 bytes out: thread = (pixel tid / 4, 32-byte quarter tid % 4) ----
        {
            const int pix = tid >> 2, qt = tid & 3;
            const int oy = oy0 + (pix >> 3), ox = ox0 + (pix & 7);
            if (oy < p.H2 && ox < p.W2) {
                const unsigned* src = stage + pix * PP + qt * 8;
                const u32x2 a0 = *reinterpret_cast<const u32x2*>(src), a1 = *reinterpret_cast<const u32x2*>(src + 2);
                const u32x2 a2 = *reinterpret_cast<const u32x2*>(src + 4), a3 = *reinterpret_cast<const u32x2*>(src + 6);
                u32x4* d = reinterpret_cast<u32x4*>(p.y + (((size_t)b * p.H2 + oy) * p.W2 + ox) * 64 + qt * 16);
                d[0] = u32x4{a0[0], a0[1], a1[0], a1[1]};
                d[1] = u32x4{a2[0], a2[1], a3[0], a3[1]};
            }
        }
    }
}

}  // namespace

extern "C" int sp_hrnet_stem_ok(int batch, int h, int w) {
    return batch > 0 && h >= 4 && w >= 4 && h % 2 == 0 && w % 2 == 0 && (long long)batch * 3 * h * w * 4 < (1ll << 31) ? 1 : 0;
}

extern "C" int sp_hrnet_stem(const float* x, const void* w1_packed, int k1_pad, const float* scale1, const float* shift1, const void* w2_packed,
                             const float* scale2, const float* shift2, void* y, int batch, int h, int w, void* stream) {
    SP_REQUIRE(x && w1_packed && scale1 && shift1 && w2_packed && scale2 && shift2 && y, "sp_hrnet_stem: null pointer");
    SP_REQUIRE(sp_hrnet_stem_ok(batch, h, w), "sp_hrnet_stem: batch %d of %d x %d images (even sizes, 32-bit offsets)", batch, h, w);
    SP_REQUIRE(k1_pad >= 48 && k1_pad % 8 == 0, "sp_hrnet_stem: k1_pad %d (the packed 3x3 pixel-pair stem has K >= 48)", k1_pad);
    HStemArgs a;
    a.x = x; a.w1 = reinterpret_cast<const __bf16*>(w1_packed); a.w2 = reinterpret_cast<const __bf16*>(w2_packed);
    a.sc1 = scale1; a.sh1 = shift1; a.sc2 = scale2; a.sh2 = shift2; a.y = reinterpret_cast<__bf16*>(y);
    a.batch = batch; a.H = h; a.W = w;
    a.H1 = (h + 2 - 3) / 2 + 1; a.W1 = (w + 2 - 3) / 2 + 1;
    a.H2 = (a.H1 + 2 - 3) / 2 + 1; a.W2 = (a.W1 + 2 - 3) / 2 + 1;
    a.tiles_y = (a.H2 + T2 - 1) / T2; a.tiles_x = (a.W2 + T2 - 1) / T2;
    a.n_tiles = batch * a.tiles_y * a.tiles_x;
    a.k1_pad = k1_pad;
    a.x_bytes = (unsigned)((long long)batch * 3 * h * w * 4);
    if (sp_name_query_active()) {
        sp_name_query_set("hrnet_stem_kernel");
        return SP_OK;
    }
    static bool opted[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hrnet_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            sp_set_error("hrnet stem: hipFuncSetAttribute(max dynamic LDS = %d) failed on device %d", LDS_BYTES, dev);
            return SP_ELAUNCH;
        }
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        opted[dev] = true;
    }
    const int slots = cus[dev];                            // one workgroup per CU (398 registers per lane: one wave per SIMD)
    const int rounds = (a.n_tiles + slots - 1) / slots;
    const int grid = (a.n_tiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(hrnet_stem_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
    return sp_check_launch("hrnet_stem_kernel");
}
