// conv_dual.hip - the tail of a stage-opening Bottleneck with 64 mid channels as ONE launch (bf16):
//
//     y = relu( bn3(conv1x1_{64->256}(t)) + bn_d(conv1x1_{64->256}(x)) )        nets/pose_resnet_dconv.py:120-131 with :99-103's projection shortcut
//
// (layer1.0 of the ResNet pose nets and of HRNet at 64x48: t = the block's 3x3 output, x = the block input, both 64 channels).  Run as two
// launches the shortcut's 256-channel tensor is written (201 MB at bs=128) and read again as conv3's residual (201 MB): 703 MB for the pair,
// both launches on the HBM roof (profiles/r05_dconv_bf16_kernel_stats.csv: 68.6 + 101 us).  Here both 1x1 products of a 128-pixel tile are
// accumulated side by side and meet in the epilogue: 301 MB (t and x in, y out).
//
// Same bits as the two launches: each product is the per-conv kernels' MFMA chain (4 k steps of 16 in order, one v_mfma_f32_32x32x16_bf16
// chain per output), the shortcut value is rounded to bf16 exactly where the two-launch program stores it, and the sum / ReLU / rounding
// are conv3's epilogue, element by element - which is how it is tested (test_dual_pointwise_tail_equals_the_two_launches_bitwise).
//
// One persistent 8-wave workgroup per CU.  Both weight matrices sit in LDS in fragment order (2 x 32 KB) for the whole launch; the two
// A tiles of a tile (128 pixels x 128 B each) arrive by LDS-DMA (buffer_load ... lds, 1-KiB pieces of 8 pixels, conv_ring.hip's swizzled row
// image) into one of two buffers while the previous tile is multiplied; wave w = row block w >> 1 x column half w & 1 (4 column blocks of 32
// channels, both products: 128 accumulator registers); per column block the combined 32x32 block is transposed through the wave's LDS slab so
// that a lane stores 8 consecutive channels of one pixel (16-byte stores, 64 bytes per pixel and instruction).
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int DP_TM = 128;                         // pixels per tile
constexpr int DP_K = 64, DP_N = 256;
constexpr int DP_WB = 4 * 8 * 2 * 32 * 16;         // one weight matrix in fragment order [k step][column block][k half][column] 16-B = 32,768 B
constexpr int DP_AB = DP_TM * 128;                 // one A tile: [128 pixels][128 B] = 16,384 B
constexpr int DP_TRB = 32 * 32 * 4;                // per wave: fp32 transpose slab of one 32x32 block
constexpr int DP_LDS = 2 * DP_WB + 2 * 2 * DP_AB + 8 * DP_TRB;   // 163,840 B: the CU's whole LDS
constexpr unsigned OOB = 0x80000000u;

struct DualArgs {
    const void* a_main;   // t  [rows][64] bf16
    const void* a_short;  // x  [rows][64] bf16
    const void* w_main;   // packed [>=256][64] bf16
    const void* w_short;  // packed [>=256][64] bf16
    const float *s_main, *h_main, *s_short, *h_short;   // folded BatchNorms (scale, shift); null = 1 / 0
    void* y;              // [rows][256] bf16
    int rows, tiles, relu;
    int a_bytes, w_bytes, y_bytes;
};

__device__ __forceinline__ u32x4 dual_rsrc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000u;
    return r;
}

// one LDS-DMA piece (conv_ring.hip dma16: inline asm on purpose, see there): 64 lanes x 16 bytes, lane l's bytes from rsrc + voff (zeros when out
// of range) to LDS at lds_addr + 16 l
__device__ __forceinline__ void dual_dma16(unsigned lds_addr, unsigned voff, u32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc)
                 : "memory");
}

__global__ __launch_bounds__(512, 2) void dual_pw_bf16_kernel(const DualArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemd[];
    unsigned char* const Wl = smemd;                              // [2 products][DP_WB]
    unsigned char* const At = smemd + 2 * DP_WB;                  // [2 buffers][2 products][DP_AB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tr = reinterpret_cast<float*>(smemd + 2 * DP_WB + 4 * DP_AB + wave * DP_TRB);
    const int fr = lane & 31, fh = lane >> 5;
    const int G = gridDim.x;

    const u32x4 ar0 = dual_rsrc(p.a_main, p.a_bytes), ar1 = dual_rsrc(p.a_short, p.a_bytes);
    const __amdgpu_buffer_rsrc_t wr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_main), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_short), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);

    // ---- both weight matrices -> LDS in fragment order, once per workgroup: fragment (ks, nb, kh, n) = W[nb*32 + n][ks*16 + kh*8 .. +8] ----
    // (read in MEMORY order - consecutive lanes take consecutive 16-byte pieces of the 128-byte rows - and scattered on the LDS side; reading in fragment
    //  order gathered 16 bytes from each of 32 rows per wave-instruction: conv_bneck.hip's A/B of the same change, profiles/r06_bneck8_ab.txt)
    static_assert(DP_K == 64, "eight 16-byte pieces per row");
    for (int q = tid; q < 2 * DP_WB / 16; q += 512) {
        const int g = q / (DP_WB / 16), pi = q - g * (DP_WB / 16);
        const int row = pi >> 3, c = pi & 7;                     // output channel, 16-byte column
        const int r = (c >> 1) * 512 + (row >> 5) * 64 + (c & 1) * 32 + (row & 31);
        const unsigned off = (unsigned)(pi * 16);
        *reinterpret_cast<u32x4*>(Wl + (g * (DP_WB / 16) + r) * 16) = g ? __builtin_amdgcn_raw_buffer_load_b128(wr1, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(wr0, off, 0, 0);
    }

    // ---- A tiles by LDS-DMA: a tile's two images are 32 pieces of 1 KiB (8 pixels x 128 B); wave w issues pieces w and w + 8 of both ----
    const unsigned at_lds = (unsigned)(size_t)(lds_void_t*)At;
    const int pr = lane >> 3, pc = lane & 7;
    auto request = [&](int tile, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int piece = wave + 8 * h;
            const int row = 8 * piece + pr;                                   // row of the tile; chunk position pc holds source chunk pc ^ swizzle
            const long long m = (long long)tile * DP_TM + row;
            const unsigned src = m < p.rows ? (unsigned)((m * DP_K + ((pc ^ ((row >> 1) & 7)) << 3)) * 2) : OOB;
            const unsigned dst = at_lds + (unsigned)(buf * 2 * DP_AB + piece * 1024);
            dual_dma16(dst, src, ar0);
            dual_dma16(dst + DP_AB, src, ar1);
        }
    };

    // this lane's folded BatchNorms in the ACCUMULATOR layout (channel = column fr of the block): 4 column blocks x (main, shortcut)
    const int mb = wave >> 1, chalf = wave & 1;
    float s3v[4], h3v[4], sdv[4], hdv[4];
#pragma unroll
    for (int nbl = 0; nbl < 4; ++nbl) {
        const int ch = chalf * 128 + nbl * 32 + fr;
        s3v[nbl] = p.s_main ? p.s_main[ch] : 1.f;  h3v[nbl] = p.h_main ? p.h_main[ch] : 0.f;
        sdv[nbl] = p.s_short ? p.s_short[ch] : 1.f; hdv[nbl] = p.h_short ? p.h_short[ch] : 0.f;
    }
    int aofs[4];                                                     // A fragment of k step ks: row mb*32 + fr, chunk 2 ks + fh (swizzled)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aofs[ks] = (mb * 32 + fr) * 128 + (((2 * ks + fh) ^ ((fr >> 1) & 7)) << 4);
    const unsigned char* const wfrag = Wl + (fh * 32 + fr) * 16 + (chalf * 4) * 1024;    // + g * DP_WB + ks * 8192 + nbl * 1024

    int tile = blockIdx.x;
    if (tile < p.tiles) request(tile, 0);
    int buf = 0;
    for (; tile < p.tiles; tile += G, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of `tile` (and its stores of the previous tile) are done
        __syncthreads();                                           // ... everybody's; (first tile: the weights too); buffer buf ^ 1 is free again
        if (tile + G < p.tiles) request(tile + G, buf ^ 1);
        const unsigned char* const a0 = At + buf * 2 * DP_AB;
        f32x16 acc[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][nbl][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 af[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const u32x4*>(a0 + g * DP_AB + aofs[ks]);
#pragma unroll
            for (int nbl = 0; nbl < 4; ++nbl)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 bq = *reinterpret_cast<const u32x4*>(wfrag + g * DP_WB + ks * 8192 + nbl * 1024);
                    acc[g][nbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[ks]), __builtin_bit_cast(bf16x8, bq), acc[g][nbl], 0, 0, 0);
                }
        }
        // ---- epilogue: per column block, combine in the accumulator layout, transpose through the slab, 16-byte stores ----
        const long long m0 = (long long)tile * DP_TM + mb * 32;
#pragma unroll
        for (int nbl = 0; nbl < 4; ++nbl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = acc[1][nbl][r] * sdv[nbl] + hdv[nbl];          // the shortcut's epilogue ...
                const float rs = (float)(__bf16)sv;                             // ... and its bf16 store, as the two-launch program rounds it
                float v = acc[0][nbl][r] * s3v[nbl] + h3v[nbl];                 // conv3's epilogue: scale / shift, + residual, ReLU
                v += rs;
                if (p.relu) v = v > 0.f ? v : 0.f;
                tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = v;
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2), chunk = lane & 3;
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4);
                bf16x8 o8;
                o8[0] = (__bf16)t0[0]; o8[1] = (__bf16)t0[1]; o8[2] = (__bf16)t0[2]; o8[3] = (__bf16)t0[3];
                o8[4] = (__bf16)t1[0]; o8[5] = (__bf16)t1[1]; o8[6] = (__bf16)t1[2]; o8[7] = (__bf16)t1[3];
                const long long m = m0 + row;
                const unsigned off = m < p.rows ? (unsigned)((m * DP_N + chalf * 128 + nbl * 32 + chunk * 8) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, off, 0, 0);
            }
        }
    }
}

bool dual_ok(long long rows, int c_main, int c_short, int c_out) {
    return rows > 0 && rows * DP_N < (1ll << 30) && c_main == DP_K && c_short == DP_K && c_out == DP_N;
}

}  // namespace

extern "C" int sp_dual_pw_bf16_ok(int64_t rows, int c_main, int c_short, int c_out) { return dual_ok(rows, c_main, c_short, c_out) ? 1 : 0; }

extern "C" int sp_dual_pw_bf16(const void* a_main, const void* w_main_packed, const float* scale_main, const float* shift_main, const void* a_short,
                               const void* w_short_packed, const float* scale_short, const float* shift_short, void* y, int64_t rows, int c_main,
                               int c_short, int c_out, int relu, void* stream) {
    SP_REQUIRE(a_main && w_main_packed && a_short && w_short_packed && y, "sp_dual_pw_bf16: null pointer");
    SP_REQUIRE(dual_ok(rows, c_main, c_short, c_out), "sp_dual_pw_bf16: two bf16 1x1 stride-1 products of 64 channels each into 256 (got %d + %d -> %d, %lld rows)",
               c_main, c_short, c_out, (long long)rows);
    SP_REQUIRE(y != a_main && y != a_short, "sp_dual_pw_bf16: y must not alias an input");
    if (sp_name_query_active()) { sp_name_query_set("dual_pw_bf16_kernel"); return SP_OK; }
    DualArgs a;
    a.a_main = a_main; a.a_short = a_short; a.w_main = w_main_packed; a.w_short = w_short_packed;
    a.s_main = scale_main; a.h_main = shift_main; a.s_short = scale_short; a.h_short = shift_short; a.y = y;
    a.rows = (int)rows; a.tiles = (int)((rows + DP_TM - 1) / DP_TM); a.relu = relu ? 1 : 0;
    a.a_bytes = (int)(rows * DP_K * 2); a.w_bytes = DP_N * DP_K * 2; a.y_bytes = (int)(rows * DP_N * 2);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dual_pw_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS);
    if (e != hipSuccess) { sp_set_error("sp_dual_pw_bf16: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", DP_LDS, hipGetErrorString(e)); return SP_ELAUNCH; }
    const int grid = a.tiles < cus ? a.tiles : cus;
    hipLaunchKernelGGL(dual_pw_bf16_kernel, dim3(grid), dim3(512), DP_LDS, (hipStream_t)stream, a);
    return sp_check_launch("dual_pw_bf16_kernel");
}
