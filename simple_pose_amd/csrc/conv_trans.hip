// conv_trans.hip - HRNet's transition1 as ONE launch (bf16): both convolutions that read layer1's 256-channel output
// (nets/pose_hrnet.py:327-366 `_make_transition_layer`, used at :431-437):
//     transition1.0 = conv3x3(256 -> 32, stride 1, pad 1) + BN + ReLU      (the high-resolution branch)
//     transition1.1 = conv3x3(256 -> 64, stride 2, pad 1) + BN + ReLU      (the new half-resolution branch)
//
// Why: through the implicit GEMM these two launches moved 965 + 452 MB for a 201 MB input at bs=128 and took 169 + 139 us
// (profiles/r03_hrnet_w32_bf16_traffic.json): N = 32 / 64 output channels give a gathered input pixel too little MFMA work per byte, and
// every pixel was gathered once per tap (and once per launch).  Here a persistent 8-wave workgroup owns a tile of 32 x 16 pixels: the
// 34 x 18 halo is staged ONCE per 64-channel chunk in LDS and serves all nine taps of BOTH convolutions (the stride-2 outputs of the tile
// are its 16 x 8 even-centred pixels: the same halo); the weights of one (chunk, tap) - 32 + 64 rows of 128 bytes - stream through a
// double-buffered LDS stage, requested three stages ahead into a ring of register sets (they do not depend on the tile, so the stream
// runs on across tiles).  Per stage a wave runs 8 + 4 v_mfma_f32_32x32x16_bf16: two 32-pixel row tiles of the stride-1 output against
// the 32-channel filter, one 32-pixel x 32-channel tile of the stride-2 output.
//
// Reduction order: (channel chunk, tap, channel) - NOT the implicit GEMM's (tap, channel): results agree with the per-conv program to
// fp32 accumulation order (bf16 outputs: equal or one rounding apart), not bit for bit; tests/test_gpu_parity.py pins both convolutions
// against float64 on the same bf16 operands at the bar of every other bf16 conv.
#include "sp_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TR_ = 32, TC_ = 16;                 // tile: rows x columns of the stride-1 output
constexpr int HR_ = TR_ + 2, HC_ = TC_ + 2;       // halo 34 x 18
constexpr int NPIX = HR_ * HC_;                   // 612 pixels
constexpr int CIN = 256, CHUNK = 64, NCH = CIN / CHUNK;
constexpr int NA = 32, NB = 64;                   // output channels of the two convolutions
constexpr int X_BYTES = NPIX * 144;               // one 64-channel chunk of the halo, 144-byte pixel rows (see xoff): 88,128 B
constexpr int W_STAGE = (NA + NB) * 128;          // weights of one (chunk, tap): 12,288 B
constexpr int LDS_BYTES = X_BYTES + 2 * W_STAGE;  // 112,704 B
constexpr int NHP = (NPIX * 8 + 511) / 512;       // 16-byte halo pieces per thread (10)
constexpr int NSTAGE = NCH * 9;                   // 36 weight stages per tile
constexpr unsigned OOB = 0x80000000u;

struct TransArgs {
    const void* x;        // NHWC bf16 [B,H,W,256]
    const void* wa;       // packed [32][k_pad] bf16, K = (tap, channel)
    const void* wb;       // packed [64][k_pad]
    const float* sa; const float* ha;     // folded BatchNorm of transition1.0 (scale, shift) [32]
    const float* sb; const float* hb;     // ... of transition1.1 [64]
    void* ya;             // NHWC bf16 [B,H,W,32]
    void* yb;             // NHWC bf16 [B,H/2,W/2,64]
    int B, H, W, k_pad;
    int tiles_y, tiles_x;
    int x_bytes, wa_bytes, wb_bytes, ya_bytes, yb_bytes;
};

// Halo pixel P = hy * 18 + hx is a row of 144 bytes in LDS: 128 bytes of channels + 16 of padding.  LINEAR on purpose: a tap is then a
// compile-time byte offset of the ds_read (an XOR swizzle made every (tap, row tile) address a register of its own: 108 of them, spilled),
// and the padding alone spreads the banks - 16 consecutive pixels reading the same 16-byte piece (one ds_read_b128 beat of the stride-1
// fragments) start 36 banks apart modulo 64: 0, 36, 8, 44, ... all distinct.  The stride-2 fragments (every other pixel of two halo rows two
// apart per beat) meet two-way conflicts: they are a fifth of the reads.
constexpr int PIXS = 144;
__device__ __forceinline__ int xoff(int P, int pc) { return P * PIXS + (pc << 4); }
__device__ __forceinline__ int woff(int n, int pc) { return (n << 7) + ((pc ^ ((n >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512, 2) void hrnet_transition1_kernel(const TransArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Xs = smem;
    unsigned char* const Ws = smem + X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int per_img = p.tiles_y * p.tiles_x;
    const int ntiles = p.B * per_img;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t war = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wa), (short)0, p.wa_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wb), (short)0, p.wb_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yar = __builtin_amdgcn_make_buffer_rsrc(p.ya, (short)0, p.ya_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ybr = __builtin_amdgcn_make_buffer_rsrc(p.yb, (short)0, p.yb_bytes, 0x00020000);

    // ---- this thread's halo pieces: piece q = tid + 512 i -> pixel q >> 3, 16-byte piece q & 7 (coordinates recomputed per use: ten
    //      multiply-shifts against keeping thirty registers live across the MFMA loop) ----
    const int h_pc = tid & 7;                                   // (512 is a multiple of 8: the same piece index for every i)
    u32x4 hv[NHP];
    auto req_halo = [&](int tile, int c) {
        const int b = tile / per_img, rem = tile - b * per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const int q = tid + 512 * i, P = q >> 3;
            const int hy = (P * 3641) >> 16, hx = P - hy * HC_;           // P / 18 for P < 1,024
            const int iy = ty * TR_ - 1 + hy, ix = tx * TC_ - 1 + hx;
            const bool ok = tile < ntiles && q < NPIX * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * CIN + c * CHUNK + h_pc * 8) * 2) : OOB, 0, 0);
        }
    };
    auto put_halo = [&]() {
#pragma unroll
        for (int i = 0; i < NHP; ++i) {
            const int q = tid + 512 * i, P = q >> 3;
            if (q < NPIX * 8) *reinterpret_cast<u32x4*>(Xs + xoff(P, h_pc)) = hv[i];
        }
    };

    // ---- this thread's weight pieces of a stage: rows 0..31 = the stride-1 filter, 32..95 = the stride-2 filter; 768 pieces of 16 B ----
    const int w_row0 = tid >> 3, w_pc = tid & 7;                // rows 0..63: waves 0-3 read wa, waves 4-7 rows 0..31 of wb (wave-uniform)
    const int w_lds0 = woff(w_row0, w_pc);
    const int w_lds1 = woff(64 + (tid >> 3), w_pc);             // tid < 256: rows 64..95 = rows 32..63 of wb
    const unsigned w_src0 = (unsigned)(((w_row0 & 31) * p.k_pad + w_pc * 8) * 2);
    const unsigned w_src1 = (unsigned)(((32 + (tid >> 3)) * p.k_pad + w_pc * 8) * 2);
    u32x4 wv[3][2];
    auto req_w = [&](int s, u32x4* dst) {                       // s: stage of a tile (weights do not depend on the tile)
        const int c = s / 9, t = s - c * 9;
        const unsigned koff = (unsigned)((t * CIN + c * CHUNK) * 2);
        dst[0] = (wave < 4) ? __builtin_amdgcn_raw_buffer_load_b128(war, w_src0 + koff, 0, 0)
                            : __builtin_amdgcn_raw_buffer_load_b128(wbr, w_src0 + koff, 0, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(wbr, tid < 256 ? w_src1 + koff : OOB, 0, 0);
    };
    auto put_w = [&](int buf, const u32x4* src) {
        *reinterpret_cast<u32x4*>(Ws + buf * W_STAGE + w_lds0) = src[0];
        if (tid < 256) *reinterpret_cast<u32x4*>(Ws + buf * W_STAGE + w_lds1) = src[1];
    };

    // ---- fragment coordinates: stride-1 rows 4w + 2m + (fr >> 4), column fr & 15; stride-2 tile (mt = w >> 1, nt = w & 1):
    //      row 4 mt + (fr >> 3), column fr & 7 of the 16 x 8 half-resolution tile ----
    const int a_r = 4 * wave + (fr >> 4), a_c = fr & 15;
    const int mt = wave >> 1, nt = wave & 1;
    const int b_r = 2 * (4 * mt + (fr >> 3)), b_c = 2 * (fr & 7);          // halo coordinates of tap (0, 0)
    const int x_a = xoff(a_r * HC_ + a_c, fh);                             // + tap offset + 32 j (k step) [+ two halo rows: second row tile]
    const int x_b = xoff(b_r * HC_ + b_c, fh);
    int w_fa[4], w_fb[4];                                                  // weight fragments: row fr (stride-1 filter) / 32 + 32 nt + fr
#pragma unroll
    for (int j = 0; j < 4; ++j) { w_fa[j] = woff(fr, 2 * j + fh); w_fb[j] = woff(NA + nt * 32 + fr, 2 * j + fh); }

    f32x16 acc0, acc1, accb;
    float* const tr = reinterpret_cast<float*>(smem) + wave * 1024;          // epilogue transpose: 4 KB per wave inside the (then idle) halo

    // prologue: first halo chunk and the first three weight stages in flight, stage 0 in LDS
    int tile = blockIdx.x;
    req_halo(tile, 0);
    req_w(0, wv[0]); req_w(1, wv[1]); req_w(2, wv[2]);
    put_w(0, wv[0]);
    req_w(3, wv[0]);

    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; accb[r] = 0.f; }
        // chunks in pairs: 18 stages = a whole number of periods of the weight ring (3 register sets, 2 LDS buffers), so inside the pair
        // every ring index is a compile-time constant while the pair loop itself stays rolled (fully unrolled, the 36 stages' addresses
        // spilled 181 registers)
#pragma unroll 1
        for (int cp = 0; cp < NCH / 2; ++cp) {
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const int c = 2 * cp + ch;
            put_halo();                                                     // (every wave has passed the barrier that ended the halo's last use)
            if (c + 1 < NCH) req_halo(tile, c + 1); else req_halo(tile + gridDim.x, 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int s = ch * 9 + t;                                   // stage within the pair: compile-time after unrolling
                const int sg = cp * 18 + s;                                 // stage within the tile (weights)
                // weights of stage s + 1 -> the other LDS buffer (its last readers finished before the barrier that ended stage s - 1);
                // then the register set is free for stage s + 4
                put_w((s + 1) & 1, wv[(s + 1) % 3]);
                req_w(sg + 4 < NSTAGE ? sg + 4 : sg + 4 - NSTAGE, wv[(s + 1) % 3]);
                if (t == 0) __syncthreads();                                // the new halo chunk is visible
                const unsigned char* Wb = Ws + (s & 1) * W_STAGE;
                const int ty = t / 3, tx = t - 3 * ty;
                const int tap = (ty * HC_ + tx) * PIXS;                     // compile-time: folded into the ds_read offsets
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 wA = *reinterpret_cast<const u32x4*>(Wb + w_fa[j]);
                    const u32x4 wB = *reinterpret_cast<const u32x4*>(Wb + w_fb[j]);
                    const u32x4 xa0 = *reinterpret_cast<const u32x4*>(Xs + x_a + tap + j * 32);
                    const u32x4 xa1 = *reinterpret_cast<const u32x4*>(Xs + x_a + tap + 2 * HC_ * PIXS + j * 32);
                    const u32x4 xb = *reinterpret_cast<const u32x4*>(Xs + x_b + tap + j * 32);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa0), __builtin_bit_cast(bf16x8, wA), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa1), __builtin_bit_cast(bf16x8, wA), acc1, 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xb), __builtin_bit_cast(bf16x8, wB), accb, 0, 0, 0);
                }
                __syncthreads();            // stage s is done everywhere: its weight buffer and (after the chunk's last tap) the halo may be overwritten
            }
          }
        }
        // ---- epilogue: BatchNorm + ReLU, bf16, 16-byte NHWC stores.  C/D map: column = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        //      (pixel of the tile); through the wave's private LDS slice so that a lane owns 8 consecutive channels of one pixel ----
        const int b = tile / per_img, rem = tile - b * per_img;
        const int tyy = rem / p.tiles_x, txx = rem - tyy * p.tiles_x;
        const int chunk = lane & 3;
        auto store_tile = [&](const f32x16& acc, const float* sc_p, const float* sh_p, int which) {
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = sc_p[chunk * 8 + e]; sh[e] = sh_p[chunk * 8 + e]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[r];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2);                     // pixel of the 32-pixel tile
                float v[8];
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4 * e4);
                    v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                }
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = v[e] * sc[e] + sh[e];
                    o8[e] = (__bf16)(y > 0.f ? y : 0.f);
                }
                if (which < 2) {                                           // stride-1 output: tile rows 4w + 2 which + (row >> 4), column row & 15
                    const int oy = tyy * TR_ + 4 * wave + 2 * which + (row >> 4), ox = txx * TC_ + (row & 15);
                    const unsigned off = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * NA + chunk * 8) * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yar, off, 0, 0);
                } else {                                                   // stride-2 output: rows 4 mt + (row >> 3), column row & 7 of the half-resolution tile
                    const int H2 = p.H >> 1, W2 = p.W >> 1;
                    const int oy = tyy * (TR_ / 2) + 4 * mt + (row >> 3), ox = txx * (TC_ / 2) + (row & 7);
                    const unsigned off = (oy < H2 && ox < W2) ? (unsigned)((((b * H2 + oy) * W2 + ox) * NB + nt * 32 + chunk * 8) * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), ybr, off, 0, 0);
                }
            }
        };
        store_tile(acc0, p.sa, p.ha, 0);
        store_tile(acc1, p.sa, p.ha, 1);
        store_tile(accb, p.sb + nt * 32, p.hb + nt * 32, 2);
        __syncthreads();                                                    // the transposes are done before the next tile's halo lands
    }
}

int trans_cus() {
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cache[dev]) {
        hipDeviceProp_t prop;
        cache[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cache[dev];
}

}  // namespace

extern "C" int sp_hrnet_transition1_ok(int c_in, int h, int w) { return c_in == CIN && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0; }

extern "C" int sp_hrnet_transition1(const void* x, int batch, int h, int w, const void* wa_packed, int k_pad, const float* scale_a, const float* shift_a,
                                    const void* wb_packed, const float* scale_b, const float* shift_b, void* y_hi, void* y_lo, void* stream) {
    SP_REQUIRE(x && wa_packed && wb_packed && scale_a && shift_a && scale_b && shift_b && y_hi && y_lo, "sp_hrnet_transition1: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "sp_hrnet_transition1: even input size needed (got %dx%d)", h, w);
    SP_REQUIRE(k_pad == 9 * CIN, "sp_hrnet_transition1: both filters are 3x3 on 256 channels packed (tap, channel): k_pad must be %d (got %d)", 9 * CIN, k_pad);
    const long long in_elems = (long long)batch * h * w * CIN;
    SP_REQUIRE(in_elems < (1ll << 30), "sp_hrnet_transition1: input too large for 32-bit buffer offsets");
    TransArgs a;
    a.x = x; a.wa = wa_packed; a.wb = wb_packed; a.sa = scale_a; a.ha = shift_a; a.sb = scale_b; a.hb = shift_b; a.ya = y_hi; a.yb = y_lo;
    a.B = batch; a.H = h; a.W = w; a.k_pad = k_pad;
    a.tiles_y = (h + TR_ - 1) / TR_; a.tiles_x = (w + TC_ - 1) / TC_;
    a.x_bytes = (int)(in_elems * 2); a.wa_bytes = NA * k_pad * 2; a.wb_bytes = NB * k_pad * 2;
    a.ya_bytes = (int)((long long)batch * h * w * NA * 2); a.yb_bytes = (int)((long long)batch * (h / 2) * (w / 2) * NB * 2);
    if (sp_name_query_active()) { sp_name_query_set("hrnet_transition1_kernel"); return SP_OK; }
    static bool opted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&hrnet_transition1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            sp_set_error("sp_hrnet_transition1: hipFuncSetAttribute(max dynamic LDS = %d) failed", LDS_BYTES);
            return SP_ELAUNCH;
        }
        opted[dev] = true;
    }
    const int tiles = batch * a.tiles_y * a.tiles_x;
    const int grid = tiles < trans_cus() ? tiles : trans_cus();
    hipLaunchKernelGGL(hrnet_transition1_kernel, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return sp_check_launch("hrnet_transition1_kernel");
}
