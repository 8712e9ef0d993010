// conv_direct.hip - 3x3 stride-1 convolution for SMALL channel counts (bf16, 32 -> 32 channels: HRNet's high-resolution branch).
// The implicit GEMM of conv_igemm.hip gathers every input pixel once per tap, i.e. it moves 9x the input through L2 -> CU; with
// only 32 output channels there is too little MFMA work per byte to hide that (measured: 43 us per layer at bs=128 against a 15 us
// HBM bound).  Here a workgroup loads the (8+2) x (16+2) pixel halo tile of its 8x16 output tile ONCE into LDS and forms the nine
// taps from there; the 3x3x32x32 weights live in registers (18 MFMA B fragments per lane).  Same reduction order as the implicit
// GEMM (tap-major, channel-minor, one v_mfma_f32_32x32x16_bf16 chain per output tile) -> bit-identical results, which is how it
// is tested.  Replaces nets/pose_hrnet.py BasicBlock convs (conv3x3 + BN [+ residual] + ReLU) on 32-channel branches.
#include "sp_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16;                 // output tile (pixels); one wave = two tile rows = 32 pixels
constexpr int HW_ = TW + 2, HH_ = TH + 2;      // halo tile
constexpr int CI = 32, CO = 32;
constexpr int PIX_BYTES = CI * 2;              // 64 B per pixel = 4 chunks of 16 B
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

// 16-byte chunk c (0..3) of halo pixel P sits at P*64 + ((c ^ ((P >> 1) & 3)) * 16): eight consecutive pixels reading the same chunk
// (one ds_read_b128 beat) then cover all eight 16-byte slots of a 128-byte bank row
__device__ __forceinline__ int xoff(int P, int c) { return P * PIX_BYTES + ((c ^ ((P >> 1) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void conv3x3_c32_direct_kernel(const DirectArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][HH_ * HW_ * PIX_BYTES];   // 2 x 11,520 B: halo tile, double-buffered
    __shared__ __attribute__((aligned(16))) float Tr[4][32 * 32];                         // epilogue transpose, 4 KB per wave (private)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.x_bytes, 0x00020000);

    // this thread's three halo chunks: pixel P (of 180), 16-byte chunk c - fixed for the whole launch
    int hP[3], hc[3], hy[3], hx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = tid + 256 * i;
        hP[i] = q >> 2; hc[i] = q & 3;
        hy[i] = hP[i] / HW_; hx[i] = hP[i] - hy[i] * HW_;
    }
    u32x4 hv[3], rvn[2];
    unsigned noff[2];
    // everything tile `tile` needs from memory -> registers: its halo pixels and the residual behind this lane's two output chunks
    // (out-of-image / past-the-end: zeros through an out-of-range offset)
    auto request = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = ty * TH - 1 + hy[i], ix = tx * TW - 1 + hx[i];
            const bool ok = tile < ntiles && tid + 256 * i < HH_ * HW_ * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * CI + hc[i] * 8) * 2) : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + (lane >> 2), chunk = lane & 3;      // 16 pixels x 4 chunks of 8 channels per pass
            const int oy = ty * TH + 2 * wave + (row >> 4), ox = tx * TW + (row & 15);
            noff[it] = (tile < ntiles && oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * CO + chunk * 8) * 2) : OOB;
            rvn[it] = u32x4{0u, 0u, 0u, 0u};
            if (p.res) rvn[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, noff[it], 0, 0);
        }
    };
    request(blockIdx.x);
    // ---- weights -> registers, once per (persistent) workgroup: fragment f = tap*2 + ks: W[n = lane % 32][f*16 + (lane / 32) * 8 .. + 8]
    u32x4 wf[18];
#pragma unroll
    for (int f = 0; f < 18; ++f)
        wf[f] = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)((fr * p.k_pad + f * 16 + fh * 8) * 2), 0, 0);
    float sc[8], sh[8];                                       // this lane's 8 output channels in the epilogue
    {
        const int chunk = lane & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = p.scale ? p.scale[chunk * 8 + e] : 1.f; sh[e] = p.shift ? p.shift[chunk * 8 + e] : 0.f; }
    }

    const int py = 2 * wave + (fr >> 4), px = fr & 15;       // output pixel of this lane's A row inside the tile
    float* tr = Tr[wave];
    int cur = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        unsigned char* X = Xs[cur];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (tid + 256 * i < HH_ * HW_ * 4) *reinterpret_cast<u32x4*>(X + xoff(hP[i], hc[i])) = hv[i];
        const unsigned ooff[2] = {noff[0], noff[1]};          // this tile's output offsets and residual (arrived with its halo)
        const u32x4 rv[2] = {rvn[0], rvn[1]};
        __syncthreads();                                      // tile `tile` is in LDS; buffer cur^1 is no longer being read
        request(tile + gridDim.x);                            // the next tile's operands fly during this tile's MFMAs and stores

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int P = (py + tap / 3) * HW_ + px + tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(X + xoff(P, ks * 2 + fh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wf[tap * 2 + ks]), acc, 0, 0, 0);
            }
        }
        // ---- epilogue: C/D map col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the wave's 32);
        //      through the wave's private LDS slice so that a lane owns 8 consecutive channels of one pixel (16-byte stores)
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[r];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 16 + (lane >> 2), chunk = lane & 3;
            float v[8];
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) {
                const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4 * e4);
                v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (p.res) {
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[it]);
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
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff[it], 0, 0);
        }
    }
}

// ---- 64 -> 64 channels --------------------------------------------------------------------------------------------------------------
// (HRNet's second branch: 64 of its 293 convs at 32x24; ResNet-50 layer1.*.conv2 at 64x48.)  Same idea, different budget: the filter is
// 72 KiB, too large for registers, so it lives in LDS for the whole (persistent, one per CU) workgroup in fragment order - the B
// operand of every MFMA is ONE conflict-free ds_read_b128 - next to a double-buffered halo tile of the 16x8-pixel output tile
// (18 x 10 pixels x 128 B).  A wave owns 32 pixels x 64 channels: per (tap, 16-channel K step) one A fragment feeds two MFMAs.  The
// implicit GEMM spends 26 us on such a layer at bs=128 (9 short K tiles per workgroup, every input pixel gathered nine times) against
// ~8 us of HBM time and ~7 us of MFMA time.
constexpr int T6H = 16, T6W = 8;                   // output tile: 16 rows x 8 columns; wave w = rows 4w..4w+3
constexpr int H6W = T6W + 2, H6H = T6H + 2;        // halo tile 18 x 10 = 180 pixels
constexpr int C6 = 64, PIX6 = C6 * 2;              // 128 B per pixel = 8 chunks of 16 B
constexpr int W6_BYTES = 9 * 4 * 2 * 2 * 32 * 16;  // [tap][k step][column block][k half][column] 16-B fragments = 73,728 B
constexpr int X6_BYTES = H6H * H6W * PIX6;         // 23,040 B per halo buffer
constexpr int TR6_BYTES = 32 * 64 * 4;             // epilogue transpose, per wave
constexpr int LDS6 = W6_BYTES + 2 * X6_BYTES + 4 * TR6_BYTES;

// chunk c (0..7) of halo pixel (hy, hx) -> byte offset.  Sixteen lanes of one ds_read_b128 beat hold four rows x four consecutive
// columns reading the same chunk: (hx & 1) picks the half of a 256-byte bank row (180 pixels x 128 B: pixel parity = column parity),
// ((hx >> 1) & 1) | ((hy & 3) << 1) spreads the rest over its eight 16-byte slots.
__device__ __forceinline__ int x6off(int hy, int hx, int c) {
    return (hy * H6W + hx) * PIX6 + ((c ^ (((hx >> 1) & 1) | ((hy & 3) << 1))) << 4);
}

__global__ __launch_bounds__(256, 1) void conv3x3_c64_direct_kernel(const DirectArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    unsigned char* const Ws = smem6;
    unsigned char* const Xs0 = smem6 + W6_BYTES;
    float* const Tr = reinterpret_cast<float*>(smem6 + W6_BYTES + 2 * X6_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.x_bytes, 0x00020000);

    // this thread's six halo chunks (180 pixels x 8 chunks = 1,440 = 5.6 per thread): pixel q >> 3, chunk q & 7 - fixed for the launch
    constexpr int NH = (H6H * H6W * 8 + 255) / 256;
    int hy[NH], hx[NH], hc[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int q = tid + 256 * i, P = q >> 3;
        hc[i] = q & 7;
        hy[i] = P / H6W; hx[i] = P - hy[i] * H6W;
    }
    // TWO tiles' operands in flight in registers (one wave per SIMD: nothing else hides a loaded-HBM round trip of several us, and
    // one tile of MFMAs is only ~1.2 us): set S holds tile i, i+2, ... of this workgroup
    u32x4 hv[2][NH], rvn[2][4];
    unsigned noff[2][4];
    auto request = [&](auto slot, int tile) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int iy = ty * T6H - 1 + hy[i], ix = tx * T6W - 1 + hx[i];
            const bool ok = tile < ntiles && tid + 256 * i < H6H * H6W * 8 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * C6 + hc[i] * 8) * 2) : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), chunk = lane & 7;       // 8 pixels x 8 chunks of 8 channels per pass
            const int oy = ty * T6H + 4 * wave + (row >> 3), ox = tx * T6W + (row & 7);
            noff[S][it] = (tile < ntiles && oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C6 + chunk * 8) * 2) : OOB;
            rvn[S][it] = u32x4{0u, 0u, 0u, 0u};
            if (p.res) rvn[S][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, noff[S][it], 0, 0);
        }
    };
    const int G = gridDim.x;
    request(std::integral_constant<int, 0>{}, blockIdx.x);
    request(std::integral_constant<int, 1>{}, blockIdx.x + G);
    // ---- filter -> LDS in fragment order, once per workgroup: fragment (f = tap*4 + ks, nb, kh, n) = W[nb*32 + n][f*16 + kh*8 .. +8] ----
    for (int q = tid; q < W6_BYTES / 16; q += 256) {
        const int n = q & 31, kh = (q >> 5) & 1, nb = (q >> 6) & 1, f = q >> 7;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)(((nb * 32 + n) * p.k_pad + f * 16 + kh * 8) * 2), 0, 0);
        *reinterpret_cast<u32x4*>(Ws + q * 16) = v;
    }
    float sc[8], sh[8];                                       // this lane's 8 output channels in the epilogue
    {
        const int chunk = lane & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = p.scale ? p.scale[chunk * 8 + e] : 1.f; sh[e] = p.shift ? p.shift[chunk * 8 + e] : 0.f; }
    }
    const int py = 4 * wave + (fr >> 3), px = fr & 7;        // output pixel of this lane's A row inside the tile
    const unsigned char* const wfrag = Ws + (fh * 32 + fr) * 16;
    float* tr = Tr + wave * (32 * 64);
    auto body = [&](auto slot, int tile) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        unsigned char* X = Xs0 + S * X6_BYTES;                 // tiles alternate between the two halo buffers as between the register sets
#pragma unroll
        for (int i = 0; i < NH; ++i)
            if (tid + 256 * i < H6H * H6W * 8) *reinterpret_cast<u32x4*>(X + x6off(hy[i], hx[i], hc[i])) = hv[S][i];
        const unsigned ooff[4] = {noff[S][0], noff[S][1], noff[S][2], noff[S][3]};
        const u32x4 rv[4] = {rvn[S][0], rvn[S][1], rvn[S][2], rvn[S][3]};
        __syncthreads();                                      // tile `tile` (and, first time round, the filter) is in LDS; the other buffer is free
        request(slot, tile + 2 * G);                          // the tile after next: its operands fly during two tiles of MFMAs and stores

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // 36 steps (tap, 16-channel K step), each one A fragment + two B fragments -> two MFMAs.  One wave per SIMD issues in order, so the
        // fragments of step s+PF are requested before the MFMAs of step s (a ring of PF+1 register sets; without it every step waits a
        // full LDS round trip: measured 3.5x the MFMA time).
        constexpr int PF = 3, NSTEP = 36;
        u32x4 fa[PF + 1], fb0[PF + 1], fb1[PF + 1];
        auto frags = [&](int st) __attribute__((always_inline)) {
            const int tap = st >> 2, ks = st & 3;
            const int ay = py + tap / 3, ax = px + tap % 3;
            fa[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(X + x6off(ay, ax, ks * 2 + fh));
            const unsigned char* wb = wfrag + st * 2048;
            fb0[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(wb);
            fb1[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(wb + 1024);
        };
#pragma unroll
        for (int st = 0; st < PF; ++st) frags(st);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            if (st + PF < NSTEP) frags(st + PF);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb0[st % (PF + 1)]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb1[st % (PF + 1)]), acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: through the wave's private LDS slice so that a lane owns 8 consecutive channels of one pixel (16-byte stores) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            tr[row * 64 + fr] = acc0[r];
            tr[row * 64 + 32 + fr] = acc1[r];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), chunk = lane & 7;
            float v[8];
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) {
                const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 64 + chunk * 8 + 4 * e4);
                v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (p.res) {
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[it]);
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
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff[it], 0, 0);
        }
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += 2 * G) {
        body(std::integral_constant<int, 0>{}, tile);
        if (tile + G < ntiles) body(std::integral_constant<int, 1>{}, tile + G);
    }
}

}  // namespace

// Eligibility: what the engine checks before it offers this kernel to the tuner (sp_conv3x3_direct_ok) and what the launch re-checks.
static bool direct_ok(const sp_conv_desc* d) {
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
    SP_REQUIRE(direct_ok(d), "sp_conv3x3_direct: needs a bf16 3x3 stride-1 pad-1 convolution with 32 -> 32 or 64 -> 64 channels (NHWC bf16 out)");
    SP_REQUIRE(d->batch > 0, "sp_conv3x3_direct: bad batch");
    const long long elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    SP_REQUIRE(elems < (1ll << 29), "sp_conv3x3_direct: tensor too large");
    DirectArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.k_pad = d->k_pad; a.batch = d->batch;
    a.relu = (d->flags & SP_CONV_RELU) ? 1 : 0;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    if (d->c_in == 64) {
        if (sp_name_query_active()) { sp_name_query_set("conv3x3_c64_direct_kernel"); return SP_OK; }
        a.tiles_x = (d->in_w + T6W - 1) / T6W; a.tiles_y = (d->in_h + T6H - 1) / T6H;
        const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
        SP_REQUIRE(tiles < (1ll << 31), "sp_conv3x3_direct: too many tiles");
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        // one persistent workgroup per CU (the filter takes 72 KiB of its LDS), walking its tiles with the next halo in flight
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS6);
        if (e != hipSuccess) { sp_set_error("sp_conv3x3_direct: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDS6, hipGetErrorString(e)); return SP_ELAUNCH; }
        const long long grid = tiles < cus ? tiles : cus;
        hipLaunchKernelGGL(conv3x3_c64_direct_kernel, dim3((unsigned)grid), dim3(256), LDS6, (hipStream_t)stream, a);
        return sp_check_launch("conv3x3_c64_direct_kernel");
    }
    a.tiles_x = (d->in_w + TW - 1) / TW; a.tiles_y = (d->in_h + TH - 1) / TH;
    const long long blocks = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(blocks < (1ll << 31), "sp_conv3x3_direct: too many tiles");
    // persistent workgroups: the weights (18 KB per wave) are fetched once per workgroup, the next tile's operands are requested
    // before the current tile's MFMAs; 2 workgroups per CU are resident (~180 VGPRs: 72 of them hold the weights)
    const long long grid = blocks < 256 * 2 ? blocks : 256 * 2;
    hipLaunchKernelGGL(conv3x3_c32_direct_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return sp_check_launch("conv3x3_c32_direct_kernel");
}
