// conv_block.hip - one HRNet BasicBlock (nets/pose_hrnet.py:34-51) of the 32-channel branch as ONE launch, bf16:
//
//     out = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x)
//
// The two convolutions of the block are HBM-bound when run one by one (conv_direct.hip: 75 MB per launch at bs=128 for 7 GFLOP): the
// intermediate t is written and read back, and the block input is read twice more (as conv1's operand and as conv2's residual).
// Here a workgroup loads the (8+4) x (16+4) pixel halo of its 8x16 output tile ONCE, computes t on the (8+2) x (16+2) pixels conv2
// needs (1.4x conv1's MFMAs - cheap: this path has 4x more MFMA time than it uses), keeps t in LDS as bf16 (what the two-launch
// path stores), and takes the residual from the halo tile it already holds: 25 MB in + 25 MB out per block instead of 125 MB.
// Same MFMA chains (tap-major, channel-minor v_mfma_f32_32x32x16_bf16), same bf16 rounding of t, same epilogue arithmetic as the two
// launches -> bit-identical results, which is how it is tested.
#include "sp_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 16;                  // output tile
constexpr int TTH = TH + 2, TTW = TW + 2;       // t tile: 10 x 18 = 180 pixels (6 MFMA row blocks of 32, 12 slots idle)
constexpr int XH = TH + 4, XW = TW + 4;         // x halo: 12 x 20 = 240 pixels
constexpr int C = 32, PIXB = C * 2;             // 64 B per pixel = 4 chunks of 16 B
constexpr int NT = TTH * TTW;
constexpr unsigned OOB = 0x80000000u;
constexpr int XS_BYTES = XH * XW * PIXB;        // 15,360 B per halo buffer
constexpr int TS_BYTES = 192 * PIXB;            // 12,288 B
constexpr int W2_BYTES = 18 * 1024;             // conv2's 18 B fragments, one KiB each (lane-linear)
constexpr int TR_BYTES = 32 * 32 * 4;           // per-wave transpose scratch

struct BlockArgs {
    const void* x;
    const void* w1; const float* scale1; const float* shift1;
    const void* w2; const float* scale2; const float* shift2;
    void* y;
    int H, W, k_pad, batch, tiles_x, tiles_y;
    int x_bytes, w_bytes;
};

// 16-byte chunk c (0..3) of pixel P (row-major index inside its tile) -> byte offset (conv_direct.hip's image: eight consecutive
// pixels reading the same chunk cover the eight 16-byte slots of two 128-byte rows)
__device__ __forceinline__ int poff(int P, int c) { return P * PIXB + ((c ^ ((P >> 1) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void basic_block_c32_kernel(const BlockArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char Xs[2][XS_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Ts[TS_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char W2s[W2_BYTES];
    __shared__ __attribute__((aligned(16))) float Tr[4][32 * 32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);

    // this thread's four halo chunks (240 pixels x 4 chunks = 960 = 3.75 per thread): pixel q >> 2, chunk q & 3
    int hP[4], hc[4], hy[4], hx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        hP[i] = q >> 2; hc[i] = q & 3;
        hy[i] = hP[i] / XW; hx[i] = hP[i] - hy[i] * XW;
    }
    // THREE tiles' halos in flight in registers: this kernel moves so few bytes per tile (15 KB in, 8 KB out) that with one tile in
    // flight per workgroup the chip holds < 8 MB in flight - Little's law then caps it near 2.9 TB/s whatever the kernel does
    constexpr int NSET = 3;
    u32x4 hv[NSET][4];
    auto request = [&](auto slot, int tile) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = ty * TH - 2 + hy[i], ix = tx * TW - 2 + hx[i];
            const bool ok = tile < ntiles && tid + 256 * i < XH * XW * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hv[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * C + hc[i] * 8) * 2) : OOB, 0, 0);
        }
    };
    const int G = gridDim.x;
    request(std::integral_constant<int, 0>{}, blockIdx.x);
    request(std::integral_constant<int, 1>{}, blockIdx.x + G);
    request(std::integral_constant<int, 2>{}, blockIdx.x + 2 * G);
    // conv1's filter -> registers (fragment f = tap*2 + ks: W1[n = lane % 32][f*16 + (lane / 32) * 8 .. + 8]); conv2's -> LDS, lane-linear
    u32x4 wf[18];
#pragma unroll
    for (int f = 0; f < 18; ++f)
        wf[f] = __builtin_amdgcn_raw_buffer_load_b128(w1r, (unsigned)((fr * p.k_pad + f * 16 + fh * 8) * 2), 0, 0);
    for (int q = tid; q < 18 * 64; q += 256) {
        const int f = q >> 6, l = q & 63;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(((l & 31) * p.k_pad + f * 16 + (l >> 5) * 8) * 2), 0, 0);
        *reinterpret_cast<u32x4*>(W2s + q * 16) = v;
    }
    float sc1[8], sh1[8], sc2[8], sh2[8];                    // this lane's 8 channels in the epilogues
    {
        const int chunk = lane & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc1[e] = p.scale1 ? p.scale1[chunk * 8 + e] : 1.f; sh1[e] = p.shift1 ? p.shift1[chunk * 8 + e] : 0.f;
            sc2[e] = p.scale2 ? p.scale2[chunk * 8 + e] : 1.f; sh2[e] = p.shift2 ? p.shift2[chunk * 8 + e] : 0.f;
        }
    }
    float* tr = Tr[wave];
    const unsigned char* const w2frag = W2s + lane * 16;
    int cur = 0;
    auto body = [&](auto slot, int tile) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        unsigned char* X = Xs[cur];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + 256 * i < XH * XW * 4) *reinterpret_cast<u32x4*>(X + poff(hP[i], hc[i])) = hv[S][i];
        int t_ = tile;
        const int tlx = t_ % p.tiles_x; t_ /= p.tiles_x;
        const int tly = t_ % p.tiles_y;
        const int b = t_ / p.tiles_y;
        __syncthreads();            // the halo (and, first time round, conv2's filter) is in LDS; every wave is done with the previous tile's Ts
        request(slot, tile + NSET * G);   // refill the register set just emptied: three tiles ahead

        // ---- conv1 on the t tile: blocks wave, wave + 4 (six blocks of 32 t pixels; slots 180..191 of the last one are idle) ----
        for (int blk = wave; blk < 6; blk += 4) {
            const int q = blk * 32 + fr;
            const int qq = q < NT ? q : NT - 1;             // idle slots compute a duplicate and are not stored
            const int ty = qq / TTW, tx = qq - ty * TTW;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // 18 steps (tap, 16-channel K step); the A fragment of step s+PF is requested before the MFMA of step s (in-order issue: the
            // wave issues in order)
            constexpr int PF = 2;
            u32x4 fa[PF + 1];
            auto afrag = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 1, ks = st & 1;
                fa[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(X + poff((ty + tap / 3) * XW + tx + tap % 3, ks * 2 + fh));
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) afrag(st);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PF < 18) afrag(st + PF);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st % (PF + 1)]), __builtin_bit_cast(bf16x8, wf[st]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue 1: t = relu(acc * scale1 + shift1) as bf16 into Ts - ZERO outside the image (conv2 pads t, not x)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[r];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2), chunk = lane & 3;
                const int tq = blk * 32 + row;
                const int py = tq / TTW, px = tq - py * TTW;
                const int iy = tly * TH - 1 + py, ix = tlx * TW - 1 + px;
                const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                float v[8];
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4 * e4);
                    v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                }
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z = v[e] * sc1[e] + sh1[e];
                    z = z > 0.f ? z : 0.f;
                    o8[e] = (__bf16)(inside ? z : 0.f);
                }
                if (tq < NT) *reinterpret_cast<u32x4*>(Ts + poff(tq, chunk)) = __builtin_bit_cast(u32x4, o8);
            }
        }
        __syncthreads();            // t is complete

        // ---- conv2 on the output tile: wave w = output rows 2w, 2w+1 (32 pixels) ----
        {
            const int py = 2 * wave + (fr >> 4), px = fr & 15;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = 2;
            u32x4 fa[PF + 1], fb[PF + 1];
            auto frags = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 1, ks = st & 1;
                fa[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(Ts + poff((py + tap / 3) * TTW + px + tap % 3, ks * 2 + fh));
                fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(w2frag + st * 1024);
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) frags(st);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PF < 18) frags(st + PF);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // epilogue 2: out = relu(acc * scale2 + shift2 + x), the residual from the halo tile (pixel (oy+2, ox+2) of it)
#pragma unroll
            for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[r];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + (lane >> 2), chunk = lane & 3;
                const int oyl = 2 * wave + (row >> 4), oxl = row & 15;
                const int oy = tly * TH + oyl, ox = tlx * TW + oxl;
                float v[8];
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + row * 32 + chunk * 8 + 4 * e4);
                    v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                }
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(X + poff((oyl + 2) * XW + oxl + 2, chunk)));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = v[e] * sc2[e] + sh2[e];
                    v[e] += (float)r8[e];
                    v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
                const unsigned off = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C + chunk * 8) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, off, 0, 0);
            }
        }
        cur ^= 1;
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += NSET * G) {
        body(std::integral_constant<int, 0>{}, tile);
        if (tile + G < ntiles) body(std::integral_constant<int, 1>{}, tile + G);
        if (tile + 2 * G < ntiles) body(std::integral_constant<int, 2>{}, tile + 2 * G);
    }
}

bool block_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;
    return d && (d->flags & SP_CONV_BF16) && !(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) && d->c_in == 32 &&
           d->c_out == 32 && d->out_c == 32 && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) &&
           d->dy0 == -1 && d->dx0 == -1 && d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == 320 &&
           d->n_pad >= 32 && d->grid_h == d->in_h && d->grid_w == d->in_w && d->out_h == d->in_h && d->out_w == d->in_w && d->oy_mul == 1 &&
           d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0;
}

}  // namespace

extern "C" int sp_basic_block_c32_ok(const sp_conv_desc* d) { return block_ok(d) ? 1 : 0; }

extern "C" int sp_basic_block_c32(const sp_conv_desc* d, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                                  const void* w2_packed, const float* scale2, const float* shift2, void* y, void* stream) {
    SP_REQUIRE(d && x && w1_packed && w2_packed && y, "sp_basic_block_c32: null pointer");
    SP_REQUIRE(block_ok(d), "sp_basic_block_c32: `desc` must describe the block's bf16 3x3 stride-1 pad-1 convolutions with 32 -> 32 channels");
    SP_REQUIRE(x != y, "sp_basic_block_c32: the output must not alias the input (neighbouring tiles read the input's halo)");
    SP_REQUIRE(d->batch > 0, "sp_basic_block_c32: bad batch");
    if (sp_name_query_active()) { sp_name_query_set("basic_block_c32_kernel"); return SP_OK; }
    const long long elems = (long long)d->batch * d->in_h * d->in_w * 32;
    SP_REQUIRE(elems < (1ll << 29), "sp_basic_block_c32: tensor too large");
    BlockArgs a;
    a.x = x; a.w1 = w1_packed; a.scale1 = scale1; a.shift1 = shift1; a.w2 = w2_packed; a.scale2 = scale2; a.shift2 = shift2; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.k_pad = d->k_pad; a.batch = d->batch;
    a.tiles_x = (d->in_w + TW - 1) / TW; a.tiles_y = (d->in_h + TH - 1) / TH;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_basic_block_c32: too many tiles");
    const long long grid = tiles < 256 * 2 ? tiles : 256 * 2;     // persistent: two workgroups per CU, filters fetched once each
    hipLaunchKernelGGL(basic_block_c32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return sp_check_launch("basic_block_c32_kernel");
}
