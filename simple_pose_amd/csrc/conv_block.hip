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
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

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

// ---- round 6: the same block on 8 x 48-pixel strips, eight waves ------------------------------------------------------------------------
// The four-wave kernel above ties with the two launches it replaces (40 against 2 x 18-20 us at bs=128): ten vector instructions per MFMA
// (swizzled fragment addresses, fp32 transposes through LDS for both epilogues) and a chain of dependent steps per small tile.  This form
// removes the vector work instead of hiding it:
//   * operands SWAPPED: the filter fragment is the MFMA's A operand, the pixels' fragment its B operand, so the accumulator holds, per lane,
//     ONE pixel x 16 channels (four runs of four consecutive channels) - t goes to LDS and y to memory in NHWC order without a transpose
//     (the products and the order of the K steps are those of the unswapped chain: same bits, which is how it is tested);
//   * LDS images in four 16-byte-chunk PLANES ([chunk][pixel][16 B]): a fragment read of 32 consecutive pixels is conflict-free without
//     a swizzle, so every tap is the row block's base address + an immediate offset (no vector instruction per read);
//   * both filters live in registers (144 VGPRs), the BatchNorm scale / shift come from a 512-byte LDS table in accumulator order;
//   * tiles are 8 x 48 pixels (HRNet's 64 x 48 maps: full-width strips, 16 + 12 row blocks of MFMA work per tile against 12 + 12 without
//     the halo), a workgroup walks CONSECUTIVE strips (the four halo rows two strips share are L2-hot), the next strip's halo is in
//     flight in registers during the whole of this strip's arithmetic and lands in the other of two LDS buffers: two barriers per strip.
#ifdef SP_BB32_DIAG
// DIAGNOSTIC BUILD ONLY (tools/diag_bb32.py --stamps; never the shipped library): per-wave cycle sums of the phases, read back with
// sp_bb32_debug_read().  [block % 256][wave][10]: 0 prologue, 1 conv1 MFMA loops, 2 conv1 epilogues, 3 conv2 MFMA loops, 4 conv2 epilogues + stores,
// 5 barrier after conv1, 6 wait + barrier at the end of the strip, 7 halo requests, 8 kernel lifetime (s_memtime), 9 kernel lifetime in s_memrealtime
// ticks (100 MHz)
__device__ unsigned long long sp_bb32_dbg[256 * 8 * 10];
extern "C" int sp_bb32_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sp_bb32_dbg), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#define SP_KSTAMP(var)                                                                      \
    unsigned long long var;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#define SP_KACC(slot, a, b) dg[slot] += (b) - (a);
#else
#define SP_KSTAMP(var)
#define SP_KACC(slot, a, b)
#endif
#ifndef BB_PF
#define BB_PF 3
#endif
namespace s48 {
constexpr int TH = 8, TW = 48;
constexpr int TTW = TW + 2, NT = (TH + 2) * TTW;           // t tile: 10 x 50 = 500 pixels = 16 row blocks of 32 (12 slots idle)
constexpr int XH = TH + 4, XW = TW + 4, NX = XH * XW;     // x halo: 12 x 52 = 624 pixels (planes padded to 640: ten 64-pixel LDS-DMA pieces)
constexpr int XPLANE = 640 * 16, TPLANE = 512 * 16;
constexpr int X_BYTES = 4 * XPLANE, T_BYTES = 4 * TPLANE;
constexpr int PPW = 5;                                     // LDS-DMA pieces per wave and strip (4 planes x 10 pieces over 8 waves)
constexpr int WSTAGE = 36 * 512;                           // one filter in fragment order: [16-byte column][row][16 B]
constexpr int LDS_BYTES = T_BYTES + 512 + 2 * X_BYTES;
static_assert(LDS_BYTES <= 160 * 1024 && 2 * WSTAGE <= X_BYTES, "LDS");

// one LDS-DMA piece (conv_ring.hip dma16: inline asm on purpose, see there): 64 lanes x 16 bytes, lane l's bytes from rsrc + voff (zeros when out
// of range) to LDS at lds_addr + 16 l
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc)
                 : "memory");
}

__global__ __launch_bounds__(512) void basic_block_c32_w8_kernel(const BlockArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: t tile [0, 32 KB) and the scale / shift table first (their reads are base + 16-bit immediate), then the two halo buffers
    unsigned char* const Ts = smem;
    float* const tab = reinterpret_cast<float*>(smem + T_BYTES);
    unsigned char* const Xs = smem + T_BYTES + 512;
    const unsigned xs_lds = (unsigned)(size_t)(lds_void_t*)Xs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = blockIdx.x * per;
    const int tile_end = tile0 + per < ntiles ? tile0 + per : ntiles;
    if (tile0 >= ntiles) return;                           // (the whole workgroup)
#ifdef SP_BB32_DIAG
    unsigned long long dg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif
    SP_KSTAMP(k0)

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);

    // a strip's halo arrives by LDS-DMA: piece id = 5 wave + i covers pixels (id % 10) * 64 + lane of chunk plane id / 10
    int hyx[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int P = ((wave * PPW + i) % 10) * 64 + lane;
        const int hy = P / XW, hx = P - hy * XW;
        hyx[i] = P < NX ? (hy << 8) | hx : (1 << 30);       // (row 2^22: never inside - the planes' padding receives zeros)
    }
    unsigned hsrc[PPW];                                     // this lane's source offsets of the strip being requested
    auto addresses = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int y0 = ty * TH - 2, x0 = tx * TW - 2;
        const int base = ((b * p.H + y0) * p.W + x0) * (C * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int id = wave * PPW + i;
            const int hy = hyx[i] >> 8, hx = hyx[i] & 255;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = tile < tile_end && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            hsrc[i] = ok ? (unsigned)(base + (hy * p.W + hx) * (C * 2) + (id / 10) * 16) : OOB;
        }
    };
    auto piece = [&](int i, int buf) __attribute__((always_inline)) {
        const int id = wave * PPW + i;
        dma16(xs_lds + (unsigned)(buf * X_BYTES + (id / 10) * XPLANE + (id % 10) * 1024), hsrc[i], xr);
    };
    addresses(tile0);
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(i, 0);
    // both filters -> LDS (coalesced: the 32 rows of 640 B are 1,280 consecutive 16-byte pieces; every CU reading the fragments straight from L2 - 32
    // rows per instruction, the same 36 KB for all 2,048 waves - took 10 us) in fragment order, then -> registers: fragment f = tap*2 + ks is
    // W[n = lane % 32][f*16 + (lane / 32) * 8 .. + 8], the MFMA's A operand here
    {
        unsigned char* const stage = Xs + X_BYTES;          // (the second halo buffer is free until the first strip's request for the next)
        u32x4 wv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = tid + 512 * i, pp = q < 1280 ? q : q - 1280;
            wv[i] = __builtin_amdgcn_raw_buffer_load_b128(q < 1280 ? w1r : w2r, (unsigned)(pp * 16), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int q = tid + 512 * i, pp = q < 1280 ? q : q - 1280;
            const int r = pp / 40, cidx = pp - r * 40;
            if (cidx < 36) *reinterpret_cast<u32x4*>(stage + (q < 1280 ? 0 : WSTAGE) + (cidx * 32 + r) * 16) = wv[i];
        }
        if (tid < 128) {
            const int k = tid >> 5, c = tid & 31;
            const float* src = k == 0 ? p.scale1 : (k == 1 ? p.shift1 : (k == 2 ? p.scale2 : p.shift2));
            tab[tid] = src ? src[c] : ((k & 1) ? 0.f : 1.f);
        }
    }
    __syncthreads();
    u32x4 wf1[18], wf2[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        wf1[f] = *reinterpret_cast<const u32x4*>(Xs + X_BYTES + ((2 * f + fh) * 32 + fr) * 16);
        wf2[f] = *reinterpret_cast<const u32x4*>(Xs + X_BYTES + WSTAGE + ((2 * f + fh) * 32 + fr) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the first strip's halo
    __syncthreads();                                        // the first halo is complete; every wave has its fragments (the staging area is free)
    SP_KSTAMP(k1)
    SP_KACC(0, k0, k1)

    int cur = 0;
    for (int tile = tile0; tile < tile_end; ++tile) {
        unsigned char* const X = Xs + cur * X_BYTES;
        int t_ = tile;
        const int tlx = t_ % p.tiles_x; t_ /= p.tiles_x;
        const int tly = t_ % p.tiles_y;
        const int b = t_ / p.tiles_y;
        SP_KSTAMP(ka0)
        // the next strip's halo -> the other buffer (free since the barrier that ended the last strip): addresses now, the five LDS-DMA pieces between
        // the MFMAs of this wave's first conv1 block (issued in one burst by all eight waves they took 1,000-1,250 cycles per strip)
        addresses(tile + 1);
        SP_KSTAMP(ka)
#if BB_STAGGER
        if (wave >= 4) __builtin_amdgcn_s_sleep(BB_STAGGER);
#endif
        // ---- conv1 on the t tile: row blocks wave, wave + 8 ----
#pragma unroll 1
        for (int blk = wave; blk < 16; blk += 8) {
            const int q = blk * 32 + fr;
            const int qq = q < NT ? q : NT - 1;             // idle slots compute a duplicate and are not stored
            const int ty = qq / TTW, tx = qq - ty * TTW;
            const unsigned char* const xb = X + fh * XPLANE + (ty * XW + tx) * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = BB_PF;
            u32x4 fb[PF + 1];
            SP_KSTAMP(m0)
            auto frag = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 1, ks = st & 1;
                fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(xb + ks * 2 * XPLANE + ((tap / 3) * XW + tap % 3) * 16);
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) frag(st);
            f32x4 scA[2], shA[2];                   // scale / shift of channel groups 0, 1 (requested before the last MFMAs; groups 2, 3 follow after the loop)
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PF < 18) frag(st + PF);
                if (st == 14) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        scA[g] = *reinterpret_cast<const f32x4*>(tab + 8 * g + 4 * fh);
                        shA[g] = *reinterpret_cast<const f32x4*>(tab + 32 + 8 * g + 4 * fh);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf1[st]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (blk < 8 && st % 3 == 1 && st / 3 < PPW) piece(st / 3, cur ^ 1);
            }
            SP_KSTAMP(m1)
            SP_KACC(1, m0, m1)
            // epilogue 1: t = relu(acc * scale1 + shift1) as bf16 into Ts - ZERO outside the image (conv2 pads t, not x).
            // acc[4g + j] = channel 8g + 4 fh + j of this lane's pixel
            f32x4 scB[2], shB[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                scB[g] = *reinterpret_cast<const f32x4*>(tab + 8 * (g + 2) + 4 * fh);
                shB[g] = *reinterpret_cast<const f32x4*>(tab + 32 + 8 * (g + 2) + 4 * fh);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int iy = tly * TH - 1 + ty, ix = tlx * TW - 1 + tx;
            const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = g < 2 ? scA[g & 1] : scB[g & 1];
                const f32x4 sh = g < 2 ? shA[g & 1] : shB[g & 1];
                f32x2 z01, z23;                       // (max(z, +0) == z > 0 ? z : 0 for every z, -0 and NaN included: v_max_f32 orders -0 below +0 and drops a NaN)
                z01[0] = __builtin_fmaxf(acc[4 * g + 0] * sc[0] + sh[0], 0.f); z01[1] = __builtin_fmaxf(acc[4 * g + 1] * sc[1] + sh[1], 0.f);
                z23[0] = __builtin_fmaxf(acc[4 * g + 2] * sc[2] + sh[2], 0.f); z23[1] = __builtin_fmaxf(acc[4 * g + 3] * sc[3] + sh[3], 0.f);
                u32x2 o;
                o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(z01, bf16x2));
                o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(z23, bf16x2));
                o[0] = inside ? o[0] : 0u; o[1] = inside ? o[1] : 0u;
                if (q < NT) *reinterpret_cast<u32x2*>(Ts + g * TPLANE + q * 16 + fh * 8) = o;
            }
            SP_KSTAMP(m2)
            SP_KACC(2, m1, m2)
        }
        SP_KSTAMP(kb)
        __syncthreads();            // t is complete
        SP_KSTAMP(kc)
#if BB_STAGGER
        if (wave >= 4) __builtin_amdgcn_s_sleep(BB_STAGGER);
#endif

        // ---- conv2 on the output tile: twelve row blocks; waves 0 .. 3 take two (w, w + 8), waves 4 .. 7 one: three per SIMD ----
#pragma unroll 1
        for (int ob = wave; ob < 12; ob += 8) {
            const int o = ob * 32 + fr;
            const int oyl = o / TW, oxl = o - oyl * TW;
            const unsigned char* const tb = Ts + fh * TPLANE + (oyl * TTW + oxl) * 16;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = BB_PF;
            u32x4 fb[PF + 1];
            SP_KSTAMP(n0)
            auto frag = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 1, ks = st & 1;
                fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(tb + ks * 2 * TPLANE + ((tap / 3) * TTW + tap % 3) * 16);
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) frag(st);
            const unsigned char* const rb = X + ((oyl + 2) * XW + oxl + 2) * 16 + fh * 8;     // the residual: pixel (oy + 2, ox + 2) of the halo tile
            f32x4 scA[2], shA[2];
            u32x2 rA[2];
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + PF < 18) frag(st + PF);
                if (st == 14) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        scA[g] = *reinterpret_cast<const f32x4*>(tab + 64 + 8 * g + 4 * fh);
                        shA[g] = *reinterpret_cast<const f32x4*>(tab + 96 + 8 * g + 4 * fh);
                        rA[g] = *reinterpret_cast<const u32x2*>(rb + g * XPLANE);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf2[st]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            SP_KSTAMP(n1)
            SP_KACC(3, n0, n1)
            // epilogue 2: out = relu(acc * scale2 + shift2 + x)
            f32x4 scB[2], shB[2];
            u32x2 rB[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                scB[g] = *reinterpret_cast<const f32x4*>(tab + 64 + 8 * (g + 2) + 4 * fh);
                shB[g] = *reinterpret_cast<const f32x4*>(tab + 96 + 8 * (g + 2) + 4 * fh);
                rB[g] = *reinterpret_cast<const u32x2*>(rb + (g + 2) * XPLANE);
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned d[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sc = g < 2 ? scA[g & 1] : scB[g & 1];
                const f32x4 sh = g < 2 ? shA[g & 1] : shB[g & 1];
                const bf16x4 r4 = __builtin_bit_cast(bf16x4, g < 2 ? rA[g & 1] : rB[g & 1]);
                f32x2 v01, v23;
                v01[0] = acc[4 * g + 0] * sc[0] + sh[0]; v01[1] = acc[4 * g + 1] * sc[1] + sh[1];
                v23[0] = acc[4 * g + 2] * sc[2] + sh[2]; v23[1] = acc[4 * g + 3] * sc[3] + sh[3];
                v01[0] = __builtin_fmaxf(v01[0] + (float)r4[0], 0.f); v01[1] = __builtin_fmaxf(v01[1] + (float)r4[1], 0.f);
                v23[0] = __builtin_fmaxf(v23[0] + (float)r4[2], 0.f); v23[1] = __builtin_fmaxf(v23[1] + (float)r4[3], 0.f);
                d[g][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2));
                d[g][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2));
            }
            // lanes l and l + 32 hold the two halves of each 16-byte chunk of pixel l: swap so that every lane holds whole chunks
            // (lanes < 32: chunks 0 and 2, lanes >= 32: chunks 1 and 3)
            const int oy = tly * TH + oyl, ox = tlx * TW + oxl;
            const unsigned pix = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C) * 2) : OOB;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(d[2 * pr][0], d[2 * pr + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(d[2 * pr][1], d[2 * pr + 1][1], false, false);
                u32x4 o;
                o[0] = s0[0]; o[1] = s1[0]; o[2] = s0[1]; o[3] = s1[1];
                __builtin_amdgcn_raw_buffer_store_b128(o, yr, pix == OOB ? OOB : pix + (unsigned)((2 * pr + fh) * 16), 0, 0);
            }
            SP_KSTAMP(n2)
            SP_KACC(4, n1, n2)
        }
        SP_KSTAMP(kd)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the next strip's halo have landed
        __syncthreads();            // the other buffer is complete; every wave is done with Ts and with this buffer
        SP_KSTAMP(kf)
        SP_KACC(5, kb, kc) SP_KACC(6, kd, kf) SP_KACC(7, ka0, ka)
        cur ^= 1;
    }
#ifdef SP_BB32_DIAG
    {
        SP_KSTAMP(kz)
        unsigned long long rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
        dg[8] = kz - k0; dg[9] = rt1 - rt0;
        if (lane == 0)
            for (int k = 0; k < 10; ++k) sp_bb32_dbg[((blockIdx.x & 255) * 8 + wave) * 10 + k] = dg[k];
    }
#endif
}

}  // namespace s48

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
    // SP_BB32_W8=0: the round-2 four-wave kernel on 8 x 16 tiles (kept for A/B); default: the eight-wave kernel on 8 x 48 strips
    static const bool w8 = [] { const char* e = getenv("SP_BB32_W8"); return !(e && e[0] == '0'); }();
    if (sp_name_query_active()) { sp_name_query_set(w8 ? "basic_block_c32_w8_kernel" : "basic_block_c32_kernel"); return SP_OK; }
    const long long elems = (long long)d->batch * d->in_h * d->in_w * 32;
    SP_REQUIRE(elems < (1ll << 29), "sp_basic_block_c32: tensor too large");
    BlockArgs a;
    a.x = x; a.w1 = w1_packed; a.scale1 = scale1; a.shift1 = shift1; a.w2 = w2_packed; a.scale2 = scale2; a.shift2 = shift2; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.k_pad = d->k_pad; a.batch = d->batch;
    if (w8) {
        SP_REQUIRE(d->in_h < (1 << 20) && d->in_w < (1 << 20), "sp_basic_block_c32: image too large");
        a.tiles_x = (d->in_w + s48::TW - 1) / s48::TW; a.tiles_y = (d->in_h + s48::TH - 1) / s48::TH;
        a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
        const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
        SP_REQUIRE(tiles < (1ll << 31), "sp_basic_block_c32: too many tiles");
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&s48::basic_block_c32_w8_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, s48::LDS_BYTES);
        SP_REQUIRE(attr == hipSuccess, "sp_basic_block_c32: cannot reserve %d bytes of LDS (%s)", s48::LDS_BYTES, hipGetErrorString(attr));
        // persistent, one workgroup per CU; consecutive strips per workgroup, and as many workgroups as keeps the longest share minimal
        const long long per = (tiles + 255) / 256;
        const long long grid = (tiles + per - 1) / per;
        hipLaunchKernelGGL(s48::basic_block_c32_w8_kernel, dim3((unsigned)grid), dim3(512), s48::LDS_BYTES, (hipStream_t)stream, a);
        return sp_check_launch("basic_block_c32_w8_kernel");
    }
    a.tiles_x = (d->in_w + TW - 1) / TW; a.tiles_y = (d->in_h + TH - 1) / TH;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_basic_block_c32: too many tiles");
    const long long grid = tiles < 256 * 2 ? tiles : 256 * 2;     // persistent: two workgroups per CU, filters fetched once each
    hipLaunchKernelGGL(basic_block_c32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    return sp_check_launch("basic_block_c32_kernel");
}
