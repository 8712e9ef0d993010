// conv_wgrad.hip - weight gradients of the conv / transposed-conv family, every layer of a group in ONE launch.
//
//   dW[n][k] = sum_m G[m][n] * A[m][k]          (reduction over the pixels m of one launch grid)
//
// G is a plain NHWC tensor [M][N] (conv: dz, the gradient at the conv output; transposed conv: the layer INPUT x), A is the implicit
// im2col of the other tensor, gathered exactly like the forward kernel gathers its A operand (conv: x with the layer's taps / stride /
// padding; transposed conv k4s2p1: dy with 4x4 taps, stride 2, pad 1).  Replaces loss.backward()'s weight-gradient half for nn.Conv2d /
// nn.ConvTranspose2d (nets/pose_resnet_dconv.py:99-103,158,236-244; processors/ddp_pose_resnet_solver.py:117-119).
//
// Round 3 structure (round 2: 128x128 tiles, 4 waves, one K step in flight, one launch + one slab-reduce launch PER LAYER):
//   * a "unit" = one dW tile (TG columns of G x TA columns of A: 256x256 / 256x128 / 128x256 / 256x64 / 64x256) over a range of pixels;
//     units have about equal cost, the pixel ranges of a layer are cut at a size that depends on the layer alone, so the bits of a
//     layer's gradient do not depend on what else is in the launch;
//   * one launch runs the units of up to 16 layers (a gradient bucket's worth): blockIdx -> (layer, tile, pixel range) through a
//     prefix table in the kernel arguments - no device-side job table to upload, activations may move between steps;
//   * 8 waves per workgroup, operands global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`) into a ring of NS stages with counted
//     `s_waitcnt vmcnt(N)` and one raw `s_barrier` per stage, as csrc/conv_ring.hip; both operands are pixel-major in memory while
//     the MFMA wants 8 consecutive pixels per lane, i.e. a transposed read: `ds_read_b64_tr_b16` (bf16) straight from the row-major
//     LDS image, XOR-swizzled on 64-byte segments so that the four rows a 16-lane group addresses fall on disjoint banks (the swizzle
//     is applied by the per-lane SOURCE offset of the DMA: its LDS destination is lane-linear); fp32 reads one float per lane;
//   * every unit writes its partial tile into a slab [split][n][k] with plain stores; ONE fold launch per group sums a layer's splits
//     in index order and scatters into the reference weight layout (Conv2d [O,I,kh,kw], ConvTranspose2d [I,O,kh,kw]) -
//     deterministic, no float atomics, no inter-workgroup hand-off inside a launch.
#include "sp_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int WG_MAXJ = 16;              // layers per launch (kernel-argument table)
constexpr unsigned OOB = 0x80000000u;    // every tensor is < 2 GiB (host-checked): an offset the descriptor's range check rejects

struct WgLayer {
    const void* g;       // [M][n_ld]
    const void* a;       // NHWC [B][in_h][in_w][c_in]
    float* slab;         // [splits][n_rows][k_cols] fp32
    int M, n_ld;
    int in_h, in_w, c_in;
    int grid_w, ghw;
    unsigned mg_w, mg_hw;   // exact division by grid_w / ghw for dividends < 2^31: q = umulhi(n, magic) >> shift (magic 0: divisor 1)
    int sh_w, sh_hw;
    int taps_w, k_total;    // valid A columns: taps_h * taps_w * c_in
    int stride, dy0, dy_step, dx0, dx_step;
    int tiles_a, tiles;     // tiles = tiles_g * tiles_a
    int px_per_split;       // multiple of the stage depth
    int n_rows, k_cols;     // slab extents (multiples of the tile)
    int g_bytes, a_bytes;
    int plain_a;            // 1x1, stride 1, no padding: A is the plain [M][c_in] tensor
    int tile;               // index into kTiles
    int unit_end;           // units of layers 0..this one
};
struct WgArgs {
    WgLayer L[WG_MAXJ];
    int n_layers;
    int units;           // units of the launch (>= gridDim.x: a workgroup walks units blockIdx.x, blockIdx.x + gridDim.x, ...)
};

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4 r;                                   // (readfirstlane: an "s" asm operand must be provably wave-uniform)
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane(bytes);                              // num_records (bytes)
    r[3] = 0x00020000u;
    return r;
}

// One LDS-DMA piece (see csrc/conv_ring.hip): 64 lanes x 16 bytes, lane l's bytes from `rsrc` base + voff (zeros when voff is out of
// range), written to LDS at lds_addr + 16 * l.  Issued from inline asm so that hipcc does not drain the ring in front of every LDS read;
// ordered by this file's own counted `s_waitcnt vmcnt(N)` + s_barrier.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, u32x4 rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                 "s"(rsrc)
                 : "memory");   // (hipcc rejects "m0" in a clobber list: reserved register; it keeps nothing live in M0 across statements)
}

__device__ __forceinline__ int fdiv(int n, unsigned magic, int sh) {
    const int q = (int)(__umulhi((unsigned)n, magic) >> sh);      // (computed either way: a select, not a branch, inside the stage loop)
    return magic ? q : n;
}

// One unit: the TG x TA tile (tg, ta) of layer `p` over pixels [m_begin, m_end), partial sums to slab split `split`.
//   BF16: 2-byte operands, v_mfma_f32_32x32x16_bf16, PIX pixels per stage (multiple of 16); else fp32 operands, v_mfma_f32_32x32x2_f32.
template <bool BF16, int TG, int TA, int WR, int WC, int PIX, int NS>
__device__ __forceinline__ void wgrad_unit(const WgLayer& p, const int tg, const int ta, const int split, const int m_begin, const int m_end,
                                           unsigned char* const ring) {
    static_assert(WR * WC == 8, "8 waves per workgroup");
    constexpr int EB = BF16 ? 2 : 4, EPC = 16 / EB;         // bytes per element, elements per 16-byte chunk
    constexpr int WM = TG / WR, WN = TA / WC, TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int RBG = TG * EB, RBA = TA * EB;             // bytes per pixel row of the two LDS images
    constexpr int SBG = PIX * RBG, SBA = PIX * RBA, SB = SBG + SBA;
    static_assert(SBG % 8192 == 0 && SBA % 8192 == 0, "whole 1-KiB pieces per wave");
    constexpr int LG = SBG / 8192, LA = SBA / 8192, L = LG + LA;   // pieces per wave per stage
    constexpr int RPG = 1024 / RBG, RPA = 1024 / RBA;       // pixel rows per piece
    constexpr int LPRG = RBG / 16, LPRA = RBA / 16;         // lanes (16-byte chunks) per row
    constexpr int D = NS - 1;
    constexpr int KS = BF16 ? PIX / 16 : PIX / 2;           // MFMA k-steps per stage
    constexpr int NM = KS * TM * TN;                        // MFMAs per wave per stage
    static_assert(NM >= L, "a DMA piece per MFMA at most");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int g0 = tg * TG, k0 = ta * TA;
    // what the stage loop reads of the layer record, in registers: the DMA asm statements are memory barriers to the compiler, which
    // would otherwise re-load these kernel arguments (and wait on lgkmcnt, i.e. on the LDS reads too) after every piece
    const int n_ld = p.n_ld, c_in = p.c_in, in_h = p.in_h, in_w = p.in_w, grid_w = p.grid_w, ghw = p.ghw, stride = p.stride;
    const unsigned mg_w = p.mg_w, mg_hw = p.mg_hw;
    const int sh_w = p.sh_w, sh_hw = p.sh_hw;
    const bool plain_a = p.plain_a != 0;
    const unsigned ring_lds = (unsigned)(size_t)(lds_void_t*)ring;

    // the G descriptor ends at this unit's last pixel: rows beyond m_end read as zeros without a per-lane test (same for a plain A)
    const long long g_end = (long long)m_end * p.n_ld * EB;
    const u32x4 gr = make_rsrc(p.g, (unsigned)(g_end < p.g_bytes ? g_end : p.g_bytes));
    const long long a_end = (long long)m_end * p.c_in * EB;
    const u32x4 ar = make_rsrc(p.a, (unsigned)((p.plain_a && a_end < p.a_bytes) ? a_end : p.a_bytes));

    // chunk swizzle of a bf16 image: physical 64-byte segment s of row r holds logical segment s ^ f(r) (inside aligned groups of four
    // segments; rows of 128 bytes have two segments and flip on bit 1 of r): the four rows of a transposing read hit disjoint banks
    auto lchunk_of = [](int c16, int r, int rb) __attribute__((always_inline)) {
        if (!BF16) return c16;
        const int pseg = c16 >> 2, sub = c16 & 3;
        const int lseg = rb == 128 ? (pseg ^ ((r >> 1) & 1)) : ((pseg & ~3) | ((pseg ^ r) & 3));
        return lseg * 4 + sub;
    };

    // ---- loader statics: this lane's chunk of every piece its wave issues ----
    unsigned g_voff[LG];
#pragma unroll
    for (int j = 0; j < LG; ++j) {
        const int r = (wave + 8 * j) * RPG + lane / LPRG;
        const int col = g0 + lchunk_of(lane % LPRG, r, RBG) * EPC;
        g_voff[j] = col < p.n_ld ? (unsigned)((r * p.n_ld + col) * EB) : OOB;
    }
    int a_row[LA], a_dy[LA], a_dx[LA], a_coff[LA];      // gathered A: row in the stage, tap shift, channel byte offset (-1: no such column)
    unsigned a_voff[LA];                                  // plain A
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int r = (wave + 8 * j) * RPA + lane / LPRA;
        const int kc = k0 + lchunk_of(lane % LPRA, r, RBA) * EPC;
        const bool valid = kc < p.k_total;
        const int tap = kc / p.c_in, c_off = kc - tap * p.c_in;
        const int ty = tap / p.taps_w, tx = tap - ty * p.taps_w;
        a_row[j] = r;
        a_dy[j] = ty * p.dy_step + p.dy0;
        a_dx[j] = tx * p.dx_step + p.dx0;
        a_coff[j] = valid ? c_off * EB : -1;
        a_voff[j] = valid ? (unsigned)((r * p.c_in + kc) * EB) : OOB;
    }
    const int ST = (m_end - m_begin + PIX - 1) / PIX;      // stages of this unit (>= 1)
    int ld_m0 = m_begin, ld_slot = 0;
    auto loader_piece = [&](int o) __attribute__((always_inline)) {   // o in [0, L): one 1-KiB piece of the stage at pixel ld_m0
        const unsigned slot = ring_lds + (unsigned)(ld_slot * SB);
        if (o < LG) {
            dma16(slot + (unsigned)((wave + 8 * o) * 1024), g_voff[o] + (unsigned)(ld_m0 * n_ld * EB), gr);
        } else {
            const int j = o - LG;
            unsigned off;
            if (plain_a) {
                off = a_voff[j] + (unsigned)(ld_m0 * c_in * EB);
            } else {
                const int m = ld_m0 + a_row[j];
                const int b = fdiv(m, mg_hw, sh_hw), rem = m - b * ghw;
                const int gy = fdiv(rem, mg_w, sh_w), gx = rem - gy * grid_w;
                const int iy = gy * stride + a_dy[j], ix = gx * stride + a_dx[j];
                const int ok = (int)(a_coff[j] >= 0) & (int)(m < m_end) & (int)((unsigned)iy < (unsigned)in_h) & (int)((unsigned)ix < (unsigned)in_w);
                const unsigned addr = (unsigned)(((b * in_h + iy) * in_w + ix) * c_in * EB + a_coff[j]);
                off = ok ? addr : OOB;
            }
            dma16(slot + (unsigned)(SBG + (wave + 8 * j) * 1024), off, ar);
        }
    };
    auto loader_advance = [&]() __attribute__((always_inline)) {
        ld_slot = (ld_slot + 1 == NS) ? 0 : ld_slot + 1;
        ld_m0 += PIX;
    };

#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d < ST) {
#pragma unroll
            for (int o = 0; o < L; ++o) loader_piece(o);
            loader_advance();
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    // ---- fragment addressing ----
    // bf16: ds_read_b64_tr_b16 delivers, per 16-lane group, a 4-row x 16-column block column-major (lane i gets column i of the 4 rows):
    // two of them build the 8-pixel operand of a lane.  lane -> 16-lane group g4 (column half g4 & 1, pixel half g4 >> 1); inside the
    // group lane 4q + pp addresses row q, columns 4pp .. 4pp+3.
    // fp32: one float per lane, pixel 2 kp + (lane >> 5), column lane & 31.
    int fg[TM], fa[TN];
    {
        const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const int r0 = 8 * (g4 >> 1) + q;
        auto frag_addr = [&](int col0, int rb) __attribute__((always_inline)) {
            if (BF16) {
                const int c = col0 + 16 * (g4 & 1) + 4 * pp;
                const int seg = c >> 5;
                const int pseg = rb == 128 ? (seg ^ ((r0 >> 1) & 1)) : ((seg & ~3) | ((seg ^ r0) & 3));
                return r0 * rb + pseg * 64 + (c & 31) * 2;
            }
            return (lane >> 5) * rb + (col0 + (lane & 31)) * 4;
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) fg[i] = frag_addr(wr * WM + i * 32, RBG);
#pragma unroll
        for (int n = 0; n < TN; ++n) fa[n] = SBG + frag_addr(wc * WN + n * 32, RBA);
    }
    using frag_t = std::conditional_t<BF16, s16x8, float>;
    frag_t qg[2][TM], qa[2][TN];
    const unsigned char* sl = ring;
    auto read_frags = [&](int ks, int buf) __attribute__((always_inline)) {
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned char* b = sl + fg[i] + ks * 16 * RBG;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + 4 * RBG));
                qg[buf][i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const unsigned char* b = sl + fa[n] + ks * 16 * RBA;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(b + 4 * RBA));
                qa[buf][n] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) qg[buf][i] = *reinterpret_cast<const float*>(sl + fg[i] + ks * 2 * RBG);
#pragma unroll
            for (int n = 0; n < TN; ++n) qa[buf][n] = *reinterpret_cast<const float*>(sl + fa[n] + ks * 2 * RBA);
        }
    };

#define SP_SB() __builtin_amdgcn_sched_barrier(0)
    int slot = 0;
    auto stage = [&](auto issue_tag) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issue_tag)::value;   // a stage D ahead exists and is requested during this one
        sl = ring + slot * SB;
        read_frags(0, 0);
        SP_SB();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_frags(ks + 1, (ks + 1) & 1);
            SP_SB();
#pragma unroll
            for (int t = 0; t < TM * TN; ++t) {
                const int i = t / TN, n = t % TN;
                if constexpr (BF16)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qg[ks & 1][i]), __builtin_bit_cast(bf16x8, qa[ks & 1][n]),
                                                                        acc[i][n], 0, 0, 0);
                else
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(qg[ks & 1][i], qa[ks & 1][n], acc[i][n], 0, 0, 0);
                if constexpr (ISSUE) {
                    const int qq = ks * TM * TN + t;          // every DMA piece in the shadow of an MFMA of its own
#pragma unroll
                    for (int o = (qq * L) / NM; o < ((qq + 1) * L) / NM; ++o) loader_piece(o);
                }
                SP_SB();
            }
        }
        if constexpr (ISSUE) loader_advance();
        slot = (slot + 1 == NS) ? 0 : slot + 1;
    };
    // stage g has landed once at most the D-1 younger stages are still outstanding (each wave waits for ITS pieces; the barrier then makes
    // every wave's pieces visible, and says everyone is done reading the slot the new DMA overwrites)
    int g = 0;
    for (; g + D < ST; ++g) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * L) : "memory");
        __builtin_amdgcn_s_barrier();
        SP_SB();
        stage(std::true_type{});
    }
    for (; g < ST; ++g) {                                  // the last D stages: nothing left to request
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SP_SB();
        stage(std::false_type{});
    }
#undef SP_SB

    // ---- partial tile -> slab [split][n][k]; C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
    const int fr = lane & 31, fh = lane >> 5;
    float* out = p.slab + ((size_t)split * p.n_rows + g0 + wr * WM) * p.k_cols + k0 + wc * WN + fr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                out[(size_t)row * p.k_cols + n * 32] = acc[i][n][r];
            }
}

// One launch = the units of up to WG_MAXJ layers, whatever their tile shapes: a workgroup looks its unit up in the prefix table and runs
// the matching instantiation (a launch per tile shape would serialise half-empty grids on the stream: measured 250 us of 1,060 per step)
template <bool BF16>
__global__ __launch_bounds__(512, 2) void conv_wgrad_group_kernel(const WgArgs args) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // blocks b and b + 8 share an XCD (and its L2): every XCD takes a contiguous run of units - the tiles of one pixel range.
    // The grid may be smaller than the number of units (args.units; a capped grid leaves CUs to the launches of the step's main chain):
    // a workgroup then walks units k, k + gridDim.x, ...
    const int G = args.units;
  for (int orig = blockIdx.x; orig < G; orig += gridDim.x) {
    if (orig != (int)blockIdx.x) __syncthreads();          // the previous unit's last LDS reads are done before the next unit's DMA lands
    int u;
    {
        const int xcd = orig & 7, q = G >> 3, r = G & 7;
        u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int l = 0, begin = 0;
    while (l + 1 < args.n_layers && u >= args.L[l].unit_end) { begin = args.L[l].unit_end; ++l; }
    const WgLayer& p = args.L[l];
    const int local = u - begin;
    const int split = local / p.tiles, t = local - split * p.tiles;
    const int tg = t / p.tiles_a, ta = t - tg * p.tiles_a;
    const int m_begin = split * p.px_per_split;
    const int m_end = min(p.M, m_begin + p.px_per_split);
    constexpr int H = BF16 ? 1 : 2;     // fp32 stages hold half the pixels (same bytes)
    switch (p.tile) {
        case 0: wgrad_unit<BF16, 256, 256, 2, 4, 32 / H, 4>(p, tg, ta, split, m_begin, m_end, smem); break;
        case 1: wgrad_unit<BF16, 256, 128, 4, 2, 32 / H, 4>(p, tg, ta, split, m_begin, m_end, smem); break;
        case 2: wgrad_unit<BF16, 128, 256, 2, 4, 32 / H, 4>(p, tg, ta, split, m_begin, m_end, smem); break;
        case 3: wgrad_unit<BF16, 256, 64, 4, 2, 64 / H, 3>(p, tg, ta, split, m_begin, m_end, smem); break;
        default: wgrad_unit<BF16, 64, 256, 1, 8, 64 / H, 3>(p, tg, ta, split, m_begin, m_end, smem); break;
    }
  }
}

// ---- fold: dst[n*s_n + c*s_c + (ty*kw + tx)] = sum_s slab[s][n][(ty*taps_w + tx)*c_in + c]   for n < n_valid, c < c_valid, tx < kw,
// every layer of a group in one launch, splits summed in index order.  A block takes `rows` consecutive n: it reads their valid slab
// columns as float4 (coalesced along k = (tap, c)), and - the reference layouts keep a row n contiguous as [c][ty][tx] - turns each row
// into that order through LDS, so the stores are coalesced too (a direct store would scatter 4-byte values 36 / 64 bytes apart). ----
constexpr int FOLD_LDS_FLOATS = 5120;
constexpr int FOLD_PART_ITEMS = 2048;          // float4 partial sums a block may hold in LDS (32 KB)
struct FoldLayer {
    const float* slab;
    float* dst;
    long long s_n, s_c;
    int splits, n_rows, k_cols, n_valid, c_in, c_valid, taps_h, taps_w, kw;
    int k_total;         // taps_h * taps_w * c_in
    int rows;            // n per block
    int sg;              // split groups summed by different threads (power of two, 1: none): group g takes splits g, g + sg, ...
    int dense;           // s_c == taps_h * kw and rows * c_valid * s_c <= FOLD_LDS_FLOATS: rows go through LDS
    int block_end;       // blocks of layers 0..this one
};
struct FoldArgs {
    FoldLayer L[WG_MAXJ];
    int n_layers;
};

__global__ __launch_bounds__(256) void wgrad_fold_kernel(const FoldArgs args) {
    __shared__ float rowbuf[FOLD_LDS_FLOATS];
    __shared__ f32x4 part[FOLD_PART_ITEMS];
    int l = 0, begin = 0;
    const int b = blockIdx.x;
    while (l + 1 < args.n_layers && b >= args.L[l].block_end) { begin = args.L[l].block_end; ++l; }
    const FoldLayer& p = args.L[l];
    const int n0 = (b - begin) * p.rows;
    const int rows = min(p.rows, p.n_valid - n0);
    const int k4 = p.k_total >> 2;                       // float4 items per row
    const int items = rows * k4;
    const size_t stride = (size_t)p.n_rows * p.k_cols;
    const int row_len = p.c_valid * (int)p.s_c;
    const int SG = p.sg;
    // every order below is fixed by the layer's shape: bit-reproducible
    for (int it = threadIdx.x; it < items * SG; it += 256) {
        const int g = it / items, i = it - g * items;
        const int r = i / k4, k = (i - r * k4) * 4;
        const float* src = p.slab + (size_t)(n0 + r) * p.k_cols + k;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;       // 4 independent chains keep 4 loads in flight
        int sp = g;
        for (; sp + 3 * SG < p.splits; sp += 4 * SG) {
            s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * stride);
            s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + SG) * stride);
            s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 2 * SG) * stride);
            s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 3 * SG) * stride);
        }
        for (; sp < p.splits; sp += SG) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)sp * stride);
        part[it] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < items; i += 256) {
        f32x4 v = part[i];
        for (int g = 1; g < SG; ++g) v += part[g * items + i];
        const int r = i / k4, k = (i - r * k4) * 4;
        const int tap = k / p.c_in, c = k - tap * p.c_in;    // the 4 columns are channels c..c+3 of one tap (c_in % 4 == 0)
        const int ty = tap / p.taps_w, tx = tap - ty * p.taps_w;
        if (ty >= p.taps_h || tx >= p.kw) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e >= p.c_valid) continue;
            const int pos = (c + e) * (int)p.s_c + ty * p.kw + tx;
            if (p.dense) rowbuf[r * row_len + pos] = v[e];
            else p.dst[(long long)(n0 + r) * p.s_n + pos] = v[e];
        }
    }
    if (!p.dense) return;
    __syncthreads();
    if (p.s_n == row_len) {                                  // the rows of this block are one contiguous run of dst
        float* d = p.dst + (long long)n0 * p.s_n;
        for (int i = threadIdx.x; i < rows * row_len; i += 256) d[i] = rowbuf[i];
    } else {
        for (int i = threadIdx.x; i < rows * row_len; i += 256) {
            const int r = i / row_len;
            p.dst[(long long)(n0 + r) * p.s_n + (i - r * row_len)] = rowbuf[i];
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
struct TileCfg { int tg, ta, pix_bf16, ns; double ns_per_px_bf16, ns_per_px_f32; };
// ns_per_px: what one workgroup spends per pixel of its range (bf16: ~4 TFLOP/s per CU on the 256x256 tile, the narrow tiles stream
// at a CU's share of HBM; fp32: 0.55 TFLOP/s per CU): sizes the pixel ranges so that units cost about the same
constexpr TileCfg kTiles[5] = {
    {256, 256, 32, 4, 30.0, 240.0}, {256, 128, 32, 4, 18.0, 120.0}, {128, 256, 32, 4, 18.0, 120.0}, {256, 64, 64, 3, 25.0, 60.0}, {64, 256, 64, 3, 25.0, 60.0}};

int pick_tile(int n, int k) {
    const int g = n >= 192 ? 256 : (n >= 96 ? 128 : 64), a = k >= 192 ? 256 : (k >= 96 ? 128 : 64);
    if (g == 256 && a == 256) return 0;
    if (g == 256 && a == 128) return 1;
    if (g == 128 && a == 256) return 2;
    if (g == 256 && a == 64) return 3;
    if (g == 64 && a == 256) return 4;
    if (g == 128 && a == 128) return 2;      // (not a ResNet-50 shape: correct, half the tile idle)
    if (g == 128 && a == 64) return 3;
    return 4;                                  // 64 x 128, 64 x 64
}

void magic_for(int d, unsigned* magic, int* shift) {   // q = umulhi(n, magic) >> shift, exact for 0 <= n < 2^31
    if (d <= 1) { *magic = 0; *shift = 0; return; }
    int l = 0;
    while ((1ll << l) < d) ++l;
    *magic = (unsigned)(((1ull << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);
    *shift = l - 1;
}

double unit_ns(bool bf16) {
    static double cached[2] = {0.0, 0.0};
    double& c = cached[bf16 ? 1 : 0];
    if (c == 0.0) {
        const char* e = getenv(bf16 ? "SP_WGRAD_UNIT_NS_BF16" : "SP_WGRAD_UNIT_NS_F32");   // development knob (tools/bench_wgrad.py)
        c = (e && atof(e) > 0) ? atof(e) : (bf16 ? 50000.0 : 120000.0);   // (fp32: 200 / 120 / 80 us measured 3,963 / 3,597 / 3,853 us for the 57 layers)
    }
    return c;
}

struct Plan {                 // how one job is cut: depends on the job alone
    int tile, tiles_g, tiles_a, splits, px_per_split, n_rows, k_cols;
    long long slab_floats;
};

int plan_job(const sp_wgrad_job& j, Plan* pl) {
    const sp_conv_desc* d = &j.desc;
    SP_REQUIRE(d->stride_x == 0 || d->stride_x == d->stride, "sp_conv2d_wgrad: separate x / y strides are a forward-only feature");
    SP_REQUIRE(j.g && j.a && j.dw, "sp_conv2d_wgrad: null pointer");
    const bool bf16 = d->flags & SP_CONV_BF16;
    const int es = bf16 ? 2 : 4, epc = 16 / es;
    SP_REQUIRE(d->c_in > 0 && d->c_in % epc == 0 && d->taps_h > 0 && d->taps_w > 0, "sp_conv2d_wgrad: bad c_in / taps");
    SP_REQUIRE(j.g_channels > 0 && j.g_channels % epc == 0 && j.n_valid > 0 && j.n_valid <= j.g_channels, "sp_conv2d_wgrad: bad g_channels/n_valid");
    SP_REQUIRE(j.c_valid > 0 && j.c_valid <= d->c_in && j.kw_valid > 0 && j.kw_valid <= d->taps_w, "sp_conv2d_wgrad: bad c_valid/kw_valid");
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    const long long a_elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    SP_REQUIRE(M > 0 && M * j.g_channels < (1ll << 29) && a_elems < (1ll << 29), "sp_conv2d_wgrad: tensor too large");
    const int k_total = d->taps_h * d->taps_w * d->c_in;
    SP_REQUIRE(k_total <= 4 * FOLD_PART_ITEMS, "sp_conv2d_wgrad: taps * c_in = %d exceeds %d", k_total, 4 * FOLD_PART_ITEMS);
    pl->tile = pick_tile(j.n_valid, k_total);
    const TileCfg& t = kTiles[pl->tile];
    pl->tiles_g = (j.n_valid + t.tg - 1) / t.tg;
    pl->tiles_a = (k_total + t.ta - 1) / t.ta;
    pl->n_rows = pl->tiles_g * t.tg;
    pl->k_cols = pl->tiles_a * t.ta;
    const int pix = bf16 ? t.pix_bf16 : t.pix_bf16 / 2;
    const double unit_px = unit_ns(bf16) / (bf16 ? t.ns_per_px_bf16 : t.ns_per_px_f32);
    long long splits = (long long)((double)M / unit_px + 0.5);
    if (splits < 1) splits = 1;
    // A layer with more units than CUs (the stem: 393,216 pixels -> 384 units of 50 us) is the step's TAIL when its gradient is the last to
    // exist - 1.5 rounds of workgroups take the time of 2.  Its splits are rounded so that the units fill whole rounds of 256 (round 5: the
    // stem's 512 units of 37 us finish in 2 x 37 instead of 2 x 50 us).  Still a function of the layer alone: bits do not depend on the group.
    static const bool fill_rounds = getenv("SP_WGRAD_FILL_ROUNDS") && atoi(getenv("SP_WGRAD_FILL_ROUNDS"));   // (env: development knob, off: A/B below)
    if (fill_rounds) {
        const long long tiles = (long long)pl->tiles_g * pl->tiles_a, units = tiles * splits;
        if (units > 256) {
            const long long rounds = (units + 128) / 256;
            splits = rounds * 256 / tiles;
            if (splits < 1) splits = 1;
        }
    }
    long long pps = ((M + splits - 1) / splits + pix - 1) / pix * pix;
    splits = (M + pps - 1) / pps;
    pl->splits = (int)splits;
    pl->px_per_split = (int)pps;
    pl->slab_floats = splits * (long long)pl->n_rows * pl->k_cols;
    return SP_OK;
}

template <bool BF16>
int launch_group(const WgArgs& a, int units, hipStream_t s) {
    size_t lds = 0;
    for (int i = 0; i < a.n_layers; ++i) {
        const TileCfg& t = kTiles[a.L[i].tile];
        const size_t need = (size_t)t.ns * t.pix_bf16 * (t.tg + t.ta) * 2;     // (fp32: half the pixels of twice the bytes)
        if (need > lds) lds = need;
    }
    const void* fn = reinterpret_cast<const void*>(&conv_wgrad_group_kernel<BF16>);
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // per device: set before every launch
    if (e != hipSuccess) {
        sp_set_error("conv_wgrad: hipFuncSetAttribute(max dynamic LDS = %zu) failed: %s", lds, hipGetErrorString(e));
        return SP_ELAUNCH;
    }
    static const int cap = getenv("SP_WGRAD_GRID_CAP") ? atoi(getenv("SP_WGRAD_GRID_CAP")) : 0;      // (env: development knob; 0 = one workgroup per unit)
    WgArgs b = a;
    b.units = units;
    hipLaunchKernelGGL((conv_wgrad_group_kernel<BF16>), dim3(cap > 0 && cap < units ? cap : units), dim3(512), lds, s, b);
    return sp_check_launch("conv_wgrad_group_kernel");
}

}  // namespace

extern "C" int sp_conv2d_wgrad_workspace(const sp_wgrad_job* jobs, int n_jobs, int64_t* bytes) {
    SP_REQUIRE(jobs && bytes && n_jobs > 0, "sp_conv2d_wgrad_workspace: bad arguments");
    long long total = 0;
    for (int i = 0; i < n_jobs; ++i) {
        Plan pl;
        const int rc = plan_job(jobs[i], &pl);
        if (rc != SP_OK) return rc;
        total += pl.slab_floats * 4;
    }
    *bytes = total;
    return SP_OK;
}

extern "C" int sp_conv2d_wgrad_batched(const sp_wgrad_job* jobs, int n_jobs, void* workspace, int64_t workspace_bytes, void* stream) {
    SP_REQUIRE(jobs && n_jobs > 0 && workspace, "sp_conv2d_wgrad_batched: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const bool bf16 = jobs[0].desc.flags & SP_CONV_BF16;
    Plan plans_stack[64];
    SP_REQUIRE(n_jobs <= 64, "sp_conv2d_wgrad_batched: at most 64 jobs per call (%d given)", n_jobs);
    long long need = 0;
    for (int i = 0; i < n_jobs; ++i) {
        SP_REQUIRE(((jobs[i].desc.flags & SP_CONV_BF16) != 0) == bf16, "sp_conv2d_wgrad_batched: jobs of one call share the operand dtype");
        const int rc = plan_job(jobs[i], &plans_stack[i]);
        if (rc != SP_OK) return rc;
        need += plans_stack[i].slab_floats * 4;
    }
    SP_REQUIRE(need <= workspace_bytes, "sp_conv2d_wgrad: workspace too small (%lld B needed, %lld given)", need, (long long)workspace_bytes);
    // slabs: one after the other in job order
    float* slab[64];
    {
        float* w = reinterpret_cast<float*>(workspace);
        for (int i = 0; i < n_jobs; ++i) { slab[i] = w; w += plans_stack[i].slab_floats; }
    }
    // ---- unit launches: <= WG_MAXJ layers each, any mix of tile shapes; the layers with the costliest units first, so that the launch
    // ends on its cheapest units (the grid's tail is as short as a unit) ----
    int order[64];
    for (int i = 0; i < n_jobs; ++i) order[i] = i;
    for (int i = 1; i < n_jobs; ++i) {                       // insertion sort, stable: descending time per unit
        const int v = order[i];
        auto cost = [&](int q) { const TileCfg& t = kTiles[plans_stack[q].tile]; return (bf16 ? t.ns_per_px_bf16 : t.ns_per_px_f32) * plans_stack[q].px_per_split; };
        int k = i;
        while (k > 0 && cost(order[k - 1]) < cost(v)) { order[k] = order[k - 1]; --k; }
        order[k] = v;
    }
    {
        WgArgs a;
        a.n_layers = 0;
        int units = 0;
        auto flush = [&]() -> int {
            if (a.n_layers == 0) return SP_OK;
            const int rc = bf16 ? launch_group<true>(a, units, s) : launch_group<false>(a, units, s);
            a.n_layers = 0;
            units = 0;
            return rc;
        };
        for (int oi = 0; oi < n_jobs; ++oi) {
            const int i = order[oi];
            const Plan& pl = plans_stack[i];
            const sp_wgrad_job& j = jobs[i];
            const sp_conv_desc* d = &j.desc;
            const int es = bf16 ? 2 : 4;
            WgLayer& L = a.L[a.n_layers];
            L.g = j.g; L.a = j.a; L.slab = slab[i];
            L.M = d->batch * d->grid_h * d->grid_w; L.n_ld = j.g_channels;
            L.in_h = d->in_h; L.in_w = d->in_w; L.c_in = d->c_in;
            L.grid_w = d->grid_w; L.ghw = d->grid_h * d->grid_w;
            magic_for(L.grid_w, &L.mg_w, &L.sh_w);
            magic_for(L.ghw, &L.mg_hw, &L.sh_hw);
            L.taps_w = d->taps_w; L.k_total = d->taps_h * d->taps_w * d->c_in;
            L.stride = d->stride; L.dy0 = d->dy0; L.dy_step = d->dy_step; L.dx0 = d->dx0; L.dx_step = d->dx_step;
            L.tiles_a = pl.tiles_a; L.tiles = pl.tiles_g * pl.tiles_a;
            L.px_per_split = pl.px_per_split;
            L.n_rows = pl.n_rows; L.k_cols = pl.k_cols;
            L.g_bytes = (int)((long long)L.M * j.g_channels * es);
            L.a_bytes = (int)((long long)d->batch * d->in_h * d->in_w * d->c_in * es);
            L.plain_a = (d->taps_h == 1 && d->taps_w == 1 && d->stride == 1 && d->dy0 == 0 && d->dx0 == 0 && d->in_h == d->grid_h && d->in_w == d->grid_w) ? 1 : 0;
            L.tile = pl.tile;
            units += L.tiles * pl.splits;
            L.unit_end = units;
            if (++a.n_layers == WG_MAXJ) {
                const int rc = flush();
                if (rc != SP_OK) return rc;
            }
        }
        const int rc = flush();
        if (rc != SP_OK) return rc;
    }
    // ---- fold launches ----
    FoldArgs f;
    f.n_layers = 0;
    int blocks = 0;
    auto flush_fold = [&]() -> int {
        if (f.n_layers == 0) return SP_OK;
        hipLaunchKernelGGL(wgrad_fold_kernel, dim3(blocks), dim3(256), 0, s, f);
        f.n_layers = 0;
        blocks = 0;
        return sp_check_launch("wgrad_fold_kernel");
    };
    for (int i = 0; i < n_jobs; ++i) {
        const Plan& pl = plans_stack[i];
        const sp_wgrad_job& j = jobs[i];
        FoldLayer& L = f.L[f.n_layers];
        L.slab = slab[i]; L.dst = j.dw; L.s_n = j.dst_stride_n; L.s_c = j.dst_stride_c;
        L.splits = pl.splits; L.n_rows = pl.n_rows; L.k_cols = pl.k_cols; L.n_valid = j.n_valid;
        L.c_in = j.desc.c_in; L.c_valid = j.c_valid; L.taps_h = j.desc.taps_h; L.taps_w = j.desc.taps_w; L.kw = j.kw_valid;
        L.k_total = L.taps_h * L.taps_w * L.c_in;
        const long long row_len = (long long)j.c_valid * j.dst_stride_c;
        L.rows = 4096 / L.k_total;                       // ~1,024 float4 items per block
        if (L.rows < 1) L.rows = 1;
        if (L.rows > 16) L.rows = 16;
        L.sg = 1;
        const int k4 = L.k_total / 4;
        if (pl.splits >= 8) {                              // few rows x many splits: sum groups of splits side by side
            while (L.sg < 16 && 2 * L.sg * 4 <= pl.splits && k4 * 2 * L.sg <= FOLD_PART_ITEMS) L.sg *= 2;
        }
        while (L.rows > 1 && (long long)L.rows * k4 * L.sg > FOLD_PART_ITEMS / 2) --L.rows;
        while (L.rows > 1 && L.rows * row_len > FOLD_LDS_FLOATS) --L.rows;
        L.dense = (j.dst_stride_c == (long long)L.taps_h * L.kw && L.rows * row_len <= FOLD_LDS_FLOATS) ? 1 : 0;
        blocks += (j.n_valid + L.rows - 1) / L.rows;
        L.block_end = blocks;
        if (++f.n_layers == WG_MAXJ) {
            const int rc = flush_fold();
            if (rc != SP_OK) return rc;
        }
    }
    return flush_fold();
}

extern "C" int sp_conv2d_wgrad(const sp_conv_desc* d, const void* g, int g_channels, const void* a, int n_valid, int c_valid, int kw_valid,
                               int64_t dst_stride_n, int64_t dst_stride_c, float* dw, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    SP_REQUIRE(d && g && a && dw && workspace, "sp_conv2d_wgrad: null pointer");
    sp_wgrad_job j;
    j.desc = *d;
    j.g = g; j.g_channels = g_channels; j.a = a;
    j.n_valid = n_valid; j.c_valid = c_valid; j.kw_valid = kw_valid;
    j.dst_stride_n = dst_stride_n; j.dst_stride_c = dst_stride_c; j.dw = dw;
    return sp_conv2d_wgrad_batched(&j, 1, workspace, workspace_bytes, stream);
}
