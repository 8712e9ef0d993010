// conv_igemm.hip - fp32 implicit-GEMM convolution family on the gfx950 matrix cores.
//
// One kernel template covers every conv / transposed-conv of the reference networks
// (nets/pose_resnet_dconv.py:99-103,158,210,236-244,173-178; nets/commons.py:31-32; nets/pose_hrnet.py):
//
//   D[m][n] = sum_k A[m][k] * Wp[n][k]
//   m = (b, gy, gx)   output pixel of this phase            (rows)
//   n = output channel                                      (columns)
//   k = (ty, tx, c)   filter tap x input channel            (depth, c fastest = NHWC contiguous)
//
// A is never materialised: each 16-byte chunk (4 channels of one tap of one pixel) is gathered straight
// from the NHWC activation with zero fill outside the image.  The transposed conv k4 s2 p1 is launched as
// its 4 output phases (blockIdx.y), each a dense 2x2-tap conv, so no multiply is wasted on the zeros a
// naive "dilate the input" formulation inserts.
//
// bf16 variant (SP_CONV_BF16): the same kernel with bf16 activations/weights (8 elements per 16-B chunk, K tile 64),
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bf16 NHWC output (the NCHW heat-map output stays fp32).
//
// Math: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate) - 256 FLOP/clk/CU, the fp32
// matrix peak of the chip (157 TFLOP/s).  Block tile BM x BN x 32, 4 waves, each wave (BM/WR)x(BN/WC).
//
// LDS image: [rows][32 floats] (128-B rows), 16-B chunk index XOR ((row>>1)&7): the ds_read_b128 of a
// 32x32x2 operand (lane = row, lane>>5 picks the chunk parity) is then conflict-free in every 16-lane
// group, and so is the 8-lanes-per-row ds_write_b128 of the staging pass.
//
// Pipeline: register-staged double buffer (issue global loads of tile t+1, run the 64 MFMAs of tile t,
// then ds_write t+1, one barrier per tile).  A tile's MFMA work is >= 2048 cycles per wave, so HBM/L2
// latency hides behind it at 2 workgroups per CU.
#include "sp_common.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SP_DIAG
// DIAGNOSTIC BUILD ONLY (never the shipped library): per-wave cycle sums of the K-tile segments, read back with
// sp_debug_read().  [block % 4096][wave][8 segments]
__device__ unsigned long long sp_dbg[4096 * 4 * 12];
extern "C" int sp_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sp_dbg), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
extern "C" int sp_debug_clear() {
    static unsigned long long z[4096 * 4 * 12];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sp_dbg), z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#define SP_STAMP_ALWAYS(var)                                                                \
    unsigned long long var;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#ifdef SP_DIAG_LIGHT
#define SP_STAMP(var) unsigned long long var = 0;
#else
#define SP_STAMP(var) SP_STAMP_ALWAYS(var)
#endif
#else
#define SP_STAMP(var)
#endif

namespace {

struct ConvArgs {
    const void* x;       // NHWC activations, fp32 or bf16
    const void* w;       // packed weights, same dtype as x
    const float* scale;
    const float* shift;
    const void* res;     // same dtype/layout as y
    const unsigned char* res_mask;   // bf16 y only: add res where its bit is set (one byte per 8 channels) - a dgrad launch that takes the
                                     // residual share of a block input's gradient as (dy of the block output, its ReLU bit mask)
    void* y;             // NHWC (dtype of x) or NCHW fp32
    int M;  // batch * grid_h * grid_w
    int in_h, in_w, c_in;
    int grid_h, grid_w;
    int c_out, n_pad, k_pad;
    int taps_h, taps_w;
    int stride, stride_x, dy0, dy_step, dx0, dx_step;   // stride: y (and x unless the descriptor says otherwise)
    int out_h, out_w, out_c;
    int oy_mul, oy_add, ox_mul, ox_add;
    int phases_x;  // 1 or 2
    unsigned flags;
    int tiles_m, tiles_n;
    int x_bytes, w_bytes, y_bytes;  // buffer-descriptor extents (w: one phase slab)
    // STATS instantiations (train-mode BatchNorm): per (phase, M tile) partial sums of the stored values, per channel
    float* stats_s;      // [rows][stats_stride] sum
    float* stats_q;      // [rows][stats_stride] sum of squares
    int stats_stride;
    // BSTATS instantiations (dgrad launches): the tensor this launch writes is dy of a BatchNorm+ReLU; with that layer's saved
    // output by (ReLU mask), input bz and statistics, the epilogue also leaves the partial sums of g = dy*(by > 0) (-> stats_s)
    // and g * xhat (-> stats_q): the BN backward reduction without its own pass over dy / y / z
    const void* by;
    const void* bz;
    const float* bmean;
    const float* binvstd;
    int bz_bytes;
    // a SECOND BatchNorm whose dy is the same g (the projection shortcut of a stage's first bottleneck: its output is added to bn3's, so
    // its dy is bn3's g): sum of g * xhat2 -> stats_q2 (sum g is shared); null: none
    const void* bz2;
    const float* bmean2;
    const float* binvstd2;
    float* stats_q2;
    // ph_n > 0: blockIdx.y walks ph_n phases with their OWN tap geometry / weights (the output phases of a stride-2 conv's dgrad, which differ
    // in tap counts: one launch instead of one per phase); the fields above with the same names are then unused
    struct PhaseGeo { const void* w; int taps_h, taps_w, k_pad, dy0, dx0, oy_add, ox_add, pad_; } ph[4];
    int ph_n;
    // grouped convolution (ResNeXt's conv2, nets/pose_resnet_dconv.py:101: groups = 32): the weight matrix is block-diagonal, so an N tile of
    // tile_n = c_in_g output channels only reads the c_in_g input channels of its own groups - K per tap is c_in_g, not c_in, the packed
    // panel of the tile is [tile_n][taps * c_in_g] (zeros where a panel spans several groups), and the A gather adds the tile's channel offset.
    // 0: dense.
    int c_in_g;
    // ABN instantiation (round 5; train-mode forward of a Bottleneck's conv3): the A operand is relu(bn(z)) of the PREVIOUS BatchNorm, formed in
    // the staging pass from z (what `x` points to) and that layer's batch statistics; the N-tile-0 workgroups also store the activation (and
    // its ReLU bit mask) the backward pass and the weight gradient read - the stand-alone BatchNorm + ReLU pass over that tensor disappears
    const float* abn_mean;
    const float* abn_invstd;
    const float* abn_gamma;
    const float* abn_beta;
    void* abn_y;                 // NHWC bf16, layout of x
    unsigned char* abn_mask;     // one byte per 8 channels of abn_y (null: no mask)
};

constexpr int BK = 32;  // floats per K tile (8 chunks of 16 B)

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (the BatchNorm element map of train.hip's bn_fwd_elem, operation by operation with contraction off: the ABN staging pass must give the bits
// of the stand-alone pass it replaces)
__device__ __forceinline__ float abn_elem(float v, float mu, float is, float g, float b) {
#pragma clang fp contract(off)
    return (v - mu) * is * g + b;
}

template <int BM, int BN, int WR, int WC, bool UNIFORM_TAP, bool BF16, bool OUT16, bool STATS, bool DEEP, bool BSTATS, bool ABN>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& p) {
    static_assert(!ABN || (UNIFORM_TAP && BF16 && OUT16 && STATS && !DEEP && !BSTATS), "ABN: the bf16 train-mode forward of a 1x1 conv");
    constexpr int ES = BF16 ? 2 : 4;     // element size of activations / weights
    constexpr int EPC = 16 / ES;         // elements per 16-byte chunk
    constexpr int BKE = 128 / ES;        // elements per K tile (one 128-byte LDS row)
    constexpr int ESO = OUT16 ? 2 : 4;   // element size of the NHWC output / residual (bf16 inputs may write fp32: SP_CONV_OUT_F32)
    static_assert(BF16 || !OUT16, "bf16 output needs bf16 inputs");
    static_assert(WR * WC == 4, "4 waves per workgroup");
    constexpr int WM = BM / WR, WN = BN / WC;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_CH = BM / 32, B_CH = BN / 32;  // 16-B chunks each thread stages per tile
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

#ifdef SP_DIAG
    SP_STAMP_ALWAYS(t_entry)
#endif
#ifdef SP_DIAG
    unsigned long long rt_entry;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_entry)::"memory");
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                    // [2][BM][32]
    float* Bs = smem + 2 * BM * BK;      // [2][BN][32]
    int* rowtab = reinterpret_cast<int*>(smem + 2 * (BM + BN) * BK);  // [BM][4]: in_base, iy0, ix0, out_off

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    // ---- tile id: XCD-aware remap (blocks b and b+8 share an XCD's L2), n fastest inside an XCD chunk ----
    const int nwg = p.tiles_m * p.tiles_n;
    int t;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = t % p.tiles_n, tm = t / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- phase (transposed conv) ----
    const int phase = blockIdx.y;
    const int py = phase / p.phases_x, px = phase % p.phases_x;
    int dy0 = p.dy0 + py, dx0 = p.dx0 + px;
    int oy_add = p.oy_add + py, ox_add = p.ox_add + px;
    int taps_h = p.taps_h, taps_w = p.taps_w, k_pad = p.k_pad, w_bytes = p.w_bytes;
    const char* __restrict__ wp = reinterpret_cast<const char*>(p.w) + (size_t)phase * p.n_pad * p.k_pad * ES;
    if (p.ph_n) {                         // (wave-uniform: scalar loads from the kernel arguments)
        const ConvArgs::PhaseGeo& g = p.ph[phase];
        dy0 = g.dy0; dx0 = g.dx0; oy_add = g.oy_add; ox_add = g.ox_add;
        taps_h = g.taps_h; taps_w = g.taps_w; k_pad = g.k_pad;
        w_bytes = p.n_pad * k_pad * ES;
        wp = reinterpret_cast<const char*>(g.w);
    }

    // ---- per-row table: one decode per row per workgroup ----
    //   x: byte offset of tap (0,0), channel 0 of this output pixel's receptive field (only used through valid taps)
    //   y: UNIFORM_TAP: bit (ty*taps_w + tx) set when that tap lies inside the image (taps <= 32, host-checked)
    //   z: (iy0 << 16) | (ix0 & 0xffff)  (generic path)      w: output element offset of the pixel, -1 = row >= M
    if (tid < BM) {
        const int m = m0 + tid;
        int4 e;
        if (m < p.M) {
            const int gw = p.grid_w, ghw = p.grid_h * gw;
            const int b = m / ghw, rem = m - b * ghw;
            const int gy = rem / gw, gx = rem - gy * gw;
            const int iy0 = gy * p.stride + dy0, ix0 = gx * p.stride_x + dx0;
            e.x = ((b * p.in_h + iy0) * p.in_w + ix0) * p.c_in * ES;
            unsigned msk = 0;
            if (UNIFORM_TAP) {
                for (int ty = 0; ty < taps_h; ++ty)
                    for (int tx = 0; tx < taps_w; ++tx) {
                        const int iy = iy0 + ty * p.dy_step, ix = ix0 + tx * p.dx_step;
                        if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) msk |= 1u << (ty * taps_w + tx);
                    }
            }
            e.y = (int)msk;
            e.z = (iy0 << 16) | (ix0 & 0xffff);
            const int oy = gy * p.oy_mul + oy_add, ox = gx * p.ox_mul + ox_add;
            e.w = (p.flags & SP_CONV_OUT_NCHW) ? (b * p.out_c * p.out_h + oy) * p.out_w + ox
                                               : ((b * p.out_h + oy) * p.out_w + ox) * p.out_c;
        } else {
            e.x = 0; e.y = 0; e.z = (int)0x80008000u; e.w = -1;   // no valid tap; iy0 = ix0 = -32768
        }
        reinterpret_cast<int4*>(rowtab)[tid] = e;
    }
    float* const abn_tab = reinterpret_cast<float*>(rowtab + BM * 4) + 3 * WR * BN;    // [c_in][4]: mean, invstd, gamma, beta (ABN only)
    if constexpr (ABN) {
        for (int c = tid; c < p.c_in; c += 256)
            *reinterpret_cast<f32x4*>(abn_tab + 4 * c) = f32x4{p.abn_mean[c], p.abn_invstd[c], p.abn_gamma[c], p.abn_beta[c]};
    }
    __syncthreads();

    // ---- staging assignment: thread -> (row = tid/8 + 32 i, chunk = tid%8) ----
    // All global traffic goes through buffer descriptors: an out-of-image tap (or a row beyond M) gets the byte offset
    // OOB, which the hardware range check turns into zeros (loads) or drops (stores) - no branches, no selects, and the
    // compiler keeps every load of a tile in flight behind the MFMAs instead of waiting at each exec-mask join.
    constexpr unsigned OOB = 0x80000000u;  // every tensor is < 2 GiB (checked on the host)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wp), (short)0, w_bytes, 0x00020000);
    const int kc = tid & 7;
    const int srow = tid >> 3;
    // Per staged row: tap-(0,0) byte offset (+ this lane's 16-byte chunk) and the tap validity mask.  Per K tile a load
    // then costs ~4 VALU (mask test, select, add) instead of two range checks and a multiply-add chain - the wave issues
    // in order, so every VALU cycle here is a cycle the next MFMA waits.
    int a_off0[A_CH];
    unsigned a_mask[A_CH];
    int a_iy[A_CH], a_ix[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int4 e = reinterpret_cast<const int4*>(rowtab)[srow + 32 * i];
        a_off0[i] = e.x + kc * 16;
        a_mask[i] = (unsigned)e.y;
        a_iy[i] = e.z >> 16;
        a_ix[i] = (int)(short)(e.z & 0xffff);
    }
    const unsigned b_voff = (unsigned)((srow * k_pad + kc * EPC) * ES);
    const unsigned b_soff0 = (unsigned)n0 * k_pad * ES;

    // Staging registers.  fp32: one set (a K tile is ~3,000 cycles of MFMA, enough to cover a load).  bf16: a K tile is only
    // 130-500 cycles; the DEEP instantiation of the 64x64 tile (chosen by the launcher for K >= 12 tiles: the long-K,
    // few-workgroup layers and most of the 32-image train step) keeps PF tiles in flight in a ring of register sets (tile j lives in set j % PF) - the loads of tile kt+PF are issued during tile kt and are
    // only needed (ds_write) at the end of tile kt+PF-1.
    static_assert(!DEEP || BF16, "the register ring is a bf16 feature");
    constexpr int PF = DEEP ? (BM * BN <= 64 * 64 ? 4 : 2) : 1;   // larger bf16 tiles: the ring's registers cost more occupancy than the depth wins (measured)
    u32x4 sa[PF][A_CH], sb[PF][B_CH];
    const int cin_chunks = p.c_in / EPC;

    // tap decode of K tile kt (scalar when UNIFORM_TAP), then the loads as A_CH + B_CH independent pieces that the main
    // loop drops one at a time into the shadows of the MFMAs
    int t_shift = 0, t_ddy = 0, t_ddx = 0, t_coff = 0, t_k0 = 0;
    unsigned t_bit = 0;
    bool t_ok = true;
    auto tile_taps = [&](int kt) {
        const int k0 = kt * BKE;
        t_k0 = k0;
        if (UNIFORM_TAP) {  // c_in % 32 == 0: the whole K tile sits inside one tap (all scalar)
            const int cg = p.c_in_g ? p.c_in_g : p.c_in;       // channels of one tap in K (grouped: the tile's own channel range)
            const int tap = k0 / cg;
            const int ty = tap / taps_w, tx = tap - ty * taps_w;
            t_bit = 1u << tap;
            t_shift = ((ty * p.dy_step * p.in_w + tx * p.dx_step) * p.c_in + (k0 - tap * cg) + (p.c_in_g ? tn * p.c_in_g : 0)) * ES;
        } else {            // small c_in (stem: NHWC4): every 16-B chunk may be a different tap
            const int q = k0 / EPC + kc;
            const int tap = q / cin_chunks;
            t_coff = (q - tap * cin_chunks) * EPC;
            const int ty = tap / taps_w, tx = tap - ty * taps_w;
            t_ok = ty < taps_h;
            t_ddy = ty * p.dy_step; t_ddx = tx * p.dx_step;
        }
    };
    using I0 = std::integral_constant<int, 0>;
    auto load_a = [&](int i, auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        unsigned off;
        if (UNIFORM_TAP) {
            off = (a_mask[i] & t_bit) ? (unsigned)(a_off0[i] + t_shift) : OOB;
        } else {
            const int iy = a_iy[i] + t_ddy, ix = a_ix[i] + t_ddx;
            const bool ok = t_ok && (unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w;
            off = ok ? (unsigned)(a_off0[i] + ((t_ddy * p.in_w + t_ddx) * p.c_in + t_coff) * ES - kc * 16) : OOB;
        }
        sa[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
    };
    auto load_b = [&](int i, auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        sb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(wr_, b_voff + (unsigned)(32 * i * k_pad * ES), b_soff0 + (unsigned)(t_k0 * ES), 0);
    };
    const __amdgpu_buffer_rsrc_t abn_yr = __builtin_amdgcn_make_buffer_rsrc(ABN ? p.abn_y : p.y, (short)0, ABN ? p.x_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t abn_mr = __builtin_amdgcn_make_buffer_rsrc((ABN && p.abn_mask) ? (void*)p.abn_mask : p.y, (short)0,
                                                                            (ABN && p.abn_mask) ? (p.x_bytes >> 4) : 0, 0x00020000);
    auto store_piece = [&](int buf, int o, auto s_tag) {  // o in [0, A_CH + B_CH)
        constexpr int S = decltype(s_tag)::value;
        if (o < A_CH) {
            u32x4 piece = sa[S][o];
            if constexpr (ABN) {
                // (1x1, stride 1, no padding: one tap, K tile t_k0 = channels t_k0 .. t_k0 + 63 of the pixel; a_mask bit 0 = row < M)
                const bool valid = (a_mask[o] & 1u) != 0;
                const bf16x8 zin = __builtin_bit_cast(bf16x8, piece);
                const float* tb = abn_tab + 4 * (t_k0 + kc * 8);
                bf16x8 yo;
                unsigned bits = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(tb + 4 * e);
                    float v = abn_elem((float)zin[e], q[0], q[1], q[2], q[3]);
                    bits |= (v > 0.f ? 1u : 0u) << e;
                    v = v > 0.f ? v : 0.f;
                    yo[e] = (__bf16)(valid ? v : 0.f);
                }
                piece = __builtin_bit_cast(u32x4, yo);
                if (tn == 0) {                     // every A element is staged by exactly one N-tile-0 workgroup: it writes y (and the mask)
                    const unsigned off = valid ? (unsigned)(a_off0[o] + t_k0 * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(piece, abn_yr, off, 0, 0);
                    if (p.abn_mask) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, abn_mr, valid ? (off >> 4) : OOB, 0, 0);
                }
            }
            *reinterpret_cast<u32x4*>(As + buf * BM * BK + swz(srow + 32 * o, kc)) = piece;
        } else *reinterpret_cast<u32x4*>(Bs + buf * BN * BK + swz(srow + 32 * (o - A_CH), kc)) = sb[S][o - A_CH];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int nk = k_pad / BKE;

    // Fragment registers are double-buffered by hand (slot = k-step parity): the ds_reads of k-step j+1 are issued
    // before the MFMAs of k-step j, so LDS latency hides behind ~1000 cycles of matrix work.
    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](const float* a, const float* b, int j, int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const f32x4*>(a + swz(i * 32 + fr, 2 * j + fh));
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[slot][n] = *reinterpret_cast<const f32x4*>(b + swz(n * 32 + fr, 2 * j + fh));
    };
    constexpr int NM = (BF16 ? 1 : 4) * TM * TN;   // MFMAs per k-step (fp32: 4 k-pairs per 16-byte fragment; bf16: one k=16 MFMA)
    constexpr int NOPS = A_CH + B_CH;              // 16-byte staging pieces per thread per K tile
    auto mfma_q = [&](int slot, int q) {           // q-th MFMA of a k-step: k-pair s = q / (TM*TN), tile (i, n)
        const int s = q / (TM * TN), i = (q / TN) % TM, n = q % TN;
        if constexpr (BF16)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[slot][i]), __builtin_bit_cast(bf16x8, fb[slot][n]),
                                                                acc[i][n], 0, 0, 0);
        else
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i][s], fb[slot][n][s], acc[i][n], 0, 0, 0);
    };
#define SP_SB() __builtin_amdgcn_sched_barrier(0)

    // prologue: tile 0 -> LDS, first fragments -> registers (bf16: tiles 1 .. PF-1 are already requested behind tile 0)
    tile_taps(0);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) load_a(i, I0{});
#pragma unroll
    for (int i = 0; i < B_CH; ++i) load_b(i, I0{});
    if constexpr (PF > 1) {
        auto request = [&](auto j_tag) {
            constexpr int J = decltype(j_tag)::value;
            if (J < nk) {
                tile_taps(J);
#pragma unroll
                for (int i = 0; i < A_CH; ++i) load_a(i, j_tag);
#pragma unroll
                for (int i = 0; i < B_CH; ++i) load_b(i, j_tag);
            }
        };
        request(std::integral_constant<int, 1>{});
        if constexpr (PF == 4) { request(std::integral_constant<int, 2>{}); request(std::integral_constant<int, 3>{}); }
    }
#pragma unroll
    for (int o = 0; o < NOPS; ++o) store_piece(0, o, I0{});
    __syncthreads();
    read_frags(As + (wr * WM) * BK, Bs + (wc * WN) * BK, 0, 0);

    // One K tile = 4 k-steps of NM MFMAs.  A wave issues in order and a buffer_load / ds_write_b128 occupies the issue
    // port for ~50-90 cycles (measured with s_memtime stamps), about one MFMA's 64 cycles in the matrix pipe.  So every
    // slow instruction is placed in the shadow of its own MFMA, never two in a row:
    //   step 0: the NOPS global loads of tile kt+1, one after every (NM/NOPS)-th MFMA
    //   step 1, 2: fragment prefetch only
    //   step 3: first half: the NOPS ds_writes of tile kt+1 (other LDS buffer), one per MFMA; then lgkmcnt(0) +
    //           s_barrier; the first fragments of tile kt+1 are requested and the second half of step 3's MFMAs covers
    //           their LDS latency.  All reads of the current buffer were issued (into registers) before step 2's MFMAs,
    //           so the barrier also frees the current buffer for tile kt+2.
#ifdef SP_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto k_tile = [&](int kt, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int cur = kt & 1;
        const float* a = As + cur * BM * BK + (wr * WM) * BK;
        const float* b = Bs + cur * BN * BK + (wc * WN) * BK;
        const float* an = As + (cur ^ 1) * BM * BK + (wr * WM) * BK;
        const float* bn = Bs + (cur ^ 1) * BN * BK + (wc * WN) * BK;
        // ---- step 0 (slot 0) ----
        SP_STAMP(s0)
        read_frags(a, b, 1, 1);
        if (MORE) tile_taps(kt + 1);
        SP_SB();
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            mfma_q(0, q);
            if (MORE) {
#pragma unroll
                for (int o = (q * NOPS) / NM; o < ((q + 1) * NOPS) / NM; ++o) {
                    if (o < A_CH) load_a(o, I0{}); else load_b(o - A_CH, I0{});
                }
            }
            SP_SB();
        }
        SP_STAMP(s1)
        // ---- step 1 (slot 1) ----
        read_frags(a, b, 2, 0);
#pragma unroll
        for (int q = 0; q < NM; ++q) mfma_q(1, q);
        SP_SB();
        SP_STAMP(s2)
        // ---- step 2 (slot 0) ----
        read_frags(a, b, 3, 1);
#pragma unroll
        for (int q = 0; q < NM; ++q) mfma_q(0, q);
        SP_SB();
        SP_STAMP(s3)
        SP_STAMP(s3w)
        // ---- step 3 (slot 1), first half + the ds_writes ----
        constexpr int H3 = NM / 2;   // MFMAs of step 3 issued before the barrier (0 for the single-MFMA bf16 64x64 tile)
        if constexpr (H3 == 0) {
            if (MORE) {
#pragma unroll
                for (int o = 0; o < NOPS; ++o) store_piece(cur ^ 1, o, I0{});
            }
        } else {
#pragma unroll
            for (int q = 0; q < H3; ++q) {
                mfma_q(1, q);
                if (MORE) {
#pragma unroll
                    for (int o = (q * NOPS) / H3; o < ((q + 1) * NOPS) / H3; ++o) store_piece(cur ^ 1, o, I0{});
                }
                SP_SB();
            }
        }
        SP_STAMP(s4)
        if (MORE) {
            __syncthreads();
        }
        SP_STAMP(s5)
        if (MORE) {
            read_frags(an, bn, 0, 0);
            SP_SB();
        }
#pragma unroll
        for (int q = NM / 2; q < NM; ++q) mfma_q(1, q);
        SP_SB();
#ifdef SP_DIAG
        SP_STAMP(s6)
        dg[0] += s1 - s0; dg[1] += s2 - s1; dg[2] += s3 - s2; dg[3] += s3w - s3; dg[4] += s4 - s3w; dg[5] += s5 - s4; dg[6] += s6 - s5; dg[7] += 1;
#endif
    };
#ifdef SP_DIAG
    SP_STAMP_ALWAYS(t_loop0)
#endif
    if constexpr (PF == 1) {
        for (int kt = 0; kt + 1 < nk; ++kt) k_tile(kt, std::true_type{});
        k_tile(nk - 1, std::false_type{});
    } else {
        // bf16: the same tile body with a ring of PF register sets.  S = kt % PF is a compile-time constant (the loop advances PF
        // tiles per trip), LOAD = tile kt+PF exists, STORE = tile kt+1 exists.
        auto k_tile_pf = [&](int kt, auto slot_tag, auto load_tag, auto store_tag) {
            constexpr int S = decltype(slot_tag)::value, SN = (S + 1) % PF;
            constexpr bool LOAD = decltype(load_tag)::value, STORE = decltype(store_tag)::value;
            const int cur = kt & 1;
            const float* a = As + cur * BM * BK + (wr * WM) * BK;
            const float* b = Bs + cur * BN * BK + (wc * WN) * BK;
            const float* an = As + (cur ^ 1) * BM * BK + (wr * WM) * BK;
            const float* bn = Bs + (cur ^ 1) * BN * BK + (wc * WN) * BK;
            // step 0: MFMAs of k-step 0 with the global loads of tile kt+PF dropped between them
            read_frags(a, b, 1, 1);
            if (LOAD) tile_taps(kt + PF);
            SP_SB();
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                mfma_q(0, q);
                if (LOAD) {
#pragma unroll
                    for (int o = (q * NOPS) / NM; o < ((q + 1) * NOPS) / NM; ++o) {
                        if (o < A_CH) load_a(o, slot_tag); else load_b(o - A_CH, slot_tag);
                    }
                }
                SP_SB();
            }
            read_frags(a, b, 2, 0);
#pragma unroll
            for (int q = 0; q < NM; ++q) mfma_q(1, q);
            SP_SB();
            read_frags(a, b, 3, 1);
#pragma unroll
            for (int q = 0; q < NM; ++q) mfma_q(0, q);
            SP_SB();
            // step 3: ds_writes of tile kt+1 (requested PF-1 tiles ago), barrier, first fragments of the next tile
            constexpr int H3 = NM / 2;
            if constexpr (H3 == 0) {
                if (STORE) {
#pragma unroll
                    for (int o = 0; o < NOPS; ++o) store_piece(cur ^ 1, o, std::integral_constant<int, SN>{});
                }
            } else {
#pragma unroll
                for (int q = 0; q < H3; ++q) {
                    mfma_q(1, q);
                    if (STORE) {
#pragma unroll
                        for (int o = (q * NOPS) / H3; o < ((q + 1) * NOPS) / H3; ++o) store_piece(cur ^ 1, o, std::integral_constant<int, SN>{});
                    }
                    SP_SB();
                }
            }
            if (STORE) {
                __syncthreads();
                read_frags(an, bn, 0, 0);
                SP_SB();
            }
#pragma unroll
            for (int q = NM / 2; q < NM; ++q) mfma_q(1, q);
            SP_SB();
        };
        // nk is a multiple of PF (the launcher only picks DEEP then): whole trips of PF tiles, no data-dependent tail - the
        // register allocator needs ~60 VGPRs more as soon as the ring's sets are conditionally defined.
        int kt = 0;
        for (; kt + PF < nk; kt += PF) {                   // every tile of the trip requests tile kt+PF and stores tile kt+1
            k_tile_pf(kt, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
            k_tile_pf(kt + 1, std::integral_constant<int, 1 % PF>{}, std::true_type{}, std::true_type{});
            if constexpr (PF == 4) {
                k_tile_pf(kt + 2, std::integral_constant<int, 2 % PF>{}, std::true_type{}, std::true_type{});
                k_tile_pf(kt + 3, std::integral_constant<int, 3 % PF>{}, std::true_type{}, std::true_type{});
            }
        }
        // last trip: nothing left to request; the very last tile has nothing to store
        if constexpr (PF == 2) {
            k_tile_pf(kt, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
            k_tile_pf(kt + 1, std::integral_constant<int, 1 % PF>{}, std::false_type{}, std::false_type{});
        } else {
            k_tile_pf(kt, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
            k_tile_pf(kt + 1, std::integral_constant<int, 1 % PF>{}, std::false_type{}, std::true_type{});
            k_tile_pf(kt + 2, std::integral_constant<int, 2 % PF>{}, std::false_type{}, std::true_type{});
            k_tile_pf(kt + 3, std::integral_constant<int, 3 % PF>{}, std::false_type{}, std::false_type{});
        }
    }
#ifdef SP_DIAG
    SP_STAMP_ALWAYS(t_loop1)
#endif
    __syncthreads();  // every wave is done with LDS before the epilogue touches anything else
#undef SP_SB

    // ---- epilogue: y = act(acc * scale + shift (+ residual)); C/D map: col = lane&31, row = (r&3)+8(r>>2)+4(lane>>5)
    // Stores on this chip are ISSUE-bound (~70 cycles per wave-instruction per CU whatever its width), so the epilogue
    // is built around 16-byte accesses:
    //   NHWC (c_out % 4 == 0): the wave's WM x WN accumulator tile is transposed through its private slice of the (now
    //       dead) A/B staging LDS, so that a lane owns 4 consecutive channels of one pixel -> dwordx4 residual loads and
    //       stores, whole 128/256-byte row segments per instruction, 4x fewer instructions than the natural layout;
    //   NCHW (heat maps): a lane already owns 4 consecutive pixels of one channel (accumulator rows) -> dwordx4 directly;
    //   anything else: scalar fallback.
    const bool nchw = p.flags & SP_CONV_OUT_NCHW;
    const bool pshuf = p.flags & SP_CONV_PIXEL_SHUFFLE;
    const bool relu = p.flags & SP_CONV_RELU;
    const int hw_out = p.out_h * p.out_w;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), (short)0, p.y_bytes, 0x00020000);
    if (!nchw && (p.c_out % (16 / ESO)) == 0) {
        static_assert(BM * BN <= 2 * (BM + BN) * BK, "transpose area must fit in the staging buffers");
        float* tr = smem + wave * (WM * WN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tr[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * WN + n * 32 + fr] = acc[i][n][r];
        constexpr int CPL = 16 / ESO;    // channels per lane = one 16-byte store (4 fp32 / 8 bf16)
        constexpr int CPR = WN / CPL;    // 16-byte output chunks per tile row
        constexpr int RPI = 64 / CPR;    // tile rows covered by one wave-instruction
        constexpr int NIT = WM / RPI;
        const int chunk = lane % CPR, rsub = lane / CPR;
        const int col = n0 + wc * WN + chunk * CPL;
        const bool col_ok = col < p.c_out;
        float sc[CPL], sh[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (col_ok) {
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
                if (p.scale) { const f32x4 t = *reinterpret_cast<const f32x4*>(p.scale + col + 4 * e4); sc[4 * e4] = t[0]; sc[4 * e4 + 1] = t[1]; sc[4 * e4 + 2] = t[2]; sc[4 * e4 + 3] = t[3]; }
                if (p.shift) { const f32x4 t = *reinterpret_cast<const f32x4*>(p.shift + col + 4 * e4); sh[4 * e4] = t[0]; sh[4 * e4 + 1] = t[1]; sh[4 * e4 + 2] = t[2]; sh[4 * e4 + 3] = t[3]; }
            }
        }
        int col_off = col;
        if (pshuf) {
            const int sub = col / p.out_c, c = col - sub * p.out_c;  // packed column order: sub-pixel major
            col_off = ((sub >> 1) * p.out_w + (sub & 1)) * p.out_c + c;
        }
        unsigned off[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int ro = rowtab[(wr * WM + it * RPI + rsub) * 4 + 3];
            off[it] = (col_ok && ro >= 0) ? (unsigned)((ro + col_off) * ESO) : OOB;
        }
        static_assert(!(STATS && BSTATS) && (!BSTATS || !OUT16 || BF16), "BSTATS: fp32 gradients, or bf16 gradients beside bf16 activations");
        float st_s[(STATS || BSTATS) ? CPL : 1], st_q[(STATS || BSTATS) ? CPL : 1];   // column sums over this lane's rows
        float b_mu[BSTATS ? CPL : 1], b_is[BSTATS ? CPL : 1];
        float st_q2[BSTATS ? CPL : 1], b_mu2[BSTATS ? CPL : 1], b_is2[BSTATS ? CPL : 1];
        if constexpr (STATS || BSTATS) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) { st_s[e] = 0.f; st_q[e] = 0.f; }
        }
        if constexpr (BSTATS) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) { b_mu[e] = 0.f; b_is[e] = 0.f; st_q2[e] = 0.f; b_mu2[e] = 0.f; b_is2[e] = 0.f; }
            if (col_ok) {
#pragma unroll
                for (int e4 = 0; e4 < CPL / 4; ++e4) {
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(p.bmean + col + 4 * e4), i4 = *reinterpret_cast<const f32x4*>(p.binvstd + col + 4 * e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { b_mu[4 * e4 + e] = m4[e]; b_is[4 * e4 + e] = i4[e]; }
                    if (p.bz2) {
                        const f32x4 m2 = *reinterpret_cast<const f32x4*>(p.bmean2 + col + 4 * e4), i2 = *reinterpret_cast<const f32x4*>(p.binvstd2 + col + 4 * e4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { b_mu2[4 * e4 + e] = m2[e]; b_is2[4 * e4 + e] = i2[e]; }
                    }
                }
            }
        }
        const bool by_mask = BSTATS && OUT16 && (p.flags & SP_CONV_BN_Y_MASK);     // the ReLU source is a bit mask: one byte per 8 channels
        const __amdgpu_buffer_rsrc_t byr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BSTATS ? p.by : p.y), (short)0,
                                                                             BSTATS ? (by_mask ? p.bz_bytes >> 4 : p.bz_bytes) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t bzr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BSTATS ? p.bz : p.y), (short)0, BSTATS ? p.bz_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t bz2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>((BSTATS && p.bz2) ? p.bz2 : p.y), (short)0, (BSTATS && p.bz2) ? p.bz_bytes : 0, 0x00020000);
        u32x4 rv[NIT];
        if (p.res) {  // every residual load of the tile in flight before the first use
#pragma unroll
            for (int it = 0; it < NIT; ++it) rv[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, off[it], 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float v[CPL];
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; ++e4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(tr + (it * RPI + rsub) * WN + chunk * CPL + 4 * e4);
                v[4 * e4] = t[0]; v[4 * e4 + 1] = t[1]; v[4 * e4 + 2] = t[2]; v[4 * e4 + 3] = t[3];
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (p.res) {
                if constexpr (OUT16) {
                    const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[it]);
                    if (p.res_mask) {
                        const __amdgpu_buffer_rsrc_t rmr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.res_mask), (short)0, p.y_bytes >> 4, 0x00020000);
                        const unsigned m = __builtin_amdgcn_raw_buffer_load_b8(rmr, off[it] == OOB ? OOB : off[it] >> 4, 0, 0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += ((m >> e) & 1u) ? (float)r8[e] : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                    }
                } else {
                    const f32x4 r4 = __builtin_bit_cast(f32x4, rv[it]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            u32x4 o;
            if constexpr (OUT16) {
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
                o = __builtin_bit_cast(u32x4, o8);
                if constexpr (STATS) {                    // BatchNorm sees the ROUNDED tensor (as torch does on a bf16 activation)
                    if (off[it] != OOB) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float z = (float)o8[e]; st_s[e] += z; st_q[e] += z * z; }
                    }
                }
                if constexpr (BSTATS) {                   // bf16 gradients: the sums are those of the ROUNDED dy (what the BatchNorm backward pass reads)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)o8[e];
                }
            } else {
                const f32x4 o4 = {v[0], v[1], v[2], v[3]};
                o = __builtin_bit_cast(u32x4, o4);
                if constexpr (STATS) {
                    if (off[it] != OOB) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { st_s[e] += v[e]; st_q[e] += v[e] * v[e]; }
                    }
                }
            }
            if constexpr (BSTATS) {           // v = dy (complete: the residual input carried the other contributions)
                float yy[CPL], zz[CPL], z2[CPL];
#pragma unroll
                for (int e = 0; e < CPL; ++e) z2[e] = 0.f;
                if constexpr (OUT16) {        // bf16 activations and gradients: same byte offsets, 16 bytes = 8 channels each
                    const bf16x8 z8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bzr, off[it], 0, 0));
                    if (by_mask) {
                        const unsigned m = __builtin_amdgcn_raw_buffer_load_b8(byr, off[it] == OOB ? OOB : off[it] >> 4, 0, 0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) yy[e] = ((m >> e) & 1u) ? 1.f : 0.f;
                    } else {
                        const bf16x8 y8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(byr, off[it], 0, 0));
#pragma unroll
                        for (int e = 0; e < 8; ++e) yy[e] = (float)y8[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) zz[e] = (float)z8[e];
                    if (p.bz2) {
                        const bf16x8 q8 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bz2r, off[it], 0, 0));
#pragma unroll
                        for (int e = 0; e < 8; ++e) z2[e] = (float)q8[e];
                    }
                } else if constexpr (BF16) {  // bf16 activations, fp32 gradients: 8 bytes per 4 channels at half the fp32 byte offset
                    const unsigned ho = off[it] == OOB ? OOB : off[it] >> 1;
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
                    const bf16x4_ y4 = __builtin_bit_cast(bf16x4_, __builtin_amdgcn_raw_buffer_load_b64(byr, ho, 0, 0));
                    const bf16x4_ z4 = __builtin_bit_cast(bf16x4_, __builtin_amdgcn_raw_buffer_load_b64(bzr, ho, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { yy[e] = (float)y4[e]; zz[e] = (float)z4[e]; }
                    if (p.bz2) {
                        const bf16x4_ q4 = __builtin_bit_cast(bf16x4_, __builtin_amdgcn_raw_buffer_load_b64(bz2r, ho, 0, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) z2[e] = (float)q4[e];
                    }
                } else {
                    const f32x4 y4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(byr, off[it], 0, 0));
                    const f32x4 z4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bzr, off[it], 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { yy[e] = y4[e]; zz[e] = z4[e]; }
                    if (p.bz2) {
                        const f32x4 q4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bz2r, off[it], 0, 0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) z2[e] = q4[e];
                    }
                }
                if (off[it] != OOB) {
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        const float g = yy[e] > 0.f ? v[e] : 0.f;
                        st_s[e] += g;
                        st_q[e] += g * ((zz[e] - b_mu[e]) * b_is[e]);
                        st_q2[e] += g * ((z2[e] - b_mu2[e]) * b_is2[e]);
                    }
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, yr, off[it], 0, 0);
        }
        if constexpr (STATS || BSTATS) {
            // lanes chunk, chunk + CPR, ... hold the same channels for different rows: fixed butterfly, then (below) one partial row per
            // (phase, M tile); the fold adds the rows in a fixed order -> deterministic statistics
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) { st_s[e] += __shfl_xor(st_s[e], o, 64); st_q[e] += __shfl_xor(st_q[e], o, 64); }
                if constexpr (BSTATS) {
                    if (p.bz2) {
#pragma unroll
                        for (int e = 0; e < CPL; ++e) st_q2[e] += __shfl_xor(st_q2[e], o, 64);
                    }
                }
            }
            // the WR wave rows of the tile meet in LDS (a slice behind the row table, reserved for these instantiations) and are added in
            // wave-row order: ONE partial row per (phase, M tile) - the fold that follows (a launch of its own, or the prologue of the
            // consuming BatchNorm pass: train.hip) reads WR times fewer rows
            float* sred = reinterpret_cast<float*>(rowtab + BM * 4);          // [3][WR][BN]
            constexpr int NARR = BSTATS ? 3 : 2;
            if (rsub == 0) {
                const int cl = wc * WN + chunk * CPL;
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    sred[(0 * WR + wr) * BN + cl + e] = st_s[e];
                    sred[(1 * WR + wr) * BN + cl + e] = st_q[e];
                    if constexpr (BSTATS) sred[(2 * WR + wr) * BN + cl + e] = st_q2[e];
                }
            }
            __syncthreads();
            for (int i = tid; i < NARR * BN; i += 256) {
                const int arr = i / BN, cl = i - arr * BN;
                const int c = n0 + cl;
                if (c >= p.c_out) continue;
                if (BSTATS && arr == 2 && !p.bz2) continue;
                float v = sred[(arr * WR) * BN + cl];
#pragma unroll
                for (int w = 1; w < WR; ++w) v += sred[(arr * WR + w) * BN + cl];
                float* dst = arr == 0 ? p.stats_s : (arr == 1 ? p.stats_q : p.stats_q2);
                dst[(size_t)(phase * p.tiles_m + tm) * p.stats_stride + c] = v;
            }
        }
    } else {
        const bool vec4 = nchw && (hw_out & 3) == 0 && !p.res;  // rows 4g..4g+3 = 4 consecutive pixels of one image
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + wc * WN + n * 32 + fr;
            const bool col_ok = col < p.c_out;
            float sc = 1.f, sh = 0.f;
            if (col_ok) {
                if (p.scale) sc = p.scale[col];
                if (p.shift) sh = p.shift[col];
            }
            int col_off;
            if (nchw) col_off = col * hw_out;
            else if (pshuf) {
                const int sub = col / p.out_c, c = col - sub * p.out_c;
                col_off = ((sub >> 1) * p.out_w + (sub & 1)) * p.out_c + c;
            } else col_off = col;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = wr * WM + i * 32 + 8 * g + 4 * fh;
                    if (vec4) {
                        const int ro = rowtab[row * 4 + 3];
                        const unsigned o = (col_ok && ro >= 0) ? (unsigned)((ro + col_off) * 4) : OOB;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[i][n][4 * g + e] * sc + sh;
                            if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, o, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ro = rowtab[(row + e) * 4 + 3];
                            const unsigned o = (col_ok && ro >= 0) ? (unsigned)((ro + col_off) * 4) : OOB;
                            float v = acc[i][n][4 * g + e] * sc + sh;
                            if (p.res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, o, 0, 0));
                            if (relu) v = v > 0.f ? v : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, o, 0, 0);
                        }
                    }
                }
            }
        }
    }
#ifdef SP_DIAG
    SP_STAMP_ALWAYS(t_issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SP_STAMP_ALWAYS(t_end)
    unsigned long long rt_end;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
    if (lane == 0 && blockIdx.y == 0) {
        unsigned long long* o = sp_dbg + ((blockIdx.x & 4095) * 4 + wave) * 12;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = dg[q];
        o[8] = t_loop0 - t_entry; o[9] = t_loop1 - t_loop0; o[10] = t_end - t_loop1; o[11] = t_end - t_issued; o[7] = (rt_end - rt_entry) | (rt_entry << 32);
    }
#endif
}

template <int BM, int BN, int WR, int WC, bool UNIFORM_TAP, bool BF16, bool OUT16, bool STATS = false, bool DEEP = false, bool BSTATS = false>
__global__ __launch_bounds__(256, (BM * BN > 128 * 128) ? 1 : 2) void conv_igemm_kernel(const ConvArgs p) {
    conv_igemm_body<BM, BN, WR, WC, UNIFORM_TAP, BF16, OUT16, STATS, DEEP, BSTATS, false>(p);
}

// the same body with the ABN staging pass (a kernel of its own name: the instantiation names of conv_igemm_kernel stay what the profiles key on)
template <int BM, int BN, int WR, int WC>
__global__ __launch_bounds__(256, (BM * BN > 128 * 128) ? 1 : 2) void conv_igemm_abn_kernel(const ConvArgs p) {
    conv_igemm_body<BM, BN, WR, WC, true, true, true, true, false, false, true>(p);
}

template <int BM, int BN, int WR, int WC, bool BF16, bool OUT16, bool STATS, bool DEEP = false, bool BSTATS = false>
int launch_t(const ConvArgs& a, int phases, bool uniform, hipStream_t stream) {
    if constexpr (!DEEP && BF16) {
        // deep-prefetch variant once the K loop is long enough to pay for its registers (64x64 tile: +17 % at K = 72 tiles, -10 %
        // at 4; thresholds 12 / 32 K tiles measured on HRNet-W32 and the ResNets); whole trips only: the K-tile count must be a
        // multiple of the ring depth
        constexpr int pf = BM * BN <= 64 * 64 ? 4 : 2;
        int nk = a.k_pad / 64;
        bool whole = nk % pf == 0;
        if (a.ph_n) {                                  // per-phase K: every phase in whole trips, the longest one long enough
            nk = 0;
            whole = true;
            for (int i = 0; i < a.ph_n; ++i) {
                const int n = a.ph[i].k_pad / 64;
                whole = whole && n % pf == 0;
                nk = n > nk ? n : nk;
            }
        }
        if (whole && nk >= (BM * BN <= 64 * 64 ? 12 : 32))
            return launch_t<BM, BN, WR, WC, BF16, OUT16, STATS, true, BSTATS>(a, phases, uniform, stream);
    }
    if (sp_name_query_active()) {
        auto t = [](bool v) { return v ? "true" : "false"; };
        sp_name_query_set("conv_igemm_kernel<%d, %d, %d, %d, %s, %s, %s, %s, %s, %s>", BM, BN, WR, WC, t(uniform), t(BF16), t(OUT16), t(STATS), t(DEEP), t(BSTATS));
        return SP_OK;
    }
    ConvArgs p = a;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = a.n_pad / BN;
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float) + (size_t)BM * 4 * sizeof(int) +
                       ((STATS || BSTATS) ? (size_t)3 * WR * BN * sizeof(float) : 0);     // + the wave rows' column sums (statistics epilogue)
    dim3 grid(p.tiles_m * p.tiles_n, phases, 1), block(256, 1, 1);
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel instantiation AND per device (the attribute belongs to the
    // function on the device that is current: a process driving several GPUs must not inherit device 0's opt-in)
    static bool opted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        const hipError_t eu = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WR, WC, true, BF16, OUT16, STATS, DEEP, BSTATS>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const hipError_t ec = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WR, WC, false, BF16, OUT16, STATS, DEEP, BSTATS>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (eu != hipSuccess || ec != hipSuccess) {
            sp_set_error("conv_igemm: hipFuncSetAttribute(max dynamic LDS = %zu) failed on device %d", lds, dev);
            return SP_ELAUNCH;
        }
        opted[dev] = true;       // (a benign race: two threads may both set the same attribute to the same value)
    }
    if (uniform)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WR, WC, true, BF16, OUT16, STATS, DEEP, BSTATS>), grid, block, lds, stream, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WR, WC, false, BF16, OUT16, STATS, DEEP, BSTATS>), grid, block, lds, stream, p);
    return sp_check_launch("conv_igemm_kernel");
}

template <int BM, int BN, int WR, int WC>
int launch_abn(const ConvArgs& a, hipStream_t stream) {
    if (sp_name_query_active()) {
        sp_name_query_set("conv_igemm_abn_kernel<%d, %d, %d, %d>", BM, BN, WR, WC);
        return SP_OK;
    }
    ConvArgs p = a;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = a.n_pad / BN;
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float) + (size_t)BM * 4 * sizeof(int) + (size_t)3 * WR * BN * sizeof(float) +
                       (size_t)a.c_in * 4 * sizeof(float);                                   // + the [c_in][4] BatchNorm table
    static bool opted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        const size_t lds_max = (size_t)2 * (BM + BN) * BK * sizeof(float) + (size_t)BM * 4 * sizeof(int) + (size_t)3 * WR * BN * sizeof(float) + 512 * 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_abn_kernel<BM, BN, WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_max) != hipSuccess) {
            sp_set_error("conv_igemm_abn: hipFuncSetAttribute(max dynamic LDS = %zu) failed on device %d", lds_max, dev);
            return SP_ELAUNCH;
        }
        opted[dev] = true;
    }
    hipLaunchKernelGGL((conv_igemm_abn_kernel<BM, BN, WR, WC>), dim3(p.tiles_m * p.tiles_n, 1, 1), dim3(256, 1, 1), lds, stream, p);
    return sp_check_launch("conv_igemm_abn_kernel");
}

template <int BM, int BN, int WR, int WC>
int launch(const ConvArgs& a, int phases, bool uniform, hipStream_t stream) {
    if (a.abn_mean) return launch_abn<BM, BN, WR, WC>(a, stream);
    if (a.bz) {                                        // dgrad launch that also reduces the BN backward sums of the tensor it writes
        if ((a.flags & SP_CONV_BF16) && !(a.flags & SP_CONV_OUT_F32)) return launch_t<BM, BN, WR, WC, true, true, false, false, true>(a, phases, uniform, stream);
        if (a.flags & SP_CONV_BF16) return launch_t<BM, BN, WR, WC, true, false, false, false, true>(a, phases, uniform, stream);
        return launch_t<BM, BN, WR, WC, false, false, false, false, true>(a, phases, uniform, stream);
    }
    if (a.stats_s) {                                   // train-mode forward: plain NHWC store of the conv's own dtype
        if (a.flags & SP_CONV_BF16) return launch_t<BM, BN, WR, WC, true, true, true>(a, phases, uniform, stream);
        return launch_t<BM, BN, WR, WC, false, false, true>(a, phases, uniform, stream);
    }
    if (a.flags & SP_CONV_BF16) {
        if (a.flags & SP_CONV_OUT_F32) return launch_t<BM, BN, WR, WC, true, false, false>(a, phases, uniform, stream);
        return launch_t<BM, BN, WR, WC, true, true, false>(a, phases, uniform, stream);
    }
    return launch_t<BM, BN, WR, WC, false, false, false>(a, phases, uniform, stream);
}

}  // namespace

extern "C" int sp_conv2d_default_tile(const sp_conv_desc* d, int* tile_m, int* tile_n);
int sp_conv_ring_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                        const void* residual, void* y, void* stream);   // conv_ring.hip
int sp_conv_pw_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                      const void* residual, void* y, void* stream);     // conv_pw.hip

static int tile_rows_per_block(int, int) { return 1; }   // partial rows per (phase, M tile): the wave rows are added inside the launch

struct BnBwdSrc { const void* y; const void* z; const float* mean; const float* invstd; const void* z2; const float* mean2; const float* invstd2; float* q2;
                  const void* res_mask = nullptr; };

struct PhaseSet { const sp_conv_desc* descs; const void* const* w; int n; };
struct AbnSrc { const float* mean; const float* invstd; const float* gamma; const float* beta; void* y; unsigned char* mask; };

static int conv_fwd_impl(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                         const void* residual, void* y, float* stats_s, float* stats_q, int stats_rows_capacity, void* stream,
                         const BnBwdSrc* bsrc = nullptr, const PhaseSet* phs = nullptr, const AbnSrc* abn = nullptr) {
    SP_REQUIRE(d && x && w_packed && y, "sp_conv2d_fwd: null pointer");
    SP_REQUIRE(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->grid_h > 0 && d->grid_w > 0 && d->c_out > 0,
               "sp_conv2d_fwd: non-positive dimension");
    const bool bf16 = d->flags & SP_CONV_BF16;
    const int es = bf16 ? 2 : 4, epc = 16 / es, bke = 128 / es;
    SP_REQUIRE(d->c_in > 0 && d->c_in % epc == 0, "sp_conv2d_fwd: c_in=%d must be a positive multiple of %d", d->c_in, epc);
    SP_REQUIRE(d->taps_h > 0 && d->taps_w > 0 && d->stride > 0 && d->stride_x >= 0, "sp_conv2d_fwd: bad taps/stride");
    const int cg = d->c_in_group > 0 ? d->c_in_group : d->c_in;      // K channels per tap (grouped: one N tile's channel range)
    SP_REQUIRE(d->k_pad % bke == 0 && d->k_pad >= d->taps_h * d->taps_w * cg,
               "sp_conv2d_fwd: k_pad=%d must be a multiple of %d and >= taps*c_in=%d", d->k_pad, bke,
               d->taps_h * d->taps_w * cg);
    if (d->c_in_group > 0) {
        SP_REQUIRE(d->c_in_group % bke == 0 && d->c_in % d->c_in_group == 0 && d->c_out == d->c_in && d->n_pad == d->c_out &&
                       d->tile_n == d->c_in_group && d->kernel == SP_CONV_KERNEL_IGEMM && d->phases_y == 1 && d->phases_x == 1 && !bsrc && !stats_s && !phs &&
                       !(d->flags & SP_CONV_PIXEL_SHUFFLE) && d->taps_h * d->taps_w <= 32,
                   "sp_conv2d_fwd: grouped launch needs c_in_group = tile_n (a multiple of %d dividing c_in), c_out == c_in == n_pad, the implicit-GEMM "
                   "kernel, one phase, no statistics epilogue (c_in_group %d, tile_n %d, c_in %d, c_out %d)", bke, d->c_in_group, d->tile_n, d->c_in, d->c_out);
    }
    SP_REQUIRE(d->n_pad % 32 == 0 && d->n_pad >= d->c_out, "sp_conv2d_fwd: n_pad=%d must be a multiple of 32 >= c_out=%d",
               d->n_pad, d->c_out);
    const bool uniform = (cg % bke == 0) && d->taps_h * d->taps_w <= 32;  // tap-validity bit mask is 32 bits wide
    if (uniform) SP_REQUIRE(d->k_pad == d->taps_h * d->taps_w * cg, "sp_conv2d_fwd: k_pad must equal taps*c_in when c_in fills whole K tiles");
    SP_REQUIRE((d->phases_y == 1 || d->phases_y == 2) && (d->phases_x == 1 || d->phases_x == 2), "sp_conv2d_fwd: phases must be 1 or 2");
    const unsigned known = SP_CONV_RELU | SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_BF16 | SP_CONV_OUT_F32 | SP_CONV_BN_Y_MASK;
    const bool out16 = bf16 && !(d->flags & SP_CONV_OUT_F32);
    SP_REQUIRE(!(d->flags & SP_CONV_BN_Y_MASK) || (bsrc && out16) || sp_name_query_active(),
               "sp_conv2d_fwd: SP_CONV_BN_Y_MASK belongs to a BSTATS dgrad launch with bf16 activations and gradients");
    if (out16 && !(d->flags & SP_CONV_OUT_NCHW)) SP_REQUIRE(d->c_out % 8 == 0, "sp_conv2d_fwd: bf16 NHWC output needs c_out %% 8 == 0 (got %d)", d->c_out);
    SP_REQUIRE((d->flags & ~known) == 0, "sp_conv2d_fwd: unknown flag bits 0x%x", d->flags);
    SP_REQUIRE(!((d->flags & SP_CONV_OUT_NCHW) && (d->flags & SP_CONV_PIXEL_SHUFFLE)), "sp_conv2d_fwd: NCHW output and pixel shuffle are exclusive");
    SP_REQUIRE(!((d->flags & SP_CONV_OUT_NCHW) && residual), "sp_conv2d_fwd: residual needs NHWC output");
    // output footprint of the launch must lie inside y
    int max_oy = (d->grid_h - 1) * d->oy_mul + d->oy_add + (d->phases_y - 1);
    int max_ox = (d->grid_w - 1) * d->ox_mul + d->ox_add + (d->phases_x - 1);
    int min_oy = d->oy_add, min_ox = d->ox_add;
    if (d->flags & SP_CONV_PIXEL_SHUFFLE) {
        SP_REQUIRE(d->c_out % 4 == 0 && d->out_c * 4 == d->c_out && d->oy_mul == 2 && d->ox_mul == 2 && d->n_pad == d->c_out,
                   "sp_conv2d_fwd: pixel shuffle needs out_c == c_out/4 == n_pad/4, oy_mul == ox_mul == 2");
        max_oy += 1; max_ox += 1;
    } else {
        SP_REQUIRE(d->out_c == d->c_out, "sp_conv2d_fwd: out_c=%d != c_out=%d", d->out_c, d->c_out);
    }
    SP_REQUIRE(d->oy_mul > 0 && d->ox_mul > 0 && min_oy >= 0 && min_ox >= 0 && max_oy < d->out_h && max_ox < d->out_w,
               "sp_conv2d_fwd: output mapping leaves the output tensor (max oy %d / out_h %d, max ox %d / out_w %d)", max_oy,
               d->out_h, max_ox, d->out_w);
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    const long long in_elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    const long long out_elems = (long long)d->batch * d->out_h * d->out_w * d->out_c;
    const long long w_elems = (long long)d->n_pad * d->k_pad;
    SP_REQUIRE(M < (1ll << 29) && in_elems < (1ll << 29) && out_elems < (1ll << 29) && w_elems < (1ll << 29),
               "sp_conv2d_fwd: tensor too large (each operand must stay below 2 GiB for 32-bit buffer offsets)");

    ConvArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.res_mask = bsrc ? reinterpret_cast<const unsigned char*>(bsrc->res_mask) : nullptr;
    SP_REQUIRE(!a.res_mask || (residual && out16 && !(d->flags & SP_CONV_OUT_NCHW)), "sp_conv2d_dgrad: a masked accumulate operand needs bf16 NHWC gradients");
    a.M = (int)M;
    a.in_h = d->in_h; a.in_w = d->in_w; a.c_in = d->c_in;
    a.grid_h = d->grid_h; a.grid_w = d->grid_w;
    a.c_out = d->c_out; a.n_pad = d->n_pad; a.k_pad = d->k_pad;
    a.taps_h = d->taps_h; a.taps_w = d->taps_w;
    a.stride = d->stride; a.stride_x = d->stride_x > 0 ? d->stride_x : d->stride; a.dy0 = d->dy0; a.dy_step = d->dy_step; a.dx0 = d->dx0; a.dx_step = d->dx_step;
    a.out_h = d->out_h; a.out_w = d->out_w; a.out_c = d->out_c;
    a.oy_mul = d->oy_mul; a.oy_add = d->oy_add; a.ox_mul = d->ox_mul; a.ox_add = d->ox_add;
    a.phases_x = d->phases_x; a.flags = d->flags; a.tiles_m = a.tiles_n = 0;
    a.x_bytes = (int)(in_elems * es); a.w_bytes = (int)(w_elems * es);
    a.y_bytes = (int)(out_elems * (((d->flags & SP_CONV_OUT_NCHW) || !out16) ? 4 : 2));
    a.stats_s = stats_s; a.stats_q = stats_q; a.stats_stride = d->n_pad;
    a.by = bsrc ? bsrc->y : nullptr; a.bz = bsrc ? bsrc->z : nullptr; a.bmean = bsrc ? bsrc->mean : nullptr; a.binvstd = bsrc ? bsrc->invstd : nullptr;
    a.bz2 = bsrc ? bsrc->z2 : nullptr; a.bmean2 = bsrc ? bsrc->mean2 : nullptr; a.binvstd2 = bsrc ? bsrc->invstd2 : nullptr; a.stats_q2 = bsrc ? bsrc->q2 : nullptr;
    a.bz_bytes = (int)(out_elems * es);
    a.c_in_g = d->c_in_group > 0 ? d->c_in_group : 0;
    a.abn_mean = nullptr; a.abn_invstd = a.abn_gamma = a.abn_beta = nullptr; a.abn_y = nullptr; a.abn_mask = nullptr;
    if (abn) {
        SP_REQUIRE(abn->mean && abn->invstd && abn->gamma && abn->beta && abn->y, "sp_conv2d_fwd_bn_stats_abn: null pointer");
        SP_REQUIRE(bf16 && out16 && stats_s && !bsrc && !phs && uniform && d->taps_h == 1 && d->taps_w == 1 && d->stride == 1 && d->stride_x <= 1 && d->dy0 == 0 &&
                       d->dx0 == 0 && d->in_h == d->grid_h && d->in_w == d->grid_w && d->phases_y == 1 && d->phases_x == 1 && d->c_in <= 512 && d->c_in_group == 0 &&
                       d->kernel == SP_CONV_KERNEL_IGEMM && d->k_pad / 64 < 12,
                   "sp_conv2d_fwd_bn_stats_abn: needs a bf16 1x1 stride-1 convolution with c_in <= 512 (< 12 K tiles) on the implicit-GEMM kernel");
        a.abn_mean = abn->mean; a.abn_invstd = abn->invstd; a.abn_gamma = abn->gamma; a.abn_beta = abn->beta; a.abn_y = abn->y; a.abn_mask = abn->mask;
    }
    a.ph_n = 0;
    int phases = d->phases_y * d->phases_x;
    if (phs) {                                         // every descriptor was validated on its own by the caller; here: what they must share
        SP_REQUIRE(phs->n >= 2 && phs->n <= 4 && phases == 1, "sp_conv2d_dgrad_phases: 2-4 single-phase descriptors");
        for (int i = 0; i < phs->n; ++i) {
            const sp_conv_desc& q = phs->descs[i];
            SP_REQUIRE(q.batch == d->batch && q.in_h == d->in_h && q.in_w == d->in_w && q.c_in == d->c_in && q.grid_h == d->grid_h &&
                           q.grid_w == d->grid_w && q.c_out == d->c_out && q.n_pad == d->n_pad && q.stride == d->stride && q.stride_x == d->stride_x &&
                           q.dy_step == d->dy_step && q.dx_step == d->dx_step && q.out_h == d->out_h && q.out_w == d->out_w && q.out_c == d->out_c &&
                           q.oy_mul == d->oy_mul && q.ox_mul == d->ox_mul && q.flags == d->flags && q.phases_y == 1 && q.phases_x == 1 &&
                           q.kernel == SP_CONV_KERNEL_IGEMM && phs->w[i],
                       "sp_conv2d_dgrad_phases: descriptor %d differs from descriptor 0 in more than its taps / offsets", i);
            const bool uq = (q.c_in % bke == 0) && q.taps_h * q.taps_w <= 32;
            SP_REQUIRE(uq == uniform, "sp_conv2d_dgrad_phases: descriptor %d: tap layout differs", i);
            a.ph[i].w = phs->w[i];
            a.ph[i].taps_h = q.taps_h; a.ph[i].taps_w = q.taps_w; a.ph[i].k_pad = q.k_pad;
            a.ph[i].dy0 = q.dy0; a.ph[i].dx0 = q.dx0; a.ph[i].oy_add = q.oy_add; a.ph[i].ox_add = q.ox_add; a.ph[i].pad_ = 0;
        }
        a.ph_n = phs->n;
        phases = phs->n;
    }
    hipStream_t s = (hipStream_t)stream;

    SP_REQUIRE(d->kernel == SP_CONV_KERNEL_IGEMM || d->kernel == SP_CONV_KERNEL_RING || d->kernel == SP_CONV_KERNEL_PW || d->kernel == SP_CONV_KERNEL_RING_LW || d->kernel == SP_CONV_KERNEL_RING_LW4,
               "sp_conv2d_fwd: unknown kernel id %d", d->kernel);
    if (d->kernel == SP_CONV_KERNEL_PW) {
        SP_REQUIRE(!stats_s && !bsrc, "sp_conv2d_fwd: the streaming 1x1 kernel has no statistics epilogue");
        return sp_conv_pw_launch(d, x, w_packed, scale, shift, residual, y, stream);
    }
    if (d->kernel == SP_CONV_KERNEL_RING || d->kernel == SP_CONV_KERNEL_RING_LW || d->kernel == SP_CONV_KERNEL_RING_LW4) {
        SP_REQUIRE(!stats_s && !bsrc, "sp_conv2d_fwd: the LDS-DMA ring kernel has no statistics epilogue");
        return sp_conv_ring_launch(d, x, w_packed, scale, shift, residual, y, stream);
    }
    // ---- tile shape: caller's choice (autotuned by the host, sp_conv_desc.tile_m/tile_n) or the built-in heuristic ----
    int bm = d->tile_m, bn = d->tile_n;
    const int np = d->n_pad;
    if (bm == 0 && bn == 0) sp_conv2d_default_tile(d, &bm, &bn);
    SP_REQUIRE(bn > 0 && np % bn == 0, "sp_conv2d_fwd: tile_n=%d must divide n_pad=%d", bn, np);
    if (bsrc) {
        SP_REQUIRE(stats_s && stats_q && bsrc->y && bsrc->z && bsrc->mean && bsrc->invstd, "sp_conv2d_dgrad_bn_bwd_stats: null pointer");
        SP_REQUIRE(!(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_RELU)) && d->c_out % (out16 ? 8 : 4) == 0 && !scale && !shift,
                   "sp_conv2d_dgrad_bn_bwd_stats: needs a plain NHWC gradient store (fp32: SP_CONV_OUT_F32 with bf16 operands; or bf16, c_out %% 8 == 0)");
        const long long rows = (long long)phases * ((M + bm - 1) / bm) * tile_rows_per_block(bm, bn);
        SP_REQUIRE(rows <= stats_rows_capacity, "sp_conv2d_dgrad_bn_bwd_stats: %lld partial rows needed, capacity %d", rows, stats_rows_capacity);
    } else if (stats_s) {
        SP_REQUIRE(stats_q, "sp_conv2d_fwd_bn_stats: null statistics pointer");
        SP_REQUIRE(!(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) && d->c_out % (bf16 ? 8 : 4) == 0,
                   "sp_conv2d_fwd_bn_stats: needs a plain NHWC store in the conv's own dtype (c_out %% %d == 0)", bf16 ? 8 : 4);
        const long long rows = (long long)phases * ((M + bm - 1) / bm) * tile_rows_per_block(bm, bn);
        SP_REQUIRE(rows <= stats_rows_capacity, "sp_conv2d_fwd_bn_stats: %lld partial rows needed, capacity %d", rows, stats_rows_capacity);
    }
#define SP_TILE(BM_, BN_, WR_, WC_) \
    if (bm == BM_ && bn == BN_) return launch<BM_, BN_, WR_, WC_>(a, phases, uniform, s);
    SP_TILE(128, 128, 2, 2)
    SP_TILE(64, 128, 2, 2)
    SP_TILE(128, 64, 2, 2)
    SP_TILE(64, 64, 2, 2)
    SP_TILE(256, 64, 4, 1)
    SP_TILE(128, 32, 4, 1)
#undef SP_TILE
    sp_set_error("sp_conv2d_fwd: unsupported tile %dx%d (supported: 128x128 64x128 128x64 64x64 256x64 128x32)", bm, bn);
    return SP_EINVAL;
}

extern "C" int sp_conv2d_fwd(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                             const float* shift, const void* residual, void* y, void* stream) {
    return conv_fwd_impl(d, x, w_packed, scale, shift, residual, y, nullptr, nullptr, 0, stream);
}

// Name of the kernel instantiation a launch of `d` resolves to (what rocprofv3's kernel trace calls it, minus the
// "void (anonymous namespace)::" / "(...Args)" decoration).  variant: 0 = sp_conv2d_fwd, 1 = sp_conv2d_fwd_bn_stats, 2 =
// sp_conv2d_dgrad_bn_bwd_stats, 3 = sp_conv3x3_direct, 4 = sp_basic_block_c32, 5 = sp_bottleneck_c64, 6 = sp_basic_block_c64 (`d` = the block's 3x3 convolution).  Nothing is
// launched; `d` is validated exactly as a launch would.
extern "C" int sp_conv2d_kernel_name(const sp_conv_desc* d, int has_residual, int variant, char* buf, int cap) {
    SP_REQUIRE(d && buf && cap > 0, "sp_conv2d_kernel_name: null pointer");
    SP_REQUIRE(variant >= 0 && variant <= 6, "sp_conv2d_kernel_name: variant %d", variant);
    void* const dummy = reinterpret_cast<void*>(16);        // never dereferenced: the launch functions return before launching
    if (variant >= 4) {                                     // the fused blocks' own dispatch (which of their kernels SP_BB32_W8 / SP_BNECK_W8 select)
        void* const dummy2 = reinterpret_cast<void*>(32);   // (x and y must differ)
        sp_name_query_begin();
        const int rcb = variant == 4 ? sp_basic_block_c32(d, dummy, dummy, nullptr, nullptr, dummy, nullptr, nullptr, dummy2, nullptr)
                      : variant == 6 ? sp_basic_block_c64(d, dummy, dummy, nullptr, nullptr, dummy, nullptr, nullptr, dummy2, nullptr)
                                     : sp_bottleneck_c64(d, dummy, dummy, nullptr, nullptr, dummy, nullptr, nullptr, dummy, nullptr, nullptr, dummy2, nullptr);
        const char* nameb = sp_name_query_end();
        if (rcb != SP_OK) return rcb;
        snprintf(buf, (size_t)cap, "%s", nameb);
        return SP_OK;
    }
    if (variant == 3) {                                     // asked of the direct kernels' own dispatch (c32 / c64 / c128 tile kernels, the 128 -> J head kernel)
        sp_name_query_begin();
        const int rc3 = sp_conv3x3_direct(d, dummy, dummy, nullptr, nullptr, has_residual ? dummy : nullptr, dummy, nullptr);
        const char* name3 = sp_name_query_end();
        if (rc3 != SP_OK) return rc3;
        snprintf(buf, (size_t)cap, "%s", name3);
        return SP_OK;
    }
    float* const fdummy = reinterpret_cast<float*>(16);
    const BnBwdSrc src = {dummy, dummy, fdummy, fdummy, nullptr, nullptr, nullptr, nullptr};
    sp_name_query_begin();
    const int rc = conv_fwd_impl(d, dummy, dummy, nullptr, nullptr, has_residual ? dummy : nullptr, dummy, variant ? fdummy : nullptr,
                                 variant ? fdummy : nullptr, 1 << 30, nullptr, variant == 2 ? &src : nullptr);
    const char* name = sp_name_query_end();
    if (rc != SP_OK) return rc;
    snprintf(buf, (size_t)cap, "%s", name);
    return SP_OK;
}

extern "C" int sp_conv2d_bn_stats_rows(const sp_conv_desc* d, int* rows) {
    if (!d || !rows) { sp_set_error("sp_conv2d_bn_stats_rows: null pointer"); return SP_EINVAL; }
    int bm = d->tile_m, bn = d->tile_n;
    if (bm == 0 && bn == 0) sp_conv2d_default_tile(d, &bm, &bn);
    SP_REQUIRE(bm > 0 && bn > 0, "sp_conv2d_bn_stats_rows: bad tile");
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    *rows = (int)((long long)d->phases_y * d->phases_x * ((M + bm - 1) / bm) * tile_rows_per_block(bm, bn));
    return SP_OK;
}

extern "C" int sp_conv2d_fwd_bn_stats(const sp_conv_desc* d, const void* x, const void* w_packed, void* y, float* stats_sum,
                                      float* stats_sumsq, int stats_rows_capacity, void* stream) {
    SP_REQUIRE(stats_sum && stats_sumsq, "sp_conv2d_fwd_bn_stats: null statistics pointer");
    return conv_fwd_impl(d, x, w_packed, nullptr, nullptr, nullptr, y, stats_sum, stats_sumsq, stats_rows_capacity, stream);
}

// sp_conv2d_fwd_bn_stats whose input is z of the PREVIOUS conv: relu(BatchNorm(z)) with that layer's batch statistics is formed while the A
// operand is staged, and written out (activation + ReLU bit mask) by the N-tile-0 workgroups - the stand-alone pass disappears.  Same bits as
// sp_bn_apply_nhwc (relu, mask) followed by sp_conv2d_fwd_bn_stats.
extern "C" int sp_conv2d_fwd_bn_stats_abn(const sp_conv_desc* d, const void* z_in, const float* in_mean, const float* in_invstd, const float* in_gamma,
                                          const float* in_beta, void* y_in, void* relu_mask_in, const void* w_packed, void* y, float* stats_sum,
                                          float* stats_sumsq, int stats_rows_capacity, void* stream) {
    SP_REQUIRE(stats_sum && stats_sumsq, "sp_conv2d_fwd_bn_stats_abn: null statistics pointer");
    const AbnSrc abn = {in_mean, in_invstd, in_gamma, in_beta, y_in, reinterpret_cast<unsigned char*>(relu_mask_in)};
    return conv_fwd_impl(d, z_in, w_packed, nullptr, nullptr, nullptr, y, stats_sum, stats_sumsq, stats_rows_capacity, stream, nullptr, nullptr, &abn);
}

extern "C" int sp_conv2d_dgrad_bn_bwd_stats(const sp_conv_desc* d, const void* dz, const void* w_packed, const void* accumulate, void* dx,
                                           const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd,
                                           float* sum_g, float* sum_g_xhat, int stats_rows_capacity, void* stream) {
    const BnBwdSrc src = {bn_y, bn_z, bn_mean, bn_invstd, nullptr, nullptr, nullptr, nullptr};
    return conv_fwd_impl(d, dz, w_packed, nullptr, nullptr, accumulate, dx, sum_g, sum_g_xhat, stats_rows_capacity, stream, &src);
}

extern "C" int sp_conv2d_dgrad_bn_bwd_stats2(const sp_conv_desc* d, const void* dz, const void* w_packed, const void* accumulate, void* dx,
                                            const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd,
                                            float* sum_g, float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd,
                                            float* sum_g_xhat2, int stats_rows_capacity, void* stream) {
    SP_REQUIRE(bn2_z && bn2_mean && bn2_invstd && sum_g_xhat2, "sp_conv2d_dgrad_bn_bwd_stats2: null pointer");
    const BnBwdSrc src = {bn_y, bn_z, bn_mean, bn_invstd, bn2_z, bn2_mean, bn2_invstd, sum_g_xhat2};
    return conv_fwd_impl(d, dz, w_packed, nullptr, nullptr, accumulate, dx, sum_g, sum_g_xhat, stats_rows_capacity, stream, &src);
}

// Built-in tile heuristic (measured on MI355X, bs=128 ResNet-50 shapes): 128x128 when the launch is many rounds deep,
// 64-row tiles when it is not (3 workgroups per CU instead of 2 balance the last round), 64x64 for the smallest M.
extern "C" int sp_conv2d_default_tile(const sp_conv_desc* d, int* tile_m, int* tile_n) {
    if (!d || !tile_m || !tile_n) { sp_set_error("sp_conv2d_default_tile: null pointer"); return SP_EINVAL; }
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    const int phases = (d->phases_y > 0 ? d->phases_y : 1) * (d->phases_x > 0 ? d->phases_x : 1);
    const int np = d->n_pad;
    if (np % 128 == 0) {
        const long long b128 = ((M + 127) / 128) * (np / 128) * phases;
        const long long b64 = ((M + 63) / 64) * (np / 128) * phases;
        if (b128 >= 2048) { *tile_m = 128; *tile_n = 128; }
        else if (b64 >= 768) { *tile_m = 64; *tile_n = 128; }
        else { *tile_m = 64; *tile_n = 64; }
    } else if (np % 64 == 0) {
        const long long b = ((M + 127) / 128) * (np / 64) * phases;
        if (b >= 1024) { *tile_m = 128; *tile_n = 64; } else { *tile_m = 64; *tile_n = 64; }
    } else { *tile_m = 128; *tile_n = 32; }
    return SP_OK;
}

// The dgrad of a stride-2 conv is one launch FAMILY: one descriptor per output phase (tap counts differ: 3x3 -> 2x2, 2x1, 1x2, 1x1).  This
// entry runs the whole family as ONE launch (blockIdx.y = phase, per-phase geometry in the kernel arguments): at 32 images each phase alone
// fills a fraction of the chip and costs a launch boundary on the step's dependent chain.  Partial rows of the BSTATS epilogue: phase-major,
// exactly where the one-launch-per-phase sequence puts them.  bn_* all NULL: plain dgrad (accumulate as sp_conv2d_fwd's residual).
extern "C" int sp_conv2d_dgrad_phases(const sp_conv_desc* descs, int n_phases, const void* dz, const void* const* w_packed, const void* accumulate,
                                      void* dx, const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                                      float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd, float* sum_g_xhat2,
                                      int stats_rows_capacity, void* stream) {
    SP_REQUIRE(descs && w_packed && n_phases >= 2 && n_phases <= 4, "sp_conv2d_dgrad_phases: 2-4 phase descriptors");
    SP_REQUIRE(!bn2_z || (bn_z && bn2_mean && bn2_invstd && sum_g_xhat2), "sp_conv2d_dgrad_phases: second BatchNorm needs the first and its own statistics");
    void* const dummy = reinterpret_cast<void*>(16);
    for (int i = 0; i < n_phases; ++i) {               // each descriptor through the full argument check of a launch of its own (nothing is launched)
        const bool outer = sp_name_query_active();
        if (!outer) sp_name_query_begin();
        const int rc = conv_fwd_impl(&descs[i], dummy, dummy, nullptr, nullptr, accumulate ? dummy : nullptr, dummy, nullptr, nullptr, 0, nullptr);
        if (!outer) sp_name_query_end();
        if (rc != SP_OK) return rc;
        SP_REQUIRE(descs[i].tile_m == descs[0].tile_m && descs[i].tile_n == descs[0].tile_n, "sp_conv2d_dgrad_phases: the phases share one tile");
    }
    const PhaseSet phs = {descs, w_packed, n_phases};
    if (bn_z) {
        const BnBwdSrc src = {bn_y, bn_z, bn_mean, bn_invstd, bn2_z, bn2_mean, bn2_invstd, sum_g_xhat2};
        return conv_fwd_impl(&descs[0], dz, w_packed[0], nullptr, nullptr, accumulate, dx, sum_g, sum_g_xhat, stats_rows_capacity, stream, &src, &phs);
    }
    return conv_fwd_impl(&descs[0], dz, w_packed[0], nullptr, nullptr, accumulate, dx, nullptr, nullptr, 0, stream, nullptr, &phs);
}

// The BSTATS dgrad launch of a 1x1 conv1 whose block input also feeds the block's residual add (an identity Bottleneck): the residual share
// of that input's gradient is g = dy_out * (ReLU mask of the block output).  Instead of bn3's backward pass writing g and this launch
// reading it back as `accumulate`, the launch takes (dy_out, mask_out) and forms g in its epilogue: 2 bytes per element less written and the
// same bits (g is dy_out or zero).  bf16 gradients only; bn2_* NULL: no second BatchNorm.
extern "C" int sp_conv2d_dgrad_bn_bwd_stats_macc(const sp_conv_desc* d, const void* dz, const void* w_packed, const void* acc_dy, const void* acc_mask,
                                                void* dx, const void* bn_y, const void* bn_z, const float* bn_mean, const float* bn_invstd, float* sum_g,
                                                float* sum_g_xhat, const void* bn2_z, const float* bn2_mean, const float* bn2_invstd, float* sum_g_xhat2,
                                                int stats_rows_capacity, void* stream) {
    SP_REQUIRE(acc_dy && acc_mask, "sp_conv2d_dgrad_bn_bwd_stats_macc: null accumulate operand");
    SP_REQUIRE(!bn2_z || (bn2_mean && bn2_invstd && sum_g_xhat2), "sp_conv2d_dgrad_bn_bwd_stats_macc: null pointer");
    BnBwdSrc src = {bn_y, bn_z, bn_mean, bn_invstd, bn2_z, bn2_mean, bn2_invstd, sum_g_xhat2};
    src.res_mask = acc_mask;
    return conv_fwd_impl(d, dz, w_packed, nullptr, nullptr, acc_dy, dx, sum_g, sum_g_xhat, stats_rows_capacity, stream, &src);
}
