// conv_ring.hip - bf16 implicit-GEMM convolution, second structure: persistent 8-wave workgroups fed by an LDS-DMA ring.
//
// Same contraction, same packed weights, same LDS row image and the same v_mfma_f32_32x32x16_bf16 chain per output as
// conv_igemm.hip (so both kernels give the same bits), but built for the layers where the 128x128 / 4-wave / register-staged
// kernel is bound by L2 -> CU traffic and by latency it cannot cover (deconvolutions, DUC convs, the 3x3 convs of
// nets/pose_resnet_dconv.py:101, :236-244 and nets/commons.py:31-32 in bf16):
//
//   * tiles of 256x256 / 256x128 / 128x256 / 256x64 / 128x128 with EIGHT waves (two per SIMD): per K tile a 256x256 workgroup
//     moves 64 KB for 8.4 MFLOP where a 128x128 one moves 32 KB for 2.1 MFLOP - half the bytes per FLOP through L2;
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, no staging registers,
//     no ds_write): the per-lane SOURCE offset does the implicit-GEMM gather (tap shift, zero fill through the descriptor's
//     range check) AND the XOR swizzle of the LDS row image, the destination is lane-linear as the hardware requires;
//   * a ring of NS K-tile slots with counted `s_waitcnt vmcnt(N)` and raw `s_barrier` (one per K tile): NS-1 K tiles stay in
//     flight across barriers;
//   * ONE workgroup per CU that walks its output tiles (XCD-aware order); the K-tile stream runs on across tile boundaries,
//     so the first K tiles of the next output tile are already landing while the current tile's epilogue runs - the short-K
//     layers no longer pay a load latency per tile.
#include "sp_common.h"
#include <type_traits>

#ifdef SP_RING_DIAG
// DIAGNOSTIC BUILD ONLY (tools/diag_ring.py; never the shipped library): per-wave cycle sums of the stage segments, read back with
// sp_ring_debug_read().  [block % 256][wave][8]: 0 vmcnt wait, 1 barrier wait, 2 fragment + MFMA section, 3 epilogues, 4 stages,
// 5 kernel lifetime, 6 100 MHz ticks of the lifetime, 7 output tiles
__device__ unsigned long long sp_ring_dbg[256 * 8 * 8];
extern "C" int sp_ring_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sp_ring_dbg), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
extern "C" int sp_ring_debug_clear() {
    static unsigned long long z[256 * 8 * 8];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sp_ring_dbg), z, sizeof(z), 0, hipMemcpyHostToDevice);
}
#define SP_RSTAMP(var)                                                                      \
    unsigned long long var;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

struct RingArgs {
    const void* x;       // NHWC bf16 activations
    const void* w;       // packed weights [phases][n_pad][k_pad] bf16
    const float* scale;
    const float* shift;
    const void* res;     // NHWC bf16, layout of y
    void* y;             // NHWC bf16
    int M;               // batch * grid_h * grid_w (rows of one phase)
    int in_h, in_w, c_in;
    int grid_h, grid_w;
    int c_out, n_pad, k_pad;
    int taps_h, taps_w;
    int stride, stride_x, dy0, dy_step, dx0, dx_step;
    int out_h, out_w, out_c;
    int oy_mul, oy_add, ox_mul, ox_add;
    int phases_x;
    unsigned flags;
    int tiles_m, tiles_n, total_tiles;   // total = phases * tiles_m * tiles_n
    int x_bytes, w_bytes, y_bytes;       // buffer-descriptor extents (w: all phases)
};

constexpr unsigned OOB = 0x80000000u;   // every tensor is < 2 GiB (host-checked): an offset the range check rejects

__device__ __forceinline__ u32x4 make_rsrc(const void* base, int bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4 r;                                   // (readfirstlane: an "s" asm operand must be provably wave-uniform)
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);                    // num_records (bytes)
    r[3] = 0x00020000u;
    return r;
}

// One LDS-DMA piece: 64 lanes x 16 bytes, lane l's bytes from `rsrc` base + voff + soff (zeros when voff is out of range), written
// to LDS at lds_addr + 16 * l (wave-uniform base in M0).  Inline asm on purpose: hipcc treats the builtin form as an LDS store it
// must wait for (`s_waitcnt vmcnt(0)` in front of every later LDS access), which would drain the ring at every K tile; issued
// from asm the transfers are invisible to its bookkeeping and ordered by this file's own counted `s_waitcnt vmcnt(N)` + s_barrier.
// M0 is written in the statement that uses it (hipcc keeps nothing live in M0 across statements on gfx950, and it does not accept
// "m0" in a clobber list - "inline asm clobber list contains reserved registers" - so the dependence cannot be declared); `s_nop 4` covers the
// M0-write -> LDS-DMA and the VALU-written-SGPR -> VMEM wait states, which nothing pads inside an asm statement.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, u32x4 rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff),
                 "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(soff))
                 : "memory");
}

// HAS_RES: a residual tensor is added in the epilogue (compile-time, so that its loads and their use sit on one path: with a run-time
// test on both, hipcc has to assume a load may still be pending at the next K tile and drains the ring there)
//
// NLW (round 5): 0 = every one of the eight MFMA waves also issues its share of the LDS-DMA pieces (rounds 2-4); 4 = the workgroup has four
// more waves (one per SIMD) that do nothing but issue the pieces ("loader waves"), the eight MFMA waves only read fragments and multiply.
// Why: a piece costs the wave that issues it ~150 cycles inside a phase that also carries ds_reads (tools/diag_ring.py; 25-60 when the wave
// does nothing else) - on the 192x128 tile 5 pieces per 12 MFMAs, i.e. more issue time than matrix time, in the very waves whose MFMA chain
// it interrupts.  The ring, the counted vmcnt + one barrier per K tile, the fragment reads and the MFMA chain per output are unchanged, so
// the bits are too (`test_ring_kernel_matches_igemm_on_ragged_shapes` covers both).
//
// MFMA waves (round 6): WR x WC = 8 (two per SIMD, rounds 2-5) or, with loader waves only, 4 (ONE per SIMD; kernel = SP_CONV_KERNEL_RING_LW4):
// the same workgroup tile cut into four wave tiles of twice the area - 192x128 as 2 x 2 waves of 96x64 reads 5 fragments per 6 MFMAs where
// eight 96x32 waves read 4 per 3 (LDS fragment bytes per K tile 128 -> 80 KB) - and tiles eight waves cannot cut (96 rows as 1 x 4 waves of
// 96x32: 256 tiles where 128-row tiles give 192).  Same slot image, same K order, same MFMA chain per output: same bits.
template <int BM, int BN, int WR, int WC, int NS, bool HAS_RES, int NLW>
__global__ __launch_bounds__(64 * (WR * WC + NLW), (WR * WC + NLW) / 4) void conv_ring_kernel(const RingArgs p) {
    constexpr int NMW = WR * WC;                       // MFMA waves
    static_assert(NMW == 8 || (NMW == 4 && NLW == 4), "8 MFMA waves, or 4 next to the 4 loader waves");
    static_assert(NLW == 0 || NLW == 4, "loader waves: none or one per SIMD");
    static_assert(BM <= 64 * NMW, "the row table of a tile is written by the MFMA waves' threads");
    constexpr int WM = BM / WR, WN = BN / WC;          // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;          // 32x32 MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1 && BM % 32 == 0 && BN % 64 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");
    constexpr int NIW = NLW ? NLW : NMW;               // waves that issue the LDS-DMA pieces
    static_assert((BM / 8) % NIW == 0 && (BN / 8) % NIW == 0, "whole pieces per issuing wave");
    constexpr int LA = BM / 8 / NIW, LB = BN / 8 / NIW, L = LA + LB;   // 1-KiB DMA pieces per issuing wave per K tile (8 rows of 128 B each)
    constexpr int SB = (BM + BN) * 128;                // bytes of one ring slot: A rows, then B rows
    constexpr int D = NS - 1;                          // K tiles in flight ahead of the one being multiplied
    constexpr int NM = 4 * TM * TN;                    // MFMAs per wave per K tile
    constexpr int TRS = WN + 4;                        // floats per row of the epilogue's transpose scratch (padded)
    static_assert(NMW * 16 * TRS * 4 <= SB, "epilogue scratch must fit in one ring slot");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;                                   // [NS][BM + BN][128 B]
    int4* const tab = reinterpret_cast<int4*>(smem + NS * SB);          // [2][BM]: per output row of a tile (see make_table)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int fr = lane & 31, fh = lane >> 5;
    const bool loader_wave = NLW > 0 && wave >= NMW;    // (wave-uniform)
    const int iw = NLW ? (wave >= NMW ? wave - NMW : 0) : wave;   // index among the issuing waves
#ifdef SP_RING_DIAG
    unsigned long long dg_wait = 0, dg_bar = 0, dg_mma = 0, dg_epi = 0, dg_tiles = 0;
    SP_RSTAMP(dg_entry)
    unsigned long long dg_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_rt0)::"memory");
#endif

    // ---- this workgroup's tiles: T = perm + G * i.  Blocks b and b+8 share an XCD (and its L2): each XCD takes a contiguous
    // chunk of every round of G tiles, N tiles fastest, so the workgroups that share A rows run on one L2 at the same time ----
    const int G = gridDim.x;
    int perm;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = G >> 3, r = G & 7;
        perm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int nt = (p.total_tiles - perm + G - 1) / G;     // >= 1: the host launches G <= total_tiles workgroups
    const int nk = p.k_pad >> 6;                           // K tiles per output tile (>= NS, host-checked)
    const int S = nt * nk;                                 // K-tile stages this workgroup streams

    // descriptors of the two DMA sources as plain SGPR quads: the LDS-DMA is issued from inline asm (below), built from kernel
    // arguments only, so provably wave-uniform
    const u32x4 xr = make_rsrc(p.x, p.x_bytes);
    const u32x4 wrs = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.y), (short)0, p.y_bytes, 0x00020000);

    // ---- per-tile row table: x = byte offset of tap (0,0), channel 0 of the row's receptive field; y = bit mask of the taps
    // that lie inside the image; w = element offset of the row's output pixel (-1: row beyond M) ----
    auto make_table = [&](int T, int4* dst) __attribute__((always_inline)) {
        int t = T;
        t /= p.tiles_n;
        const int tm = t % p.tiles_m;
        const int phase = t / p.tiles_m;
        const int py = phase / p.phases_x, px = phase - py * p.phases_x;
        if (tid < BM) {
            const int m = tm * BM + tid;
            int4 e;
            if (m < p.M) {
                const int gw = p.grid_w, ghw = p.grid_h * gw;
                const int b = m / ghw, rem = m - b * ghw;
                const int gy = rem / gw, gx = rem - gy * gw;
                const int iy0 = gy * p.stride + p.dy0 + py, ix0 = gx * p.stride_x + p.dx0 + px;
                e.x = ((b * p.in_h + iy0) * p.in_w + ix0) * p.c_in * 2;
                unsigned msk = 0;
                for (int ty = 0; ty < p.taps_h; ++ty)
                    for (int tx = 0; tx < p.taps_w; ++tx) {
                        const int iy = iy0 + ty * p.dy_step, ix = ix0 + tx * p.dx_step;
                        if ((unsigned)iy < (unsigned)p.in_h && (unsigned)ix < (unsigned)p.in_w) msk |= 1u << (ty * p.taps_w + tx);
                    }
                e.y = (int)msk;
                e.z = 0;
                const int oy = gy * p.oy_mul + p.oy_add + py, ox = gx * p.ox_mul + p.ox_add + px;
                e.w = ((b * p.out_h + oy) * p.out_w + ox) * p.out_c;
            } else {
                e.x = 0; e.y = 0; e.z = 0; e.w = -1;
            }
            dst[tid] = e;
        }
    };

    // ---- loader state: the K-tile stage that is issued next (all wave-uniform except the per-row registers) ----
    int ld_i = 0, ld_kt = 0, ld_tap = 0, ld_cofs = 0, ld_ty = 0, ld_tx = 0, ld_slot = 0;
    int a_off0[LA];
    unsigned a_mask[LA];
    unsigned b_voff[LB];
    unsigned b_tile_soff = 0;
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int brow = 8 * (iw + NIW * j) + (lane >> 3);
        b_voff[j] = (unsigned)(brow * p.k_pad * 2 + (((lane & 7) ^ ((brow >> 1) & 7)) << 4));
    }
    unsigned t_shift = 0, t_bit = 0, b_soff = 0;
    auto loader_begin = [&]() __attribute__((always_inline)) {
        if (ld_kt == 0) {                              // first K tile of an output tile: its rows and its weight slab
            int t = perm + G * ld_i;
            const int tn = t % p.tiles_n;
            t /= p.tiles_n;
            const int phase = t / p.tiles_m;
            b_tile_soff = (unsigned)((phase * p.n_pad + tn * BN) * p.k_pad * 2);
            const int4* tb = tab + (ld_i & 1) * BM;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int row = 8 * (iw + NIW * j) + (lane >> 3);
                const int4 e = tb[row];
                a_off0[j] = e.x + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
                a_mask[j] = (unsigned)e.y;
            }
        }
        t_shift = (unsigned)(((ld_ty * p.dy_step * p.in_w + ld_tx * p.dx_step) * p.c_in + ld_cofs) * 2);
        t_bit = 1u << ld_tap;
        b_soff = b_tile_soff + (unsigned)(ld_kt * 128);
    };
    const unsigned ring_lds = (unsigned)(size_t)(lds_void_t*)ring;   // LDS byte address of the ring (0 in practice: dynamic LDS starts the segment)
    auto loader_piece = [&](int o) __attribute__((always_inline)) {                    // o in [0, L): one 1-KiB LDS-DMA piece
        const unsigned slot = ring_lds + (unsigned)(ld_slot * SB);
        if (o < LA) {
            const unsigned off = (a_mask[o] & t_bit) ? (unsigned)a_off0[o] + t_shift : OOB;
            dma16(slot + (unsigned)((iw + NIW * o) * 1024), off, xr, 0u);
        } else {
            const int j = o - LA;
            dma16(slot + (unsigned)(BM * 128 + (iw + NIW * j) * 1024), b_voff[j], wrs, b_soff);
        }
    };
    auto loader_advance = [&]() __attribute__((always_inline)) {
        ld_slot = (ld_slot + 1 == NS) ? 0 : ld_slot + 1;
        ld_cofs += 64;
        if (ld_cofs == p.c_in) {
            ld_cofs = 0;
            ++ld_tap;
            if (++ld_tx == p.taps_w) { ld_tx = 0; ++ld_ty; }
        }
        if (++ld_kt == nk) { ld_kt = 0; ++ld_i; ld_tap = 0; ld_cofs = 0; ld_ty = 0; ld_tx = 0; }
    };

    // ---- prologue: table of the first tile, then D stages in flight ----
    make_table(perm, tab);
    __syncthreads();
    if (NLW == 0 || loader_wave) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d < S) {
                loader_begin();
#pragma unroll
                for (int o = 0; o < L; ++o) loader_piece(o);
                loader_advance();
            }
        }
    }
    if constexpr (NLW > 0) {
        if (loader_wave) {
            // ---- a loader wave's whole life: per stage g, wait until ITS pieces of stage g have landed, meet the MFMA waves at the stage's
            // barrier (they are then done with stage g-1, whose slot is the one stage g+D goes into), issue stage g+D; and stand at the
            // extra barrier the MFMA waves take before an output tile's epilogue.  Same number of barriers on every wave of the workgroup.
            static_assert((NS - 1) * L <= 56, "outstanding pieces must fit the vmcnt counter");
            int ckt = 0;
            for (int g = 0; g < S; ++g) {
                if (g + D < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (g + D < S) {
                    loader_begin();
#pragma unroll
                    for (int o = 0; o < L; ++o) loader_piece(o);
                    loader_advance();
                }
                if (++ckt == nk) {
                    ckt = 0;
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    // fragment addressing: row image [rows][128 B], 16-byte chunk c of row r at position c ^ ((r >> 1) & 7); this lane reads row fr
    // (+ tile offsets, multiples of 32), chunk 2*j + fh for k-step j
    const int s7 = (fr >> 1) & 7;
    int fpos[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fpos[j] = fr * 128 + (((2 * j + fh) ^ s7) << 4);

    f32x4 fa[2][TM], fb[2][TN];
    const unsigned char* sa = ring;
    const unsigned char* sb = ring;
    auto read_frags = [&](int j, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[buf][i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * 128 + fpos[j]);
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[buf][n] = *reinterpret_cast<const f32x4*>(sb + n * 32 * 128 + fpos[j]);
    };

    // ---- epilogue of one output tile: y = act(acc * scale + shift (+ residual)) as bf16 NHWC.  A 32x32 accumulator holds a column
    // per lane; 16-row slabs of the wave tile are transposed through the wave's private piece of the ring slot that was multiplied
    // last, so that a lane owns 8 consecutive channels of one pixel: 16-byte residual loads and stores, whole 64/128-byte row
    // segments per instruction (same arithmetic, element by element, as conv_igemm.hip's epilogue) ----
    auto epilogue = [&](int ti, int slot) __attribute__((always_inline)) {
        int t = perm + G * ti;
        const int n0 = (t % p.tiles_n) * BN;
        float* tr = reinterpret_cast<float*>(ring + slot * SB) + wave * (16 * TRS);
        const int4* tb = tab + (ti & 1) * BM;
        constexpr int CPR = WN / 8;          // lanes (16-byte chunks of 8 channels) per row
        constexpr int RPI = 64 / CPR;        // rows per wave-instruction
        constexpr int NIT = 16 / RPI;        // instructions per 16-row slab
        const int chunk = lane % CPR, rsub = lane / CPR;
        const int col = n0 + wc * WN + chunk * 8;
        const bool col_ok = col < p.c_out;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (col_ok) {
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) {
                if (p.scale) { const f32x4 v = *reinterpret_cast<const f32x4*>(p.scale + col + 4 * e4); sc[4 * e4] = v[0]; sc[4 * e4 + 1] = v[1]; sc[4 * e4 + 2] = v[2]; sc[4 * e4 + 3] = v[3]; }
                if (p.shift) { const f32x4 v = *reinterpret_cast<const f32x4*>(p.shift + col + 4 * e4); sh[4 * e4] = v[0]; sh[4 * e4 + 1] = v[1]; sh[4 * e4 + 2] = v[2]; sh[4 * e4 + 3] = v[3]; }
            }
        }
        int col_off = col;
        if (p.flags & SP_CONV_PIXEL_SHUFFLE) {
            const int sub = col / p.out_c, c = col - sub * p.out_c;   // packed column order: sub-pixel major
            col_off = ((sub >> 1) * p.out_w + (sub & 1)) * p.out_c + c;
        }
        const bool relu = p.flags & SP_CONV_RELU;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned off[2][NIT];
            u32x4 rv[2][NIT];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int ro = tb[wr * WM + i * 32 + hh * 16 + it * RPI + rsub].w;
                    off[hh][it] = (col_ok && ro >= 0) ? (unsigned)((ro + col_off) * 2) : OOB;
                    if constexpr (HAS_RES) rv[hh][it] = __builtin_amdgcn_raw_buffer_load_b128(rr, off[hh][it], 0, 0);
                }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        tr[((q & 3) + 8 * (q >> 2) + 4 * fh) * TRS + n * 32 + fr] = acc[i][n][8 * hh + q];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float v[8];
                    const float* src = tr + (it * RPI + rsub) * TRS + chunk * 8;
#pragma unroll
                    for (int e4 = 0; e4 < 2; ++e4) {
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(src + 4 * e4);
                        v[4 * e4] = tv[0]; v[4 * e4 + 1] = tv[1]; v[4 * e4 + 2] = tv[2]; v[4 * e4 + 3] = tv[3];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                    if constexpr (HAS_RES) {
                        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[hh][it]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, off[hh][it], 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
    };

#define SP_SB() __builtin_amdgcn_sched_barrier(0)
    // ---- the stream: stage g = K tile kt of this workgroup's ti-th output tile, in ring slot g % NS ----
    int ti = 0, kt = 0, slot = 0;
    auto stage = [&](auto issue_tag) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issue_tag)::value;   // a stage D ahead exists and is requested during this one
        if (kt == 0 && ti + 1 < nt) make_table(perm + G * (ti + 1), tab + ((ti + 1) & 1) * BM);   // read from the next barrier on
        if (ISSUE) loader_begin();
        sa = ring + slot * SB + (wr * WM) * 128;
        sb = ring + slot * SB + BM * 128 + (wc * WN) * 128;
        read_frags(0, 0);
        SP_SB();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < 3) read_frags(j + 1, (j + 1) & 1);
            SP_SB();
#pragma unroll
            for (int q = 0; q < TM * TN; ++q) {
                const int i = q / TN, n = q % TN;
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[j & 1][i]), __builtin_bit_cast(bf16x8, fb[j & 1][n]),
                                                                    acc[i][n], 0, 0, 0);
                if (ISSUE) {
                    const int qq = j * TM * TN + q;        // every DMA piece in the shadow of an MFMA of its own
                    if constexpr (NS == 2) {
                        // one K tile ahead only: the pieces must land before the NEXT barrier, so they go out with the first L MFMAs
                        // of the stage instead of being spread over all of it (tools/diag_ring.py: the wait at the next stage's
                        // vmcnt(0) went from 330-470 to 45-85 of ~3,800 cycles per stage; +4-5 % on the DUC convs.  With two or
                        // three tiles ahead - NS >= 3 - front-loading measured neutral to slightly worse)
                        if (qq < L) loader_piece(qq);
                    } else {
#pragma unroll
                        for (int o = (qq * L) / NM; o < ((qq + 1) * L) / NM; ++o) loader_piece(o);
                    }
                }
                SP_SB();
            }
        }
        if (ISSUE) loader_advance();
        const int used = slot;
        slot = (slot + 1 == NS) ? 0 : slot + 1;
        if (++kt == nk) {
#ifdef SP_RING_DIAG
            SP_RSTAMP(dg_e0)
#endif
            __builtin_amdgcn_s_barrier();                 // every wave is done reading `used`: it becomes the transpose scratch
            SP_SB();
            epilogue(ti, used);
#ifdef SP_RING_DIAG
            SP_RSTAMP(dg_e1)
            dg_epi += dg_e1 - dg_e0;
            dg_mma -= dg_e1 - dg_e0;                      // (the caller books the whole stage as MFMA section)
            ++dg_tiles;
#endif
            kt = 0;
            ++ti;
        }
    };
    // stage g has landed once at most the D-1 younger stages are still outstanding (each wave waits for ITS pieces; the barrier then
    // makes every wave's pieces visible, and says everyone is done reading the slot the new DMA overwrites)
    int g = 0;
#ifdef SP_RING_DIAG
#define SP_RD0 SP_RSTAMP(dg_t0)
#define SP_RD1 SP_RSTAMP(dg_t1)
#define SP_RD2 SP_RSTAMP(dg_t2)
#define SP_RD3 { SP_RSTAMP(dg_t3) dg_wait += dg_t1 - dg_t0; dg_bar += dg_t2 - dg_t1; dg_mma += dg_t3 - dg_t2; }
#else
#define SP_RD0
#define SP_RD1
#define SP_RD2
#define SP_RD3
#endif
    if constexpr (NLW > 0) {
        for (; g < S; ++g) {                               // MFMA waves of the loader-wave variant: nothing to wait for but the barrier
            __builtin_amdgcn_s_barrier();
            SP_SB();
            stage(std::false_type{});
        }
    } else {
    for (; g + D < S; ++g) {
        SP_RD0
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * L) : "memory");
        SP_RD1
        __builtin_amdgcn_s_barrier();
        SP_SB();
        SP_RD2
        stage(std::true_type{});
        SP_RD3
    }
    for (; g < S; ++g) {                                   // the last D stages: nothing left to request
        SP_RD0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SP_RD1
        __builtin_amdgcn_s_barrier();
        SP_SB();
        SP_RD2
        stage(std::false_type{});
        SP_RD3
    }
    }
#undef SP_SB
#ifdef SP_RING_DIAG
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SP_RSTAMP(dg_end)
        unsigned long long dg_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_rt1)::"memory");
        if (lane == 0) {
            unsigned long long* o = sp_ring_dbg + ((blockIdx.x & 255) * 8 + wave) * 8;
            o[0] = dg_wait; o[1] = dg_bar; o[2] = dg_mma; o[3] = dg_epi; o[4] = (unsigned long long)S; o[5] = dg_end - dg_entry; o[6] = dg_rt1 - dg_rt0; o[7] = dg_tiles;
        }
    }
#endif
}

int device_cus() {                       // CUs of the current device (cached per device index)
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev] == 0) {
        int v = 0;
        cache[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cache[dev];
}

template <int BM, int BN, int WR, int WC, int NS, bool HAS_RES, int NLW>
int launch_ring_t(const RingArgs& a, hipStream_t stream) {
    if (sp_name_query_active()) {
        sp_name_query_set("conv_ring_kernel<%d, %d, %d, %d, %d, %s, %d>", BM, BN, WR, WC, NS, HAS_RES ? "true" : "false", NLW);   // (as rocprofv3 prints it)
        return SP_OK;
    }
    RingArgs p = a;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = a.n_pad / BN;
    const int phases = a.total_tiles;     // on entry: number of phases
    p.total_tiles = phases * p.tiles_m * p.tiles_n;
    const int cus = device_cus();
    const int grid = p.total_tiles < cus ? p.total_tiles : cus;   // one workgroup per CU (its LDS ring takes the CU's whole LDS)
    const size_t lds = (size_t)NS * (BM + BN) * 128 + (size_t)2 * BM * sizeof(int4);
    const void* fn = reinterpret_cast<const void*>(&conv_ring_kernel<BM, BN, WR, WC, NS, HAS_RES, NLW>);
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // per device: set before every launch
    if (e != hipSuccess) {
        sp_set_error("conv_ring: hipFuncSetAttribute(max dynamic LDS = %zu) failed: %s", lds, hipGetErrorString(e));
        return SP_ELAUNCH;
    }
    hipLaunchKernelGGL((conv_ring_kernel<BM, BN, WR, WC, NS, HAS_RES, NLW>), dim3(grid, 1, 1), dim3(64 * (WR * WC + NLW), 1, 1), lds, stream, p);
    return sp_check_launch("conv_ring_kernel");
}

template <int BM, int BN, int WR, int WC, int NS, int NLW = 0>
int launch_ring(const RingArgs& a, hipStream_t stream) {
    return a.res ? launch_ring_t<BM, BN, WR, WC, NS, true, NLW>(a, stream) : launch_ring_t<BM, BN, WR, WC, NS, false, NLW>(a, stream);
}

struct RingTile { int bm, bn, ns; };
// (192-row tiles: at bs=128 the layer3 / layer4 GEMMs of ResNet-50 cut into exactly 192 tiles of 128x256 / 256x128 / 256x256 - three
// quarters of the 256 CUs for one round; 192x128 and 192x256 give 256 tiles of three quarters the work)
constexpr RingTile kRingTiles[] = {{256, 256, 2}, {256, 128, 3}, {128, 256, 3}, {256, 64, 3}, {128, 128, 4}, {192, 128, 3}, {192, 256, 2}};

// the loader-wave variant (kernel = SP_CONV_KERNEL_RING_LW: 12 waves, three per SIMD, so at most 168 VGPRs): every tile whose MFMA waves fit that
// budget - the 256x256 tile (221 VGPRs) and the 192x256 tile (179) stay on the 8-wave kernel
constexpr RingTile kRingTilesLW[] = {{256, 128, 3}, {128, 256, 3}, {256, 64, 3}, {128, 128, 4}, {192, 128, 3}};

// four MFMA waves (one per SIMD) + four loader waves (kernel = SP_CONV_KERNEL_RING_LW4, round 6): eight waves per workgroup, 256 VGPRs each
constexpr RingTile kRingTilesLW4[] = {{192, 128, 3}, {128, 128, 4}, {96, 128, 4}, {256, 128, 3}, {128, 256, 3}, {96, 256, 3}, {64, 128, 5}};

const RingTile* find_tile(int bm, int bn, int kernel = SP_CONV_KERNEL_RING) {
    if (kernel == SP_CONV_KERNEL_RING_LW4) {
        for (const RingTile& t : kRingTilesLW4)
            if (t.bm == bm && t.bn == bn) return &t;
        return nullptr;
    }
    if (kernel == SP_CONV_KERNEL_RING_LW) {
        for (const RingTile& t : kRingTilesLW)
            if (t.bm == bm && t.bn == bn) return &t;
        return nullptr;
    }
    for (const RingTile& t : kRingTiles)
        if (t.bm == bm && t.bn == bn) return &t;
    return nullptr;
}

}  // namespace

// 1 when sp_conv2d_fwd can run `d` on the LDS-DMA ring kernel with workgroup tile d->tile_m x d->tile_n (kernel = SP_CONV_KERNEL_RING)
extern "C" int sp_conv2d_ring_ok(const sp_conv_desc* d) {
    if (!d || d->c_in_group > 0) return 0;             // (grouped convolutions: implicit GEMM only)
    const RingTile* t = find_tile(d->tile_m, d->tile_n, d->kernel);   // (kernel = SP_CONV_KERNEL_RING_LW: the loader-wave tiles; anything else: the 8-wave ring's)
    if (!t) return 0;
    const unsigned allowed = SP_CONV_RELU | SP_CONV_PIXEL_SHUFFLE | SP_CONV_BF16;
    if (!(d->flags & SP_CONV_BF16) || (d->flags & ~allowed)) return 0;                 // bf16 in, bf16 NHWC out
    if (d->c_in <= 0 || d->c_in % 64 || d->taps_h * d->taps_w > 32) return 0;           // a K tile lies inside one tap
    if (d->k_pad != d->taps_h * d->taps_w * d->c_in || d->k_pad / 64 < t->ns) return 0; // the ring must fit inside one output tile's K
    if (d->c_out % 8 || d->n_pad % t->bn) return 0;
    // the epilogue hands each lane 8 consecutive packed columns as ONE 16-byte store: with the fused PixelShuffle a lane's chunk must
    // stay inside one sub-pixel (out_c = c_out / 4 a multiple of 8, no padded rows), without it the columns are the output channels
    if (d->flags & SP_CONV_PIXEL_SHUFFLE) {
        if (d->out_c % 8 || d->out_c * 4 != d->c_out || d->n_pad != d->c_out) return 0;
    } else if (d->out_c != d->c_out) {
        return 0;
    }
    return 1;
}

int sp_conv_ring_launch(const sp_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                        const void* residual, void* y, void* stream) {
    SP_REQUIRE(sp_conv2d_ring_ok(d), "sp_conv2d_fwd: descriptor / tile %dx%d not supported by the LDS-DMA ring kernel (bf16 NHWC in and out, "
               "c_in %% 64 == 0, k_pad / 64 >= ring depth, tile_n | n_pad; tiles 256x256 256x128 128x256 256x64 128x128 192x128 192x256; loader-wave "
               "variant: all but 256x256 and 192x256; four-MFMA-wave variant: 192x128 128x128 96x128 256x128 128x256 96x256 64x128)", d->tile_m, d->tile_n);
    const long long M = (long long)d->batch * d->grid_h * d->grid_w;
    const long long in_elems = (long long)d->batch * d->in_h * d->in_w * d->c_in;
    const long long out_elems = (long long)d->batch * d->out_h * d->out_w * d->out_c;
    const int phases = d->phases_y * d->phases_x;
    const long long w_elems = (long long)phases * d->n_pad * d->k_pad;
    SP_REQUIRE(M < (1ll << 29) && in_elems < (1ll << 30) && out_elems < (1ll << 30) && w_elems < (1ll << 30),
               "sp_conv2d_fwd: tensor too large (each operand must stay below 2 GiB for 32-bit buffer offsets)");
    RingArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.M = (int)M;
    a.in_h = d->in_h; a.in_w = d->in_w; a.c_in = d->c_in;
    a.grid_h = d->grid_h; a.grid_w = d->grid_w;
    a.c_out = d->c_out; a.n_pad = d->n_pad; a.k_pad = d->k_pad;
    a.taps_h = d->taps_h; a.taps_w = d->taps_w;
    a.stride = d->stride; a.stride_x = d->stride_x > 0 ? d->stride_x : d->stride;
    a.dy0 = d->dy0; a.dy_step = d->dy_step; a.dx0 = d->dx0; a.dx_step = d->dx_step;
    a.out_h = d->out_h; a.out_w = d->out_w; a.out_c = d->out_c;
    a.oy_mul = d->oy_mul; a.oy_add = d->oy_add; a.ox_mul = d->ox_mul; a.ox_add = d->ox_add;
    a.phases_x = d->phases_x; a.flags = d->flags;
    a.tiles_m = a.tiles_n = 0; a.total_tiles = phases;
    a.x_bytes = (int)(in_elems * 2); a.w_bytes = (int)(w_elems * 2); a.y_bytes = (int)(out_elems * 2);
    hipStream_t s = (hipStream_t)stream;
    const int bm = d->tile_m, bn = d->tile_n;
    if (d->kernel == SP_CONV_KERNEL_RING_LW4) {
        if (bm == 192 && bn == 128) return launch_ring<192, 128, 2, 2, 3, 4>(a, s);     // 96x64 per wave
        if (bm == 128 && bn == 128) return launch_ring<128, 128, 2, 2, 4, 4>(a, s);     // 64x64
        if (bm == 96 && bn == 128) return launch_ring<96, 128, 1, 4, 4, 4>(a, s);       // 96x32
        if (bm == 256 && bn == 128) return launch_ring<256, 128, 2, 2, 3, 4>(a, s);     // 128x64
        if (bm == 128 && bn == 256) return launch_ring<128, 256, 2, 2, 3, 4>(a, s);     // 64x128
        if (bm == 96 && bn == 256) return launch_ring<96, 256, 1, 4, 3, 4>(a, s);       // 96x64
        if (bm == 64 && bn == 128) return launch_ring<64, 128, 1, 4, 5, 4>(a, s);       // 64x32
        sp_set_error("sp_conv2d_fwd: four-wave ring tile %dx%d not instantiated", bm, bn);
        return SP_EINVAL;
    }
    if (d->kernel == SP_CONV_KERNEL_RING_LW) {
        if (bm == 256 && bn == 128) return launch_ring<256, 128, 4, 2, 3, 4>(a, s);
        if (bm == 128 && bn == 256) return launch_ring<128, 256, 2, 4, 3, 4>(a, s);
        if (bm == 256 && bn == 64) return launch_ring<256, 64, 4, 2, 3, 4>(a, s);
        if (bm == 128 && bn == 128) return launch_ring<128, 128, 2, 4, 4, 4>(a, s);
        if (bm == 192 && bn == 128) return launch_ring<192, 128, 2, 4, 3, 4>(a, s);
        sp_set_error("sp_conv2d_fwd: loader-wave ring tile %dx%d not instantiated", bm, bn);
        return SP_EINVAL;
    }
    if (bm == 256 && bn == 256) return launch_ring<256, 256, 2, 4, 2>(a, s);
    if (bm == 256 && bn == 128) return launch_ring<256, 128, 4, 2, 3>(a, s);
    if (bm == 128 && bn == 256) return launch_ring<128, 256, 2, 4, 3>(a, s);
    if (bm == 256 && bn == 64) return launch_ring<256, 64, 4, 2, 3>(a, s);
    if (bm == 128 && bn == 128) return launch_ring<128, 128, 2, 4, 4>(a, s);
    if (bm == 192 && bn == 128) return launch_ring<192, 128, 2, 4, 3>(a, s);
    if (bm == 192 && bn == 256) return launch_ring<192, 256, 2, 4, 2>(a, s);
    sp_set_error("sp_conv2d_fwd: ring tile %dx%d not instantiated", bm, bn);
    return SP_EINVAL;
}
