// conv_bneck.hip - one ResNet-50 Bottleneck (nets/pose_resnet_dconv.py:112-133, stride 1, identity shortcut) per launch, bf16:
//
//     y = relu( bn3(conv1x1_{64->256}( relu(bn2(conv3x3_{64->64}( relu(bn1(conv1x1_{256->64}(x))) ))) )) + x )
//
// (layer1.1 / layer1.2 of the pose nets at 64x48: 256 -> 64 -> 64 -> 256 channels).  Run conv by conv the block moves ~960 MB at bs=128
// (x is read by conv1 and again as conv3's residual, the two 64-channel intermediates are written and read) against 402 MB for x in and
// y out; layer1 and layer2 already sit on the HBM roofline conv by conv (profiles/r02_dconv_bf16_floors.md), so only fewer bytes move them.
//
// One persistent 8-wave workgroup per CU, 16x8-pixel output tiles (the 3x3 stage is the round-2 64-channel direct kernel's):
//   A  t1 = relu(bn1(x . W1^T)) on the 18x10-pixel halo (192-row GEMM, K = 256).  x is loaded as whole 128-byte pieces of a pixel row (a
//      wave-instruction = 8 pixels x 8 chunks: eight cache lines - round 6; the MFMA fragment gather it replaces, 32 pixels x 32 bytes per
//      instruction, cost ~110-140 cycles of address coalescing per load, 78 loads per tile) and turned into fragments through a small
//      per-wave LDS slot (ds_write_b128 + ds_read_b128 on the ring kernel's swizzled [32 rows][128 B] image); W1's fragments from LDS;
//      t1 -> LDS as bf16 (zero outside the image: the 3x3's padding applies to t1);
//   B  t2 = relu(bn2(conv3x3(t1))) from the LDS halo tile, W2 resident in LDS in fragment order (72 KiB); t2 -> the wave's LDS slice;
//   C  y = relu(bn3(t2 . W3^T) + x): K = 64, N = 256 (8 column blocks, 128 accumulator registers); W3's fragments come from L2 (32 KB,
//      requested during stage B), the residual from the centre pixels of x; 16-byte stores through the per-wave transpose scratch.
// Every accumulation chain is the per-conv kernels' (k steps of 16 in order, one v_mfma_f32_32x32x16_bf16 chain per output), the
// intermediates are rounded to bf16 exactly where the per-conv program stores them -> bit-identical results, which is how it is tested.
#include "sp_common.h"

#include <stdlib.h>
#include <type_traits>

#ifdef SP_BNECK_DIAG
// DIAGNOSTIC BUILD ONLY (tools/diag_bneck.py; never the shipped library): per-wave cycle sums of the stages, read back with
// sp_bneck_debug_read().  [block % 256][wave][8]: 0 stage A loop, 1 t1 store, 2 barrier waits, 3 stage B loop, 4 stage C (incl. the next
// tile's x requests), 6 kernel lifetime, 7 tiles
__device__ unsigned long long sp_bneck_dbg[256 * 8 * 8];
extern "C" int sp_bneck_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sp_bneck_dbg), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#define SP_BSTAMP(var)                                                                      \
    unsigned long long var;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#define SP_BACC(slot, a, b) dg[slot] += (b) - (a);
#else
#define SP_BSTAMP(var)
#define SP_BACC(slot, a, b)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BT_H = 16, BT_W = 8;                 // output tile: 16 rows x 8 columns; wave w = rows 4w..4w+3
constexpr int BH_H = BT_H + 2, BH_W = BT_W + 2;    // halo tile 18 x 10 = 180 pixels
constexpr int NHALO = BH_H * BH_W;
constexpr int CM = 64, CIO = 256;                  // mid / in-out channels
constexpr int PIXM = CM * 2;                       // 128 B per t1 pixel = 8 chunks of 16 B
constexpr int W2B = 9 * 4 * 2 * 2 * 32 * 16;       // [tap][k step][column block][k half][column] 16-B fragments = 73,728 B
constexpr int T1B = NHALO * PIXM;                  // 23,040 B: t1 on the halo
constexpr int T2B = 128 * PIXM;                    // 16,384 B: t2 of the tile, bf16 [128 pixels][64 channels]
constexpr int W1B = 16 * 2 * 2 * 32 * 16;          // [k step][column block][k half][column]                     = 32,768 B
constexpr int TRB = 32 * 32 * 4;                   // per wave: fp32 transpose slab of one 32x32 accumulator (4 KB) = one x staging slot [32 pixels][128 B]
constexpr int LDSB = W2B + W1B + T1B + T2B + 4 * TRB;   // 162,304 B of the CU's 163,840
constexpr unsigned OOB = 0x80000000u;

struct BneckArgs {
    const void* x;        // NHWC bf16 [B,H,W,256]
    void* y;              // NHWC bf16 [B,H,W,256]
    const void* w1;       // packed [>=64][256] bf16
    const void* w2;       // packed [>=64][576] bf16, K = (tap, channel)
    const void* w3;       // packed [256][64] bf16
    const float *s1, *b1, *s2, *b2, *s3, *b3;      // folded BatchNorms (scale, shift)
    int H, W, batch, tiles_x, tiles_y;
    int x_bytes, w1_bytes, w2_bytes, w3_bytes;
};

// chunk c (0..7) of halo pixel (hy, hx) -> byte offset (conv_direct.hip x6off: conflict-free ds_read_b128 beats of the 3x3 stage)
__device__ __forceinline__ int t1off(int hy, int hx, int c) {
    return (hy * BH_W + hx) * PIXM + ((c ^ (((hx >> 1) & 1) | ((hy & 3) << 1))) << 4);
}

// Four waves, one per SIMD (512 registers each), the three stages one after the other per tile.  What the measured versions taught
// (tools/diag_bneck.py, cycles per tile of a wave; MFMA work is 4,900):
//   * W1 / W3 fragments and the residual fetched from L2 inside each tile: every round trip exposed, 40,000 cycles per tile (228 us);
//   * (round 6) x fetched as MFMA fragments (32 pixels x 32 bytes per wave-instruction): ~110-140 cycles of address coalescing per load,
//     78 of them per tile = a third of the tile's 30,600 cycles.  Now whole 128-byte pieces + an LDS transposition (this file's header);
//   * eight waves in two roles (stage A one tile ahead | stages B, C): the 256-register cap spilled the 3x3 loop's addresses to scratch
//     (a scratch load per MFMA pair: 25,000 cycles for 72 MFMAs) and the roles fought for one SIMD's issue slots (236-245 us);
//   * the t1 store with a division by 10 per accumulator register: 10,400 cycles of vector instructions.
// Hence: W1 and W2 resident in LDS in fragment order, W3's 8 fragments of a wave's 64 output channels and nothing else per tile from
// L2; x requested 6 units (24 KiB per wave) ahead AND the next tile's first 6 units requested before stage C of the current one; halo coordinates
// of the t1 store folded to compile-time constants; stage C split by output channels (wave w: channels 64w..64w+63 of all 128 pixels,
// from the shared t2 tile), the residual of a 32-pixel block requested one block ahead.
__global__ __launch_bounds__(256, 1) void bottleneck_c64_kernel(const BneckArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
    unsigned char* const Ws2 = smemb;
    unsigned char* const Ws1 = smemb + W2B;
    unsigned char* const T1 = smemb + W2B + W1B;
    unsigned char* const T2 = smemb + W2B + W1B + T1B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tr = reinterpret_cast<float*>(smemb + W2B + W1B + T1B + T2B + wave * TRB);
    // this wave's x staging slot is its transpose slab: ONE slot is enough - a unit's fragments are read into registers right after its
    // pieces are written, and the LDS executes a wave's operations in order, so the next unit's writes cannot overtake those reads
    unsigned char* const xslot = reinterpret_cast<unsigned char*>(tr);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;
    const int G = gridDim.x;
#ifdef SP_BNECK_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SP_BSTAMP(tk0)

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), (short)0, p.w3_bytes, 0x00020000);

    // ---- filters -> LDS in fragment order, once per workgroup: fragment (f = k step, nb, kh, n) = W[nb*32 + n][f*16 + kh*8 .. +8] ----
    for (int q = tid; q < W2B / 16; q += 256) {
        const int n = q & 31, kh = (q >> 5) & 1, nb = (q >> 6) & 1, f = q >> 7;
        *reinterpret_cast<u32x4*>(Ws2 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(((nb * 32 + n) * 576 + f * 16 + kh * 8) * 2), 0, 0);
    }
    for (int q = tid; q < W1B / 16; q += 256) {
        const int n = q & 31, kh = (q >> 5) & 1, nb = (q >> 6) & 1, f = q >> 7;
        *reinterpret_cast<u32x4*>(Ws1 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w1r, (unsigned)(((nb * 32 + n) * 256 + f * 16 + kh * 8) * 2), 0, 0);
    }
    // W3's fragments of this wave's 64 output channels (2 column blocks x 4 k steps), in registers for the whole launch
    u32x4 w3reg[8];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w3reg[nb * 4 + ks] = __builtin_amdgcn_raw_buffer_load_b128(w3r, (unsigned)((((2 * wave + nb) * 32 + fr) * CM + ks * 16 + fh * 8) * 2), 0, 0);

    // folded bn3 of this lane's 8 channels in the store layout, for both column blocks
    float sc3[2][8], sh3[2][8];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ch = wave * 64 + nb * 32 + (lane & 3) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc3[nb][e] = p.s3 ? p.s3[ch + e] : 1.f; sh3[nb][e] = p.b3 ? p.b3[ch + e] : 0.f; }
    }
    const int nbA = wave & 1, mb0 = wave >> 1;               // stage A: column block of this wave, its halo row blocks mb0, mb0+2, mb0+4
    float sc1[8], sh1[8];                                    // folded bn1 of this lane's 8 channels in the t1 store layout
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = nbA * 32 + (lane & 3) * 8 + e;
        sc1[e] = p.s1 ? p.s1[ch] : 1.f; sh1[e] = p.b1 ? p.b1[ch] : 0.f;
    }
    const float s2v0 = p.s2 ? p.s2[fr] : 1.f, b2v0 = p.b2 ? p.b2[fr] : 0.f, s2v1 = p.s2 ? p.s2[32 + fr] : 1.f, b2v1 = p.b2 ? p.b2[32 + fr] : 0.f;
    const unsigned char* const w1frag = Ws1 + (fh * 32 + fr) * 16 + nbA * 1024;
    const unsigned char* const w2frag = Ws2 + (fh * 32 + fr) * 16;
    const int py = 4 * wave + (fr >> 3), px = fr & 7;        // stage B: output pixel of this lane's A row inside the tile

    // stage A: the 192 halo rows x 256 channels of x as 12 UNITS (k group q = 4 k steps = 128 bytes of a pixel, row block j): a unit is four
    // 1-KiB loads (8 pixels x 128 B each), requested PFU units ahead; consumed through an LDS slot as four A fragments
    constexpr int NUNIT = 12, PFU = 6, NRING = PFU + 1;
    u32x4 fa[NRING][4];
    unsigned abase[3][4];
    int cy0 = 0, cx0 = 0, cb = 0;                            // tile whose x is in flight: origin and image
    const int lp = lane >> 3, lc = lane & 7;                 // load role: pixel lp of the piece, 16-byte chunk lc of its 128 bytes
    auto tile_origin = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        cb = t / p.tiles_y; cy0 = ty * BT_H; cx0 = tx * BT_W;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int P = (mb0 + 2 * j) * 32 + 8 * i + lp;
                const int hy = (P * 6554) >> 16, hx = P - hy * BH_W;          // P / 10 for P < 192
                const int iy = cy0 - 1 + hy, ix = cx0 - 1 + hx;
                const bool ok = P < NHALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                abase[j][i] = ok ? (unsigned)((((cb * p.H + iy) * p.W + ix) * CIO + lc * 8) * 2) : OOB;
            }
    };
    auto reqU = [&](int u) __attribute__((always_inline)) {     // unit u = (k group u / 3, row block u % 3)
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[u % NRING][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, abase[u % 3][i] + (unsigned)((u / 3) * 128), 0, 0);
    };
    // staging slot image: [32 pixels][128 B], chunk c of pixel r at position c ^ ((r >> 1) & 7) (conv_ring.hip's row image: conflict-free
    // ds_read_b128 fragment beats); a lane writes chunk lc of pixel 8 i + lp and reads, for k step ks, chunk 2 ks + fh of pixel fr
    int wofs[4], rofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + lp;
        wofs[i] = r * 128 + ((lc ^ ((r >> 1) & 7)) << 4);
        rofs[i] = fr * 128 + (((2 * i + fh) ^ ((fr >> 1) & 7)) << 4);
    }
    if ((int)blockIdx.x < ntiles) {
        tile_origin(blockIdx.x);
#pragma unroll
        for (int u = 0; u < PFU; ++u) reqU(u);
    }
    __syncthreads();                                         // filters are in LDS

    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int y0 = cy0, x0 = cx0, b = cb;
        // =================== stage A: t1 = relu(bn1(x . W1^T)) on the halo ===================
        f32x16 aacc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) aacc[j][r] = 0.f;
        SP_BSTAMP(ta0)
        u32x4 fcur[4], fnext[4];
        auto stage_unit = [&](int u, u32x4 (&dst)[4]) __attribute__((always_inline)) {   // registers -> the slot -> the unit's four A fragments
            unsigned char* const sl = xslot;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sl + wofs[i]) = fa[u % NRING][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const u32x4*>(sl + rofs[i]);
        };
        stage_unit(0, fcur);
#pragma unroll
        for (int u = 0; u < NUNIT; ++u) {
            if (u + PFU < NUNIT) reqU(u + PFU);
            if (u + 1 < NUNIT) stage_unit(u + 1, fnext);         // the next unit's fragments land under this unit's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 bbq = *reinterpret_cast<const u32x4*>(w1frag + ((u / 3) * 4 + ks) * 2048);
                aacc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fcur[ks]), __builtin_bit_cast(bf16x8, bbq), aacc[u % 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) fcur[i] = fnext[i];
        }
        SP_BSTAMP(ta1)
        SP_BACC(0, ta0, ta1)
        // t1 -> LDS (bf16; exactly zero outside the image) through the wave's transpose slab, so that a lane owns 8 consecutive channels of
        // one halo pixel: two 16-byte LDS stores per row block half instead of sixteen 2-byte ones per accumulator (a first form wrote
        // every accumulator register on its own: 48 x ~18 vector instructions, 3,500 cycles per tile)
        {
            auto put = [&](const f32x16& acc, int mb) __attribute__((always_inline)) {
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
                    const int P = mb * 32 + row;
                    const int hy = P / BH_W, hx = P - hy * BH_W;
                    const bool in = (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float q = v[e] * sc1[e] + sh1[e];
                        q = q > 0.f ? q : 0.f;
                        o8[e] = (__bf16)(in ? q : 0.f);
                    }
                    if (P < NHALO) *reinterpret_cast<u32x4*>(T1 + t1off(hy, hx, nbA * 4 + chunk)) = __builtin_bit_cast(u32x4, o8);
                }
            };
            put(aacc[0], mb0);
            put(aacc[1], mb0 + 2);
            put(aacc[2], mb0 + 4);
        }
        SP_BSTAMP(ta2)
        SP_BACC(1, ta1, ta2)
        __syncthreads();                                     // t1 complete
        SP_BSTAMP(tb0)
        SP_BACC(2, ta2, tb0)

        // the residual of this wave's 64 channels for all four 32-pixel blocks (16 loads) is requested now: it flies during stage B
        // store layout of a 32x32 accumulator: lane -> pixel row it*16 + (lane >> 2) of the block, 8 channels (lane & 3) * 8 of the 32
        u32x4 rv[4][2][2];                                   // [row block][column block][it]
        auto ooff = [&](int m, int it) __attribute__((always_inline)) {
            const int row = it * 16 + (lane >> 2);
            const int oy = y0 + 4 * m + (row >> 3), ox = x0 + (row & 7);
            return (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * CIO + wave * 64 + (lane & 3) * 8) * 2) : OOB;
        };
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const unsigned o = ooff(m, it);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) rv[m][nb][it] = __builtin_amdgcn_raw_buffer_load_b128(xr, o == OOB ? OOB : o + (unsigned)(nb * 64), 0, 0);
            }
        // =================== stage B: 3x3 from the LDS halo tile (the round-2 64-channel direct kernel's loop) ===================
        {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            constexpr int PF = 3, NSTEP = 36;
            u32x4 fx[PF + 1], fb0[PF + 1], fb1[PF + 1];
            auto frags = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 2, ks = st & 3;
                const int ay = py + tap / 3, ax = px + tap % 3;
                fx[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(T1 + t1off(ay, ax, ks * 2 + fh));
                const unsigned char* wb = w2frag + st * 2048;
                fb0[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(wb);
                fb1[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(wb + 1024);
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) frags(st);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + PF < NSTEP) frags(st + PF);
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fx[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb0[st % (PF + 1)]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fx[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb1[st % (PF + 1)]), acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            SP_BSTAMP(tb1)
            SP_BACC(3, tb0, tb1)
            // t2 -> rows 32 wave .. of the shared tile as bf16 [128 pixels][64 channels], 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
            unsigned char* const t2w = T2 + wave * 32 * PIXM;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int sw = (row >> 1) & 7;
                float v0 = acc0[r] * s2v0 + b2v0, v1 = acc1[r] * s2v1 + b2v1;
                v0 = v0 > 0.f ? v0 : 0.f;
                v1 = v1 > 0.f ? v1 : 0.f;
                *reinterpret_cast<__bf16*>(t2w + row * 128 + (((fr >> 3) ^ sw) << 4) + (fr & 7) * 2) = (__bf16)v0;
                *reinterpret_cast<__bf16*>(t2w + row * 128 + (((4 + (fr >> 3)) ^ sw) << 4) + (fr & 7) * 2) = (__bf16)v1;
            }
        }
        SP_BSTAMP(tb2)
        __syncthreads();                                     // t2 complete; every wave is done reading t1
        SP_BSTAMP(tc0)
        SP_BACC(2, tb2, tc0)

        // the next tile's x: its first PFU units fly during stage C
        if (tile + G < ntiles) {
            tile_origin(tile + G);
#pragma unroll
            for (int u = 0; u < PFU; ++u) reqU(u);
        }

        SP_BSTAMP(tc0b)
        SP_BACC(5, tc0, tc0b)
        // =================== stage C: y[:, 64 wave ..] = relu(bn3(t2 . W3^T) + x), one 32-pixel row block per pass ===================
        {
            auto pass = [&](auto mtag) __attribute__((always_inline)) {
                constexpr int m = decltype(mtag)::value;
                u32x4 a3[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a3[ks] = *reinterpret_cast<const u32x4*>(T2 + (m * 32 + fr) * 128 + (((ks * 2 + fh) ^ ((fr >> 1) & 7)) << 4));
                f32x16 c0, c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[ks]), __builtin_bit_cast(bf16x8, w3reg[ks]), c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[ks]), __builtin_bit_cast(bf16x8, w3reg[4 + ks]), c1, 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                        tr[row * 32 + fr] = nb == 0 ? c0[r] : c1[r];
                    }
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
                        for (int e = 0; e < 8; ++e) v[e] = v[e] * sc3[nb][e] + sh3[nb][e];
                        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[m][nb][it]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        bf16x8 o8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
                        const unsigned o = ooff(m, it);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, o == OOB ? OOB : o + (unsigned)(nb * 64), 0, 0);
                    }
                }
            };
            pass(std::integral_constant<int, 0>{});
            pass(std::integral_constant<int, 1>{});
            pass(std::integral_constant<int, 2>{});
            pass(std::integral_constant<int, 3>{});
        }
        SP_BSTAMP(tc1)
        SP_BACC(4, tc0b, tc1)
#ifdef SP_BNECK_DIAG
        dg[7] += 1;
#endif
    }
#ifdef SP_BNECK_DIAG
    {
        SP_BSTAMP(tk1)
        dg[6] = tk1 - tk0;
        if (lane == 0) {
            unsigned long long* o = sp_bneck_dbg + ((blockIdx.x & 255) * 8 + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = dg[i];
        }
    }
#endif
}

// ---- the same block with EIGHT waves, two per SIMD (round 6) -------------------------------------------------------------------------------
// The four-wave kernel above is bound by instruction issue and exposed latency at ONE wave per SIMD (27,400 cycles per tile for 4,900 cycles of
// MFMA work, profiles/r06_bneck_stages.txt): every LDS round trip of the transposes, every load issue and every barrier wait stalls the SIMD
// outright.  Same LDS image (W2, W1, t1, t2; the transpose slabs become eight HALF slabs of 16 rows), same tile, same accumulation chains
// and roundings - so the same bits - with the work of every stage cut eight ways instead of four, so that a second wave fills the stalls:
//   A  the six halo row blocks on waves 0 .. 5, BOTH column blocks each: an x fragment (straight from the NHWC rows; the 16 k steps of the
//      next tile are requested between the MFMAs of stage B) is loaded once and feeds two MFMAs - half the vector-memory instructions of a (row block, column block) split;
//   B  8 (row tile, column block) units: one accumulator per wave, 36 MFMAs;
//   C  wave w = output channels 32 w .. 32 w + 31 of all four 32-pixel row tiles (W3's 4 fragments in registers), residual requested
//      during stage B.
// 256 registers per wave are enough now: nobody holds more than two accumulators, and the x fragments of one row block are 64 registers.
constexpr int TRH = 16 * 32 * 4;                          // per wave: fp32 transpose half-slab (16 rows x 32 columns)
constexpr int LDSB8 = W2B + W1B + T1B + T2B + 8 * TRH;    // 162,304 B
static_assert(LDSB8 == LDSB, "both kernels use the same LDS image");

__global__ __launch_bounds__(512, 2) void bottleneck_c64_w8_kernel(const BneckArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smemb8[];
    unsigned char* const Ws2 = smemb8;
    unsigned char* const Ws1 = smemb8 + W2B;
    unsigned char* const T1 = smemb8 + W2B + W1B;
    unsigned char* const T2 = smemb8 + W2B + W1B + T1B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const tr = reinterpret_cast<float*>(smemb8 + W2B + W1B + T1B + T2B + wave * TRH);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;
    const int G = gridDim.x;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), (short)0, p.w3_bytes, 0x00020000);

    // the filters -> LDS in fragment order (piece q = f*128 + nb*64 + kh*32 + n holds W[nb*32 + n][f*16 + kh*8 .. + 8]).  Read in MEMORY order - consecutive
    // lanes take consecutive 16-byte pieces of a row - and scattered on the LDS side: read in fragment order every wave-instruction gathered 16 bytes from
    // each of 32 rows, the same 104 KB for all 512 workgroups (conv_block.hip's strip kernel lost 6 us of prologue to that pattern)
#ifndef SP_BNECK_GATHER_W
    for (int pi = tid; pi < W2B / 16; pi += 512) {
        const int r = pi / 72, c = pi - r * 72;                 // row (output channel), 16-byte column of the 1,152-byte row
        const int q = (c >> 1) * 128 + (r >> 5) * 64 + (c & 1) * 32 + (r & 31);
        *reinterpret_cast<u32x4*>(Ws2 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(pi * 16), 0, 0);
    }
    for (int pi = tid; pi < W1B / 16; pi += 512) {
        const int r = pi >> 5, c = pi & 31;                     // 512-byte rows
        const int q = (c >> 1) * 128 + (r >> 5) * 64 + (c & 1) * 32 + (r & 31);
        *reinterpret_cast<u32x4*>(Ws1 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w1r, (unsigned)(pi * 16), 0, 0);
    }
#else                                                           // (the round-6 A/B: tools/r06_bneck_wload.sh)
    for (int q = tid; q < W2B / 16; q += 512) {
        const int n = q & 31, kh = (q >> 5) & 1, nb = (q >> 6) & 1, f = q >> 7;
        *reinterpret_cast<u32x4*>(Ws2 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w2r, (unsigned)(((nb * 32 + n) * 576 + f * 16 + kh * 8) * 2), 0, 0);
    }
    for (int q = tid; q < W1B / 16; q += 512) {
        const int n = q & 31, kh = (q >> 5) & 1, nb = (q >> 6) & 1, f = q >> 7;
        *reinterpret_cast<u32x4*>(Ws1 + q * 16) = __builtin_amdgcn_raw_buffer_load_b128(w1r, (unsigned)(((nb * 32 + n) * 256 + f * 16 + kh * 8) * 2), 0, 0);
    }
#endif
    // stage C role: output channels 32 wave .. +31; W3's fragments of that column block (4 k steps) in registers for the whole launch
    u32x4 w3reg[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w3reg[ks] = __builtin_amdgcn_raw_buffer_load_b128(w3r, (unsigned)(((wave * 32 + fr) * CM + ks * 16 + fh * 8) * 2), 0, 0);
    // folded BatchNorms in the ACCUMULATOR layout (this lane's channel = column fr of the block): one (scale, shift) pair per block instead of
    // eight per lane in the store layout - the same fma per element, applied before the transposition instead of after it
    const float sc3 = p.s3 ? p.s3[wave * 32 + fr] : 1.f, sh3 = p.b3 ? p.b3[wave * 32 + fr] : 0.f;
    // stage A role: waves 0 .. 5 = halo row block `wave` x BOTH column blocks (an x fragment is loaded once and feeds two MFMAs); waves 6, 7 wait
    const bool hasA = wave < 6;                               // (wave-uniform)
    float sc1[2], sh1[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) { sc1[nb] = p.s1 ? p.s1[nb * 32 + fr] : 1.f; sh1[nb] = p.b1 ? p.b1[nb * 32 + fr] : 0.f; }
    const unsigned char* const w1frag = Ws1 + (fh * 32 + fr) * 16;       // + ks * 2048 + nb * 1024
    // stage B role: row tile mB x column block nbB
    const int mB = wave >> 1, nbB = wave & 1;
    const float s2v = p.s2 ? p.s2[nbB * 32 + fr] : 1.f, b2v = p.b2 ? p.b2[nbB * 32 + fr] : 0.f;
    const unsigned char* const w2frag = Ws2 + (fh * 32 + fr) * 16 + nbB * 1024;
    const int py = 4 * mB + (fr >> 3), px = fr & 7;

    // x of the NEXT tile: all 16 k steps are requested during stage B of the current one, one load per two MFMAs - a load costs ~100 cycles of
    // issue, free between MFMAs and paid in full anywhere else (requested in one burst before stage C: 4 % slower; requested inside stage A: needed
    // 0.25 us later when HBM answers in 1-2: profiles/r06_bneck8_ab.txt)
    constexpr int NKA = 16;
    u32x4 fa[NKA];
    unsigned abase = OOB;
    int cy0 = 0, cx0 = 0, cb = 0;
    auto tile_origin = [&](int tile) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        cb = t / p.tiles_y; cy0 = ty * BT_H; cx0 = tx * BT_W;
        {
            const int P = wave * 32 + fr;
            const int hy = (P * 6554) >> 16, hx = P - hy * BH_W;          // P / 10 for P < 256
            const int iy = cy0 - 1 + hy, ix = cx0 - 1 + hx;
            const bool ok = hasA && P < NHALO && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            abase = ok ? (unsigned)((((cb * p.H + iy) * p.W + ix) * CIO + fh * 8) * 2) : OOB;
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 3   // DIAGNOSTIC BUILD ONLY (tools/r06_bneck_knockout.sh): no x loads
            abase = OOB;
#endif
        }
    };
    auto reqA = [&](int ks) __attribute__((always_inline)) {
        if (hasA) fa[ks] = __builtin_amdgcn_raw_buffer_load_b128(xr, abase + (unsigned)(ks * 32), 0, 0);
    };
    if ((int)blockIdx.x < ntiles) {
        tile_origin(blockIdx.x);
#pragma unroll
        for (int ks = 0; ks < NKA; ++ks) reqA(ks);
    }
    __syncthreads();                                         // filters are in LDS
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int y0 = cy0, x0 = cx0, b = cb;
        // an opaque zero: address arithmetic that depends on it is redone per tile instead of being hoisted out of the persistent loop, where its
        // 36 + 16 loop-invariant fragment addresses would be live across every stage (the register cap is 256 at two waves per SIMD)
        int zt = 0;
        asm volatile("" : "+v"(zt));
        const int pyt = py + zt, pxt = px + zt, frt = fr + zt;
        // =================== stage A ===================
        f32x16 a0, a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 6   // no stage A at all (t1 stays what it was)
        if (false) {
#else
        if (hasA) {
#endif
#pragma unroll
            for (int ks = 0; ks < NKA; ++ks) {
                const u32x4 b0q = *reinterpret_cast<const u32x4*>(w1frag + ks * 2048);
                const u32x4 b1q = *reinterpret_cast<const u32x4*>(w1frag + ks * 2048 + 1024);
                const bf16x8 xa = __builtin_bit_cast(bf16x8, fa[ks]);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b0q), a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b1q), a1, 0, 0, 0);
            }
            auto put = [&](const f32x16& acc, int rb, float scv, float shv, int nbA) __attribute__((always_inline)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float t = acc[8 * h + q] * scv + shv;
                        tr[((q & 3) + 8 * (q >> 2) + 4 * fh) * 32 + fr] = t > 0.f ? t : 0.f;
                    }
                    const int rl = lane >> 2, chunk = lane & 3;
                    float v[8];
#pragma unroll
                    for (int e4 = 0; e4 < 2; ++e4) {
                        const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + rl * 32 + chunk * 8 + 4 * e4);
                        v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                    }
                    const int P = rb * 32 + h * 16 + rl;
                    const int hy = (P * 6554) >> 16, hx = P - hy * BH_W;
                    const bool in = (unsigned)(y0 - 1 + hy) < (unsigned)p.H && (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
                    bf16x8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (__bf16)(in ? v[e] : 0.f);
                    if (P < NHALO) *reinterpret_cast<u32x4*>(T1 + t1off(hy, hx, nbA * 4 + chunk)) = __builtin_bit_cast(u32x4, o8);
                }
            };
            put(a0, wave, sc1[0], sh1[0], 0);
            put(a1, wave, sc1[1], sh1[1], 1);
        }
        __syncthreads();                                     // t1 complete

        // the residual of this wave's 32 channels for all four 32-pixel row tiles (8 loads): it flies during stage B
        u32x4 rv[4][2];
        auto ooff = [&](int m, int it) __attribute__((always_inline)) {
            const int row = it * 16 + (lane >> 2);
            const int oy = y0 + 4 * m + (row >> 3), ox = x0 + (row & 7);
            return (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * CIO + wave * 32 + (lane & 3) * 8) * 2) : OOB;
        };
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 2   // no residual loads
                rv[m][it] = u32x4{0u, 0u, 0u, 0u};
#else
                rv[m][it] = __builtin_amdgcn_raw_buffer_load_b128(xr, ooff(m, it), 0, 0);
#endif
        // =================== stage B ===================
        const bool more = tile + G < ntiles;
        if (more) tile_origin(tile + G);                      // (y0 / x0 / b of THIS tile were copied above)
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int PF = 3, NSTEP = 36;
            u32x4 fx[PF + 1], fb[PF + 1];
            auto frags = [&](int st) __attribute__((always_inline)) {
                const int tap = st >> 2, ks = st & 3;
                fx[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(T1 + t1off(pyt + tap / 3, pxt + tap % 3, ks * 2 + fh));
                fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(w2frag + zt + st * 2048);
            };
#pragma unroll
            for (int st = 0; st < PF; ++st) frags(st);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + PF < NSTEP) frags(st + PF);
                if (more && (st & 1) == 0 && st / 2 < NKA) reqA(st / 2);
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 4   // stage B: one MFMA in four (the fragment reads stay)
                if ((st & 3) == 0)
#endif
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fx[st % (PF + 1)]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
            }
            unsigned char* const t2w = T2 + mB * 32 * PIXM;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int sw = (row >> 1) & 7;
                float v0 = acc[r] * s2v + b2v;
                v0 = v0 > 0.f ? v0 : 0.f;
                *reinterpret_cast<__bf16*>(t2w + row * 128 + (((nbB * 4 + (fr >> 3)) ^ sw) << 4) + (fr & 7) * 2) = (__bf16)v0;
            }
        }
        __syncthreads();                                     // t2 complete; every wave is done reading t1

        // =================== stage C ===================
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            u32x4 a3[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a3[ks] = *reinterpret_cast<const u32x4*>(T2 + (m * 32 + frt) * 128 + (((ks * 2 + fh) ^ ((frt >> 1) & 7)) << 4));
            f32x16 c0;
#pragma unroll
            for (int r = 0; r < 16; ++r) c0[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[ks]), __builtin_bit_cast(bf16x8, w3reg[ks]), c0, 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int rl = lane >> 2, chunk = lane & 3;
                float v[8];
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 5   // stage C: no LDS transposition (wrong values in the right places)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = c0[8 * it + e];
#else
#pragma unroll
                for (int q = 0; q < 8; ++q) tr[((q & 3) + 8 * (q >> 2) + 4 * fh) * 32 + fr] = c0[8 * it + q] * sc3 + sh3;
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const f32x4 tt = *reinterpret_cast<const f32x4*>(tr + rl * 32 + chunk * 8 + 4 * e4);
                    v[4 * e4] = tt[0]; v[4 * e4 + 1] = tt[1]; v[4 * e4 + 2] = tt[2]; v[4 * e4 + 3] = tt[3];
                }
#endif
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[m][it]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)r8[e];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                bf16x8 o8;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] = (__bf16)v[e];
#if defined(SP_BNECK_KNOCKOUT) && SP_BNECK_KNOCKOUT == 1   // no y stores (one lane of one workgroup still stores, so that nothing is optimised away)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, (tile == 0 && tid == 0) ? ooff(m, it) : OOB, 0, 0);
#else
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o8), yr, ooff(m, it), 0, 0);
#endif
            }
        }
    }
}

bool bneck_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;
    // `d`: the block's 3x3 convolution (64 -> 64, stride 1, pad 1) on bf16 NHWC; conv1 / conv3 are 1x1 on the same grid
    return d && (d->flags & SP_CONV_BF16) && !(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) && d->c_in == CM &&
           d->c_out == CM && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) && d->dy0 == -1 &&
           d->dx0 == -1 && d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == 576 && d->n_pad >= CM &&
           d->grid_h == d->in_h && d->grid_w == d->in_w;
}

}  // namespace

extern "C" int sp_bottleneck_c64_ok(const sp_conv_desc* d) { return bneck_ok(d) ? 1 : 0; }

extern "C" int sp_bottleneck_c64(const sp_conv_desc* d, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                                 const void* w2_packed, const float* scale2, const float* shift2, const void* w3_packed, const float* scale3,
                                 const float* shift3, void* y, void* stream) {
    SP_REQUIRE(d && x && w1_packed && w2_packed && w3_packed && y, "sp_bottleneck_c64: null pointer");
    SP_REQUIRE(bneck_ok(d), "sp_bottleneck_c64: `desc` must describe the block's bf16 3x3 stride-1 pad-1 convolution with 64 -> 64 channels");
    SP_REQUIRE(d->batch > 0 && x != y, "sp_bottleneck_c64: bad batch / y must not alias x");
    const long long elems = (long long)d->batch * d->in_h * d->in_w * CIO;
    SP_REQUIRE(elems < (1ll << 30), "sp_bottleneck_c64: tensor too large");
    static const int w8 = [] { const char* e = getenv("SP_BNECK_W8"); return e ? atoi(e) : 1; }();     // (0: the four-wave kernel, for same-box A/Bs)
    if (sp_name_query_active()) { sp_name_query_set(w8 ? "bottleneck_c64_w8_kernel" : "bottleneck_c64_kernel"); return SP_OK; }
    BneckArgs a;
    a.x = x; a.y = y; a.w1 = w1_packed; a.w2 = w2_packed; a.w3 = w3_packed;
    a.s1 = scale1; a.b1 = shift1; a.s2 = scale2; a.b2 = shift2; a.s3 = scale3; a.b3 = shift3;
    a.H = d->in_h; a.W = d->in_w; a.batch = d->batch;
    a.tiles_x = (d->in_w + BT_W - 1) / BT_W; a.tiles_y = (d->in_h + BT_H - 1) / BT_H;
    a.x_bytes = (int)(elems * 2);
    a.w1_bytes = CM * CIO * 2; a.w2_bytes = CM * 576 * 2; a.w3_bytes = CIO * CM * 2;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_bottleneck_c64: too many tiles");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (e != hipSuccess) { sp_set_error("sp_bottleneck_c64: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDSB, hipGetErrorString(e)); return SP_ELAUNCH; }
    const long long grid = tiles < cus ? tiles : cus;
    if (w8) {
        const hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_c64_w8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB8);
        if (e8 != hipSuccess) { sp_set_error("sp_bottleneck_c64: hipFuncSetAttribute(max dynamic LDS = %d) failed: %s", LDSB8, hipGetErrorString(e8)); return SP_ELAUNCH; }
        hipLaunchKernelGGL(bottleneck_c64_w8_kernel, dim3((unsigned)grid), dim3(512), LDSB8, (hipStream_t)stream, a);
        return sp_check_launch("bottleneck_c64_w8_kernel");
    }
    hipLaunchKernelGGL(bottleneck_c64_kernel, dim3((unsigned)grid), dim3(256), LDSB, (hipStream_t)stream, a);
    return sp_check_launch("bottleneck_c64_kernel");
}
