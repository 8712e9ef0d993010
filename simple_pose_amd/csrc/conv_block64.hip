// conv_block64.hip - one HRNet BasicBlock (nets/pose_hrnet.py:34-51) of the 64-channel branch as ONE launch, bf16:
//
//     out = relu(bn2(conv3x3(relu(bn1(conv3x3(x))))) + x)
//
// The block's two convolutions run one by one on conv3x3_c64_tile_kernel (conv_direct.hip): at bs=128 a 32 x 24 map is ONE round of 256 tiles, a
// chain load -> compute -> store of 15.8 us per launch for 7.2 GFLOP and 38 MB, twice per block, with the intermediate t written and read back.
// conv_block.hip's strip kernel does not carry over as it is: the two 64 x 576 filters are 2 x 72 KB - neither the LDS (beside the images) nor one
// wave's registers (2 x 2 x 144 VGPRs) hold them.  So the eight waves take ROLES:
//
//   waves 0 .. 3   conv1: wave r = (output-channel half nh = r & 1, row-block parity r >> 1) keeps W1[32 nh .. + 32][576] in 144 VGPRs
//   waves 4 .. 7   conv2: the same split of W2
//
// and the block is software-pipelined over 4 x 24-pixel strips: in phase j the conv1 waves turn the halo of strip j (8 x 28 pixels, LDS buffer j & 1)
// into t of strip j (6 x 26 pixels, zero outside the image, bf16, LDS buffer j & 1) while the conv2 waves turn t of strip j - 1 into y, and the LDS-DMA
// pieces of strip j + 1 land in the other halo buffer; one barrier per phase.  Each SIMD holds one conv1 and one conv2 wave with 3 + 1 or 2 + 2 row
// blocks of 36 MFMAs per phase.  As in conv_block.hip the MFMA operands are swapped (filter = A, pixels = B: a lane's accumulator is one pixel x 16
// channels - t and y leave in NHWC order without a transpose), the LDS images are 16-byte-chunk planes (a tap is the row block's base + an immediate),
// and the chains are the two launches' chains (tap-major, channel-minor, same bf16 rounding of t, same epilogue arithmetic): bit-identical, which is
// how it is tested.  The residual is the centre of strip j - 1's halo, in L2 since one phase: conv2 reads it from memory (its LDS buffer is the one the
// next halo lands in).
#include "sp_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int C = 64, KP = 576;                             // channels; K = 9 taps x 64 (the packed rows' length: k_pad)
constexpr int TH = 4, TW = 24;                              // output strip
constexpr int TTW = TW + 2, NT = (TH + 2) * TTW;           // t tile: 6 x 26 = 156 pixels = 5 row blocks of 32 (4 slots idle)
constexpr int XH = TH + 4, XW = TW + 4, NX = XH * XW;     // x halo: 8 x 28 = 224 pixels
constexpr int XPLANE = 256 * 16, TPLANE = 160 * 16;        // planes padded to whole LDS-DMA pieces / row blocks
constexpr int X_BYTES = 8 * XPLANE, T_BYTES = 8 * TPLANE;  // 32 KB, 20 KB
// LDS: table | halo buffer 0 | t buffer 0 | t buffer 1 | halo buffer 1 = 1 + 32 + 40 + 32 = 105 KB
constexpr int P_TAB = 0, P_X0 = 1024, P_T = P_X0 + X_BYTES, P_X1 = P_T + 2 * T_BYTES, LDS_BYTES = P_X1 + X_BYTES;
constexpr int XSTRIDE = P_X1 - P_X0;
constexpr int WSTAGE = 72 * 64 * 16;                        // one filter in fragment order: [16-byte column][row][16 B] = 72 KB, staged in [t buffers, halo buffer 1]
static_assert(WSTAGE <= 2 * T_BYTES + X_BYTES, "the staging area must leave halo buffer 0 alone");
constexpr int PPW = 8;                                      // LDS-DMA pieces per conv2 wave and strip (8 planes x 4 pieces over the four conv2 waves)
constexpr unsigned OOB = 0x80000000u;

struct Block64Args {
    const void* x;
    const void* w1; const float* scale1; const float* shift1;
    const void* w2; const float* scale2; const float* shift2;
    void* y;
    int H, W, batch, tiles_x, tiles_y;
    int x_bytes, w_bytes;
};

// one LDS-DMA piece (conv_ring.hip dma16: inline asm on purpose, see there): 64 lanes x 16 bytes, lane l's bytes from rsrc + voff (zeros when out
// of range) to LDS at lds_addr + 16 l
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc)
                 : "memory");
}

__global__ __launch_bounds__(512) void basic_block_c64_kernel(const Block64Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const tab = reinterpret_cast<float*>(smem + P_TAB);         // scale1[64], shift1[64], scale2[64], shift2[64]
    unsigned char* const Tb = smem + P_T;
    unsigned char* const Xs = smem + P_X0;                            // halo buffer b at + b * XSTRIDE
    const unsigned xs_lds = (unsigned)(size_t)(lds_void_t*)Xs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const bool conv2 = wave >= 4;                           // (wave-uniform) this wave's role
    const int nh = wave & 1, par = (wave >> 1) & 1;         // its output-channel half and row-block parity
    const int ntiles = p.tiles_x * p.tiles_y * p.batch;
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = blockIdx.x * per;
    const int tile_end = tile0 + per < ntiles ? tile0 + per : ntiles;
    if (tile0 >= ntiles) return;                           // (the whole workgroup)
    const int n = tile_end - tile0;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), (short)0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, (short)0, p.x_bytes, 0x00020000);

    // a strip's halo arrives by LDS-DMA, issued by the conv2 waves (they have 1 - 2 row blocks per phase against the conv1 waves' 2 - 3): piece id =
    // 8 (wave - 4) + i covers pixels (id % 4) * 64 + lane of chunk plane id / 4
    auto request = [&](int tile, int buf) __attribute__((always_inline)) {
        int lz = lane;                                       // (opaque: keeps the per-piece pixel arithmetic out of registers between phases)
        asm volatile("" : "+v"(lz));
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y;
        const int b = t / p.tiles_y;
        const int y0 = ty * TH - 2, x0 = tx * TW - 2;
        const int base = ((b * p.H + y0) * p.W + x0) * (C * 2);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int id = (wave - 4) * PPW + i;
            const int P = (id % 4) * 64 + lz;                 // pixel of the halo (row hy, column hx); the planes' padding (P >= NX) receives zeros
            const int hy = P / XW, hx = P - hy * XW;
            const int iy = y0 + hy, ix = x0 + hx;
            const bool ok = tile < tile_end && P < NX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            dma16(xs_lds + (unsigned)(buf * XSTRIDE + (id / 4) * XPLANE + (id % 4) * 1024),
                  ok ? (unsigned)(base + (hy * p.W + hx) * (C * 2) + (id / 4) * 16) : OOB, xr);
        }
    };
    if (conv2) request(tile0, 0);

    // the filters -> registers through LDS, one after the other (72 KB each, read in memory order: the 64 rows of 1,152 B are 4,608 consecutive 16-byte
    // pieces, nine per thread; staged in fragment order [16-byte column][row] in the t buffers + halo buffer 1, which nothing uses yet): fragment f =
    // tap * 4 + ks of this wave is W[32 nh + lane % 32][f * 16 + (lane / 32) * 8 .. + 8], the MFMA's A operand
    u32x4 wf[36];
    if (tid < 256) {
        const int k = tid >> 6, c = tid & 63;
        const float* src = k == 0 ? p.scale1 : (k == 1 ? p.shift1 : (k == 2 ? p.scale2 : p.shift2));
        tab[tid] = src ? src[c] : ((k & 1) ? 0.f : 1.f);
    }
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        u32x4 wv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wv[i] = __builtin_amdgcn_raw_buffer_load_b128(round ? w2r : w1r, (unsigned)((tid + 512 * i) * 16), 0, 0);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + 512 * i;
            const int r = q / 72, cidx = q - r * 72;          // row (output channel), 16-byte column
            *reinterpret_cast<u32x4*>(Tb + (cidx * 64 + r) * 16) = wv[i];
        }
        __syncthreads();
        if ((round == 1) == conv2) {
#pragma unroll
            for (int f = 0; f < 36; ++f) wf[f] = *reinterpret_cast<const u32x4*>(Tb + ((2 * f + fh) * 64 + 32 * nh + fr) * 16);
        }
        __syncthreads();                                      // every wave has its fragments of this filter: the staging area is free again
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the first strip's halo
    __syncthreads();                                        // the first halo is complete

#pragma unroll 1
    for (int j = 0; j <= n; ++j) {
        // the next strip's halo -> the buffer conv1 read a phase ago
        if (conv2 && j + 1 < n) request(tile0 + j + 1, (j + 1) & 1);
        int frz = fr;                                        // (opaque, as above: the row blocks' pixel arithmetic is redone per phase)
        asm volatile("" : "+v"(frz));
        if (!conv2) {
            // ---- conv1 of strip j: halo buffer j & 1 -> t buffer j & 1; this wave's row blocks par, par + 2 (, par + 4), channels 32 nh .. + 32 ----
            if (j < n) {
                const unsigned char* const X = Xs + (j & 1) * XSTRIDE;
                unsigned char* const T = Tb + (j & 1) * T_BYTES;
                int t_ = tile0 + j;
                const int tlx = t_ % p.tiles_x; t_ /= p.tiles_x;
                const int tly = t_ % p.tiles_y;
#pragma unroll 1
                for (int blk = par; blk < 5; blk += 2) {
                    const int q = blk * 32 + frz;
                    const int qq = q < NT ? q : NT - 1;         // idle slots compute a duplicate and are not stored
                    const int ty = qq / TTW, tx = qq - ty * TTW;
                    const unsigned char* const xb = X + fh * XPLANE + (ty * XW + tx) * 16;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    constexpr int PF = 3;
                    u32x4 fb[PF + 1];
                    auto frag = [&](int st) __attribute__((always_inline)) {
                        const int tap = st >> 2, ks = st & 3;
                        fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(xb + ks * 2 * XPLANE + ((tap / 3) * XW + tap % 3) * 16);
                    };
#pragma unroll
                    for (int st = 0; st < PF; ++st) frag(st);
                    f32x4 scA[2], shA[2];                   // scale / shift of channel groups 0, 1 (requested before the last MFMAs; groups 2, 3 follow after the loop)
#pragma unroll
                    for (int st = 0; st < 36; ++st) {
                        if (st + PF < 36) frag(st + PF);
                        if (st == 32) {
#pragma unroll
                            for (int g = 0; g < 2; ++g) {
                                scA[g] = *reinterpret_cast<const f32x4*>(tab + 32 * nh + 8 * g + 4 * fh);
                                shA[g] = *reinterpret_cast<const f32x4*>(tab + 64 + 32 * nh + 8 * g + 4 * fh);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[st]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // epilogue 1: t = relu(acc * scale1 + shift1) as bf16 into the t buffer - ZERO outside the image (conv2 pads t, not x).
                    // acc[4g + j] = channel 32 nh + 8g + 4 fh + j of this lane's pixel = 8 bytes at + fh * 8 of chunk plane 4 nh + g
                    f32x4 scB[2], shB[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        scB[g] = *reinterpret_cast<const f32x4*>(tab + 32 * nh + 8 * (g + 2) + 4 * fh);
                        shB[g] = *reinterpret_cast<const f32x4*>(tab + 64 + 32 * nh + 8 * (g + 2) + 4 * fh);
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
                        if (q < NT) *reinterpret_cast<u32x2*>(T + (4 * nh + g) * TPLANE + q * 16 + fh * 8) = o;
                    }
                }
            }
        } else {
            // ---- conv2 of strip j - 1: t buffer (j - 1) & 1 -> y; row blocks {1} (parity 0: its SIMD's conv1 wave has three) or {0, 2} ----
            if (j >= 1) {
                const unsigned char* const T = Tb + ((j - 1) & 1) * T_BYTES;
                int t_ = tile0 + j - 1;
                const int tlx = t_ % p.tiles_x; t_ /= p.tiles_x;
                const int tly = t_ % p.tiles_y;
                const int b = t_ / p.tiles_y;
#pragma unroll 1
                for (int ob = par ? 0 : 1; ob < 3; ob += 2) {
                    const int o = ob * 32 + frz;
                    const int oyl = o / TW, oxl = o - oyl * TW;
                    const int oy = tly * TH + oyl, ox = tlx * TW + oxl;
                    const unsigned pix = (oy < p.H && ox < p.W) ? (unsigned)((((b * p.H + oy) * p.W + ox) * C + 32 * nh) * 2) : OOB;
                    // the residual in the accumulator's order: channels 32 nh + 8g + 4 fh .. + 4 of this lane's pixel
                    u32x2 rr[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) rr[g] = __builtin_amdgcn_raw_buffer_load_b64(xr, pix == OOB ? OOB : pix + (unsigned)(g * 16 + fh * 8), 0, 0);
                    const unsigned char* const tb = T + fh * TPLANE + (oyl * TTW + oxl) * 16;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    constexpr int PF = 3;
                    u32x4 fb[PF + 1];
                    auto frag = [&](int st) __attribute__((always_inline)) {
                        const int tap = st >> 2, ks = st & 3;
                        fb[st % (PF + 1)] = *reinterpret_cast<const u32x4*>(tb + ks * 2 * TPLANE + ((tap / 3) * TTW + tap % 3) * 16);
                    };
#pragma unroll
                    for (int st = 0; st < PF; ++st) frag(st);
                    f32x4 scA[2], shA[2];
#pragma unroll
                    for (int st = 0; st < 36; ++st) {
                        if (st + PF < 36) frag(st + PF);
                        if (st == 32) {
#pragma unroll
                            for (int g = 0; g < 2; ++g) {
                                scA[g] = *reinterpret_cast<const f32x4*>(tab + 128 + 32 * nh + 8 * g + 4 * fh);
                                shA[g] = *reinterpret_cast<const f32x4*>(tab + 192 + 32 * nh + 8 * g + 4 * fh);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[st]), __builtin_bit_cast(bf16x8, fb[st % (PF + 1)]), acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // epilogue 2: out = relu(acc * scale2 + shift2 + x)
                    f32x4 scB[2], shB[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        scB[g] = *reinterpret_cast<const f32x4*>(tab + 128 + 32 * nh + 8 * (g + 2) + 4 * fh);
                        shB[g] = *reinterpret_cast<const f32x4*>(tab + 192 + 32 * nh + 8 * (g + 2) + 4 * fh);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned d[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 sc = g < 2 ? scA[g & 1] : scB[g & 1];
                        const f32x4 sh = g < 2 ? shA[g & 1] : shB[g & 1];
                        const bf16x4 r4 = __builtin_bit_cast(bf16x4, rr[g]);
                        f32x2 v01, v23;
                        v01[0] = acc[4 * g + 0] * sc[0] + sh[0]; v01[1] = acc[4 * g + 1] * sc[1] + sh[1];
                        v23[0] = acc[4 * g + 2] * sc[2] + sh[2]; v23[1] = acc[4 * g + 3] * sc[3] + sh[3];
                        v01[0] = __builtin_fmaxf(v01[0] + (float)r4[0], 0.f); v01[1] = __builtin_fmaxf(v01[1] + (float)r4[1], 0.f);
                        v23[0] = __builtin_fmaxf(v23[0] + (float)r4[2], 0.f); v23[1] = __builtin_fmaxf(v23[1] + (float)r4[3], 0.f);
                        d[g][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2));
                        d[g][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2));
                    }
                    // lanes l and l + 32 hold the two halves of each 16-byte chunk of pixel l: swap so that every lane holds whole chunks
                    // (lanes < 32: chunks 0 and 2 of this half of the channels, lanes >= 32: chunks 1 and 3)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(d[2 * pr][0], d[2 * pr + 1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(d[2 * pr][1], d[2 * pr + 1][1], false, false);
                        u32x4 o4;
                        o4[0] = s0[0]; o4[1] = s1[0]; o4[2] = s0[1]; o4[3] = s1[1];
                        __builtin_amdgcn_raw_buffer_store_b128(o4, yr, pix == OOB ? OOB : pix + (unsigned)((2 * pr + fh) * 16), 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of strip j + 1's halo have landed
        __syncthreads();            // t of strip j and the halo of strip j + 1 are complete; every wave is done with t of strip j - 1 and the halo of strip j
    }
}

bool block64_ok(const sp_conv_desc* d) {
    if (d && d->c_in_group > 0) return false;
    return d && (d->flags & SP_CONV_BF16) && !(d->flags & (SP_CONV_OUT_NCHW | SP_CONV_PIXEL_SHUFFLE | SP_CONV_OUT_F32)) && d->c_in == 64 &&
           d->c_out == 64 && d->out_c == 64 && d->taps_h == 3 && d->taps_w == 3 && d->stride == 1 && (d->stride_x == 0 || d->stride_x == 1) &&
           d->dy0 == -1 && d->dx0 == -1 && d->dy_step == 1 && d->dx_step == 1 && d->phases_y == 1 && d->phases_x == 1 && d->k_pad == KP &&
           d->n_pad >= 64 && d->grid_h == d->in_h && d->grid_w == d->in_w && d->out_h == d->in_h && d->out_w == d->in_w && d->oy_mul == 1 &&
           d->ox_mul == 1 && d->oy_add == 0 && d->ox_add == 0;
}

}  // namespace

extern "C" int sp_basic_block_c64_ok(const sp_conv_desc* d) { return block64_ok(d) ? 1 : 0; }

extern "C" int sp_basic_block_c64(const sp_conv_desc* d, const void* x, const void* w1_packed, const float* scale1, const float* shift1,
                                  const void* w2_packed, const float* scale2, const float* shift2, void* y, void* stream) {
    SP_REQUIRE(d && x && w1_packed && w2_packed && y, "sp_basic_block_c64: null pointer");
    SP_REQUIRE(block64_ok(d), "sp_basic_block_c64: `desc` must describe the block's bf16 3x3 stride-1 pad-1 convolutions with 64 -> 64 channels");
    SP_REQUIRE(x != y, "sp_basic_block_c64: the output must not alias the input (neighbouring strips read the input's halo)");
    SP_REQUIRE(d->batch > 0, "sp_basic_block_c64: bad batch");
    if (sp_name_query_active()) { sp_name_query_set("basic_block_c64_kernel"); return SP_OK; }
    const long long elems = (long long)d->batch * d->in_h * d->in_w * C;
    SP_REQUIRE(elems < (1ll << 29) && d->in_h < (1 << 20) && d->in_w < (1 << 20), "sp_basic_block_c64: tensor too large");
    Block64Args a;
    a.x = x; a.w1 = w1_packed; a.scale1 = scale1; a.shift1 = shift1; a.w2 = w2_packed; a.scale2 = scale2; a.shift2 = shift2; a.y = y;
    a.H = d->in_h; a.W = d->in_w; a.batch = d->batch;
    a.tiles_x = (d->in_w + TW - 1) / TW; a.tiles_y = (d->in_h + TH - 1) / TH;
    a.x_bytes = (int)(elems * 2); a.w_bytes = d->n_pad * d->k_pad * 2;
    const long long tiles = (long long)d->batch * a.tiles_x * a.tiles_y;
    SP_REQUIRE(tiles < (1ll << 31), "sp_basic_block_c64: too many tiles");
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&basic_block_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    SP_REQUIRE(attr == hipSuccess, "sp_basic_block_c64: cannot reserve %d bytes of LDS (%s)", LDS_BYTES, hipGetErrorString(attr));
    // persistent, one workgroup per CU; consecutive strips per workgroup, and as many workgroups as keeps the longest share minimal
    const long long per = (tiles + 255) / 256;
    const long long grid = (tiles + per - 1) / per;
    hipLaunchKernelGGL(basic_block_c64_kernel, dim3((unsigned)grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return sp_check_launch("basic_block_c64_kernel");
}
