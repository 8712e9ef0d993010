// conv_stem.hip - the ResNet stem as ONE launch: fp32 NCHW image -> conv 7x7 s2 p3 (3 -> 64) -> BatchNorm (eval) -> ReLU ->
// MaxPool 3x3 s2 p1 -> pooled NHWC activation (nets/pose_resnet_dconv.py:158-162 `x = self.maxpool(self.relu(self.bn1(self.conv1(x))))`).
//
// Why: as three launches (layout change, implicit GEMM, pooling) the stem writes the 128 x 96 x 64 map of every image to HBM and reads
// it back (bs = 128: 201 MB bf16 / 403 MB fp32 each way) and the implicit GEMM pads K = 147 to 256 (bf16) / 224 (fp32).  Here a
// persistent 4-wave workgroup owns an 8 x 12 tile of POOLED pixels: the 39 x 56 image patch it needs goes to LDS once (bf16: as NHWC4
// pixels, fp32: as three planes), the 17 x 25 conv outputs behind the tile (17 x 26 GEMM rows = 14 MFMA row tiles) are formed with A
// fragments read straight from the patch - no im2col copy - and B fragments (all of the packed stem weights this wave multiplies by)
// held in registers for the life of the workgroup; BatchNorm + ReLU run on the accumulators, the conv tile is parked in LDS channel-major
// and the pooling pass reads it back with one lane per channel.  HBM sees the image (1.3x for the halo, mostly L2 hits) and the
// pooled output only.  The next tile's patch is requested before the MFMAs of the current one and lands in registers behind them.
//
// Bits: identical to sp_nchw_to_nhwc4[_bf16] -> sp_conv2d_fwd(conv1) -> sp_maxpool3x3s2_nhwc[_bf16].  The packed weights are the
// implicit GEMM's own ([64][k_pad], K ordered (ky, x slot 0..7, channel 0..3)), every MFMA gets the same k positions in the same
// order, and only instructions whose weights are all padding are dropped (they add 0.0f * finite = 0 to an accumulator):
//   bf16  v_mfma_f32_32x32x16_bf16, k-step j = 16 k = x slots 4 (j & 1) .. + 3 of row ky = j >> 1; x slot q is pixel 2 ox - 4 + q
//         (the GEMM's "pixel pair" view of the NHWC4 image, engine.py ProgramBuilder.conv); 14 k-steps, K 224..255 dropped
//   fp32  v_mfma_f32_32x32x2_f32, the GEMM issues k and k + 4 of every 8-k group together: (ky, x slot pair pp, channel s) with lanes
//         0-31 on slot 2 pp and lanes 32-63 on slot 2 pp + 1; x slot q is pixel 2 ox - 3 + q; channel 3 is padding -> 84 of 112
//         instructions remain.
#include <stdlib.h>

#include <type_traits>

#include "sp_common.h"

#ifdef SP_STEM_DIAG
// DIAGNOSTIC BUILD ONLY (tools/diag_stem.py; never the shipped library): per-wave cycle sums of the phases, read back with
// sp_stem_debug_read().  [block % 512][wave][8]: 0 park + barrier, 1 next patch requests, 2 GEMM rows (MFMA + BatchNorm + conv tile store),
// 3 barrier, 4 pooling, 6 kernel lifetime, 7 tiles
__device__ unsigned long long sp_stem_dbg[512 * 4 * 8];
extern "C" int sp_stem_debug_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sp_stem_dbg), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#define SP_SSTAMP(var)                                                                      \
    unsigned long long var;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);
#define SP_SACC(slot, a, b) dg[slot] += (b) - (a);
#else
#define SP_SSTAMP(var)
#define SP_SACC(slot, a, b)
#endif
#ifdef SP_STEM_NOSTORE                     // diagnostic experiment: the pooled results are computed but (practically) never stored
#define SP_STEM_STORE_COND(o) && (o) == 0x7fc12345u
#else
#define SP_STEM_STORE_COND(o)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct StemArgs {
    const void* x;             // fp32 [B][3][H][W], or (U8) BGR bytes [B][H][W][3] normalised on the fly: (v / 255 - mean) as datasets/coco.py:136
    float mean[3];             // U8: RGB means
    const void* w;             // packed stem weights [64][k_pad]
    const float* scale;        // [64] folded BatchNorm
    const float* shift;
    void* y;                   // [B][Hp][Wp][64]
    int batch, H, W, Hc, Wc, Hp, Wp;
    int tiles_y, tiles_x, n_tiles, k_pad;
    unsigned x_bytes;
};

constexpr unsigned OOB = 0x80000000u;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Geometry of one workgroup tile: TPH x TPW pooled pixels <- (2 TPH + 1) x (2 TPW + 1) conv pixels <- a PR x PC image patch.
// GEMM row m = cy * CWS + cx with CWS = 2 TPW + 2: one dummy column per conv row keeps every row of the conv tile dword-aligned for the
// pooling pass (its results are never read); the row count still rounds to the same number of 32-row MFMA tiles (8 x 8: 306 vs 289 -> 10).
template <bool BF16, int TPH, int TPW>
struct Cfg {
    static constexpr int CH = 2 * TPH + 1, CW = 2 * TPW + 1, CWS = CW + 1;
    static constexpr int MROWS = CH * CWS;
    static constexpr int MT = (MROWS + 31) / 32;
    static constexpr int PR = 2 * (CH - 1) + 7;
    static constexpr int PC = 2 * (CW - 1) + 8;            // bf16: the GEMM's 8 x slots; fp32: 7 taps + the padding slot the GEMM also reads
    static constexpr int NPIX = PR * PC;
    static constexpr int NPF = (NPIX + 255) / 256;
    static constexpr int PATCH_BYTES = BF16 ? NPIX * 8 : 3 * NPIX * 4;
    static constexpr int ES = BF16 ? 2 : 4;
    static constexpr int OSTRD = (MT * 32 * ES / 4) | 1;   // conv tile: dwords per channel row, odd (conflict-free across the 64 channels)
    static constexpr int LDS_BYTES = PATCH_BYTES + 64 * OSTRD * 4;
    static constexpr int NDW = BF16 ? CWS / 2 : CW;        // dwords of one conv row that the pooling pass reads
    static_assert(MT % 2 == 0 && TPH % 4 == 0, "fp32: the two row halves get MT / 2 MFMA tiles each; pooling: TPH / 4 pooled rows per wave");
};

template <bool BF16, int TPH, int TPW, bool U8 = false>
__global__ __launch_bounds__(256, BF16 ? 2 : 1) void stem_pool_kernel(const StemArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    using C = Cfg<BF16, TPH, TPW>;
    constexpr int PC = C::PC, NPIX = C::NPIX, NPF = C::NPF, MT = C::MT, CWS = C::CWS;
    unsigned char* const patch = smem;
    unsigned* const outd = reinterpret_cast<unsigned*>(smem + C::PATCH_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), (short)0, p.x_bytes, 0x00020000);
    const int plane4 = p.H * p.W * 4;

    // ---- B fragments: this wave's share of the packed weights, loaded once ----
    // bf16: every wave multiplies by all 64 output channels (waves split the GEMM rows); fp32: wave (wm, wn) = (rows half, channel half)
    const int wm = wave >> 1, wn = wave & 1;
    u32x4 fb16[BF16 ? 14 : 1][2];
    float fb32[BF16 ? 1 : 28][3];
    float sc[2], sh[2];
    if constexpr (BF16) {
        const __bf16* wp = reinterpret_cast<const __bf16*>(p.w);
#pragma unroll
        for (int j = 0; j < 14; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                fb16[j][n] = *reinterpret_cast<const u32x4*>(wp + (size_t)(n * 32 + fr) * p.k_pad + 16 * j + 8 * fh);
#pragma unroll
        for (int n = 0; n < 2; ++n) { sc[n] = p.scale[n * 32 + fr]; sh[n] = p.shift[n * 32 + fr]; }
    } else {
        const float* wp = reinterpret_cast<const float*>(p.w);
#pragma unroll
        for (int jj = 0; jj < 28; ++jj) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(wp + (size_t)(wn * 32 + fr) * p.k_pad + 8 * jj + 4 * fh);
            fb32[jj][0] = t[0]; fb32[jj][1] = t[1]; fb32[jj][2] = t[2];
        }
        sc[0] = p.scale[wn * 32 + fr]; sh[0] = p.shift[wn * 32 + fr];
        sc[1] = 0.f; sh[1] = 0.f;
    }

    // ---- patch prefetch: this thread's pixels idx = tid + 256 i of the PR x PC patch (row, column and offset inside the image are the
    // same for every tile), three planes each, zero outside the image ----
    int prel[NPF];
    unsigned prc[NPF];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / PC, c = idx - r * PC;
        prel[i] = r * p.W + c;
        prc[i] = idx < NPIX ? (unsigned)((r << 8) | c) : 0xffffff00u;     // a row that is outside every image
    }
    float pf[NPF][3];
    unsigned okmask = 0;                                   // U8: which of this thread's pixels lie inside the image (a byte 0 is a black pixel, not padding)
    auto origin = [&](int t, int& b, int& py0, int& px0) {
        const int per = p.tiles_y * p.tiles_x;
        b = t / per;
        const int r = t - b * per;
        const int ty = r / p.tiles_x;
        py0 = ty * TPH;
        px0 = (r - ty * p.tiles_x) * TPW;
    };
    // one pixel's three values into pf[i]: fp32 planes, or the B, G, R bytes of an HWC pixel (kept as integers until park())
    auto fetch = [&](int i, int pix, bool ok) {
        if constexpr (U8) {
            const unsigned off = ok ? (unsigned)(pix * 3) : OOB;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)     // channel ch of the network's RGB input is byte 2 - ch
                pf[i][ch] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b8(xr, off, 2 - ch, 0));
        } else {
            const unsigned off = ok ? (unsigned)(pix * 4) : OOB;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                pf[i][ch] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, ch * plane4, 0));
        }
    };
    auto prefetch = [&](int t) {
        int b, py0, px0;
        origin(t, b, py0, px0);
        const int iy0 = 4 * py0 - 5, ix0 = 4 * px0 - (BF16 ? 6 : 5);
        const int base = (b * (U8 ? 1 : 3) * p.H + iy0) * p.W + ix0;
        okmask = 0;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + C::PR <= p.H && ix0 + PC <= p.W) {      // the whole patch lies inside the image (uniform)
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const bool ok = (i + 1) * 256 <= NPIX || tid + 256 * i < NPIX;
                okmask |= (unsigned)ok << i;
                fetch(i, base + prel[i], ok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int gy = iy0 + (int)(prc[i] >> 8), gx = ix0 + (int)(prc[i] & 255);
                const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                okmask |= (unsigned)ok << i;
                fetch(i, base + prel[i], ok);
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NPIX) {
                float v3[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    if constexpr (U8) {        // sp_u8hwc_bgr_to_nhwc's arithmetic; padding stays 0
                        const float n = (float)__builtin_bit_cast(unsigned, pf[i][ch]) / 255.0f - p.mean[ch];
                        v3[ch] = (okmask >> i) & 1 ? n : 0.f;
                    } else {
                        v3[ch] = pf[i][ch];
                    }
                }
                if constexpr (BF16) {
                    const bf16x4 v = {(__bf16)v3[0], (__bf16)v3[1], (__bf16)v3[2], (__bf16)0.f};
                    *reinterpret_cast<u32x2*>(patch + idx * 8) = __builtin_bit_cast(u32x2, v);
                } else {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) reinterpret_cast<float*>(patch)[ch * NPIX + idx] = v3[ch];
                }
            }
        }
    };

#ifdef SP_STEM_DIAG
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    SP_SSTAMP(tk0)
    int tile = blockIdx.x;
    if (tile < p.n_tiles) prefetch(tile);
    for (; tile < p.n_tiles; tile += gridDim.x) {
        SP_SSTAMP(t0)
        park();
        __syncthreads();                                   // patch complete; every wave is past the previous tile's pooling pass
        SP_SSTAMP(t1)
        const int next = tile + gridDim.x;
        if (next < p.n_tiles) prefetch(next);              // in flight behind the MFMAs
        SP_SSTAMP(t2)

        // ---- GEMM rows of this wave: conv pixel (cy, cx) of the tile = row cy * CWS + cx ----
        for (int mt = BF16 ? wave : wm * (MT / 2); mt < (BF16 ? MT : (wm + 1) * (MT / 2)); mt += BF16 ? 4 : 1) {
            const int m = mt * 32 + fr;
            const int mc = m < C::MROWS ? m : 0;
            const int cy = mc / CWS, cx = mc - CWS * cy;
            if constexpr (BF16) {
                const unsigned char* ap = patch + ((2 * cy) * PC + 2 * cx) * 8 + fh * 16;
                f32x16 acc[2];
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
                for (int j = 0; j < 14; ++j) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(ap + (j >> 1) * (PC * 8) + (j & 1) * 32);
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, fb16[j][n]), acc[n], 0, 0, 0);
                }
                // BatchNorm + ReLU on the accumulators, bf16, channel-major into the conv tile: 4 consecutive rows = two dwords
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    unsigned* orow = outd + (n * 32 + fr) * C::OSTRD + (mt * 32 + 4 * fh) / 2;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[n][4 * g + e] * sc[n] + sh[n];
                            v = v > 0.f ? v : 0.f;
                            o[e] = (__bf16)v;
                        }
                        const u32x2 o2 = __builtin_bit_cast(u32x2, o);
                        orow[4 * g] = o2[0];
                        orow[4 * g + 1] = o2[1];
                    }
                }
            } else {
                const float* ap = reinterpret_cast<const float*>(patch) + (2 * cy) * PC + 2 * cx + fh;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int jj = 0; jj < 28; ++jj) {          // jj = ky * 4 + pp
                    float a[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) a[s] = ap[s * NPIX + (jj >> 2) * PC + 2 * (jj & 3)];
#pragma unroll
                    for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], fb32[jj][s], acc, 0, 0, 0);
                }
                float* orow = reinterpret_cast<float*>(outd) + (wn * 32 + fr) * C::OSTRD + mt * 32 + 4 * fh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * g + e] * sc[0] + sh[0];
                        v = v > 0.f ? v : 0.f;
                        orow[8 * g + e] = v;
                    }
                }
            }
        }
        SP_SSTAMP(t3)
        __syncthreads();                                   // conv tile complete
        SP_SSTAMP(t4)

        // ---- MaxPool 3x3 s2 p1: wave w owns TPH / 4 pooled rows of the tile, lane = channel.  Every value in the conv tile is a ReLU
        // output (>= +0, never NaN: `v > 0 ? v : 0` maps NaN to 0), so fp32 / bf16 order equals the unsigned order of the bit patterns and a
        // window position outside the image can stand in as 0: integer max (two bf16 per instruction), the same result as the -inf-padded
        // NaN-propagating pooling of sp_maxpool3x3s2_nhwc on such values.  A wave issues one instruction per 4+ cycles, so the pass is
        // priced in instructions: tiles whose windows and outputs all lie inside the image (uniform test) skip every check ----
        {
            int b, py0, px0;
            origin(tile, b, py0, px0);
            const unsigned* rowp = outd + lane * C::OSTRD;
            const int lim = p.Wc - 2 * px0 + 1;            // conv columns e of the tile with 2 px0 - 1 + e < Wc
            const int elo = px0 == 0 ? 1 : 0;              // column -1 of the image
            auto pool = [&](auto check_tag) {
                constexpr bool CHECK = decltype(check_tag)::value;
                constexpr int RD = BF16 ? CWS / 2 : CWS;   // dwords per conv row in the tile
#pragma unroll
                for (int pr2 = 0; pr2 < TPH / 4; ++pr2) {
                    const int pr = (TPH / 4) * wave + pr2;
                    const int py = py0 + pr;
                    if (CHECK && py >= p.Hp) continue;
                    unsigned vm[C::NDW];
                    if constexpr (CHECK) {
#pragma unroll
                        for (int d = 0; d < C::NDW; ++d) vm[d] = 0u;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            const int cyg = 2 * py - 1 + ky;   // conv row in the image
                            if ((unsigned)cyg >= (unsigned)p.Hc) continue;
                            const unsigned* rp = rowp + (2 * pr + ky) * RD;
#pragma unroll
                            for (int d = 0; d < C::NDW; ++d) {
                                const unsigned v = rp[d];
                                if constexpr (BF16) vm[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, vm[d]), __builtin_bit_cast(u16x2, v)));
                                else vm[d] = vm[d] > v ? vm[d] : v;
                            }
                        }
                    } else {
                        const unsigned* rp = rowp + (2 * pr) * RD;
#pragma unroll
                        for (int d = 0; d < C::NDW; ++d) {
                            const unsigned v0 = rp[d], v1 = rp[RD + d], v2 = rp[2 * RD + d];
                            if constexpr (BF16) {
                                const u16x2 t = __builtin_elementwise_max(__builtin_bit_cast(u16x2, v0), __builtin_bit_cast(u16x2, v1));
                                vm[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(t, __builtin_bit_cast(u16x2, v2)));
                            } else {
                                const unsigned t = v0 > v1 ? v0 : v1;
                                vm[d] = t > v2 ? t : v2;
                            }
                        }
                    }
                    unsigned short* yb = reinterpret_cast<unsigned short*>(p.y) + (((size_t)b * p.Hp + py) * p.Wp + px0) * 64 + lane;
                    unsigned* yf = reinterpret_cast<unsigned*>(p.y) + (((size_t)b * p.Hp + py) * p.Wp + px0) * 64 + lane;
                    if constexpr (BF16) {
                        if constexpr (CHECK) {
#pragma unroll
                            for (int d = 0; d < C::NDW; ++d) {     // columns outside the image -> 0
                                const unsigned mk = ((2 * d >= elo && 2 * d < lim) ? 0xffffu : 0u) | ((2 * d + 1 < lim) ? 0xffff0000u : 0u);
                                vm[d] &= mk;
                            }
                        }
#pragma unroll
                        for (int pc = 0; pc < TPW; ++pc) {
                            const unsigned a = vm[pc] & 0xffffu, bq = vm[pc] >> 16, c = vm[pc + 1] & 0xffffu;
                            const unsigned ab = a > bq ? a : bq;
                            const unsigned o = ab > c ? ab : c;
                            if ((!CHECK || px0 + pc < p.Wp) SP_STEM_STORE_COND(o)) yb[pc * 64] = (unsigned short)o;
                        }
                    } else {
                        if constexpr (CHECK) {
#pragma unroll
                            for (int e = 0; e < C::NDW; ++e)
                                if (!(e >= elo && e < lim)) vm[e] = 0u;
                        }
#pragma unroll
                        for (int pc = 0; pc < TPW; ++pc) {
                            const unsigned ab = vm[2 * pc] > vm[2 * pc + 1] ? vm[2 * pc] : vm[2 * pc + 1];
                            const unsigned o = ab > vm[2 * pc + 2] ? ab : vm[2 * pc + 2];
                            if ((!CHECK || px0 + pc < p.Wp) SP_STEM_STORE_COND(o)) yf[pc * 64] = o;
                        }
                    }
                }
            };
            const bool inside = py0 > 0 && px0 > 0 && py0 + TPH <= p.Hp && px0 + TPW <= p.Wp && 2 * (py0 + TPH) <= p.Hc && 2 * (px0 + TPW) <= p.Wc;
            if (inside) pool(std::false_type{});
            else pool(std::true_type{});
        }
        SP_SSTAMP(t5)
        SP_SACC(0, t0, t1) SP_SACC(1, t1, t2) SP_SACC(2, t2, t3) SP_SACC(3, t3, t4) SP_SACC(4, t4, t5)
#ifdef SP_STEM_DIAG
        dg[7] += 1;
#endif
    }
#ifdef SP_STEM_DIAG
    {
        SP_SSTAMP(tk1)
        dg[6] = tk1 - tk0;
        if (lane == 0) {
            unsigned long long* o = sp_stem_dbg + ((blockIdx.x & 511) * 4 + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = dg[i];
        }
    }
#endif
}

template <bool BF16, int TPH, int TPW, bool U8 = false>
int launch_stem(StemArgs a, hipStream_t stream) {
    using C = Cfg<BF16, TPH, TPW>;
    static bool opted[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!opted[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<BF16, TPH, TPW, U8>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) {
            sp_set_error("stem: hipFuncSetAttribute(max dynamic LDS = %d) failed on device %d", C::LDS_BYTES, dev);
            return SP_ELAUNCH;
        }
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        opted[dev] = true;
    }
    a.tiles_y = (a.Hp + TPH - 1) / TPH;
    a.tiles_x = (a.Wp + TPW - 1) / TPW;
    a.n_tiles = a.batch * a.tiles_y * a.tiles_x;
    const int slots = cus[dev] * (BF16 ? 2 : 1);
    // whole rounds: every persistent workgroup gets the same number of tiles (4,096 tiles at bs = 128 on 256 CUs: 8 / 16 each)
    const int rounds = (a.n_tiles + slots - 1) / slots;
    const int grid = (a.n_tiles + rounds - 1) / rounds;
    hipLaunchKernelGGL((stem_pool_kernel<BF16, TPH, TPW, U8>), dim3(grid), dim3(256), C::LDS_BYTES, stream, a);
    return sp_check_launch("stem_pool_kernel");
}

}  // namespace

extern "C" int sp_stem7_pool_ok(int batch, int h, int w) {
    // the kernel's index arithmetic: 32-bit byte offsets into the image, tiles counted in an int
    return batch > 0 && h >= 8 && w >= 8 && (long long)batch * 3 * h * w * 4 < (1ll << 31) ? 1 : 0;
}

static int stem_args(StemArgs& a, const void* x, const void* w_packed, int k_pad, const float* scale, const float* shift, void* y, int batch, int h,
                     int w, int bytes_per_pixel_plane) {
    SP_REQUIRE(x && w_packed && scale && shift && y, "sp_stem7_pool: null pointer");
    SP_REQUIRE(sp_stem7_pool_ok(batch, h, w), "sp_stem7_pool: batch %d of %d x %d images is outside the kernel's 32-bit offsets", batch, h, w);
    SP_REQUIRE(k_pad >= 224 && k_pad % 8 == 0, "sp_stem7_pool: k_pad %d (the packed 7x7 stem has K >= 224)", k_pad);
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y;
    a.mean[0] = a.mean[1] = a.mean[2] = 0.f;
    a.batch = batch; a.H = h; a.W = w;
    a.Hc = (h + 6 - 7) / 2 + 1; a.Wc = (w + 6 - 7) / 2 + 1;
    a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;
    a.tiles_y = a.tiles_x = a.n_tiles = 0;                 // set by the launcher (tile shape)
    a.k_pad = k_pad;
    a.x_bytes = (unsigned)((long long)batch * 3 * h * w * bytes_per_pixel_plane);
    return SP_OK;
}

extern "C" int sp_stem7_pool(const float* x, const void* w_packed, int k_pad, const float* scale, const float* shift, void* y, int bf16,
                             int batch, int h, int w, void* stream) {
    StemArgs a;
    const int rc = stem_args(a, x, w_packed, k_pad, scale, shift, y, batch, h, w, 4);
    if (rc != SP_OK) return rc;
    if (sp_name_query_active()) {
        sp_name_query_set("stem_pool_kernel<%s>", bf16 ? "true" : "false");
        return SP_OK;
    }
    // 8 x 12 pooled pixels per workgroup tile (14 MFMA row tiles for 96 outputs; the 8 x 8 tile needs 10 for 64 and is 5-8 % slower at
    // bs = 128: tools/diag_stem.py); SP_STEM_TILE_W=8 selects it for that comparison.  Same bits either way.
    static const int wide = [] { const char* e = getenv("SP_STEM_TILE_W"); return e ? atoi(e) : 12; }();
    if (wide == 8) return bf16 ? launch_stem<true, 8, 8>(a, (hipStream_t)stream) : launch_stem<false, 8, 8>(a, (hipStream_t)stream);
    return bf16 ? launch_stem<true, 8, 12>(a, (hipStream_t)stream) : launch_stem<false, 8, 12>(a, (hipStream_t)stream);
}

extern "C" int sp_stem7_pool_u8(const unsigned char* crops_bgr, const float* mean_rgb_host, const void* w_packed, int k_pad, const float* scale,
                                const float* shift, void* y, int bf16, int batch, int h, int w, void* stream) {
    SP_REQUIRE(mean_rgb_host, "sp_stem7_pool_u8: null pointer");
    StemArgs a;
    const int rc = stem_args(a, crops_bgr, w_packed, k_pad, scale, shift, y, batch, h, w, 1);
    if (rc != SP_OK) return rc;
    for (int c = 0; c < 3; ++c) a.mean[c] = mean_rgb_host[c];
    if (sp_name_query_active()) {
        sp_name_query_set("stem_pool_kernel<%s, u8>", bf16 ? "true" : "false");
        return SP_OK;
    }
    return bf16 ? launch_stem<true, 8, 12, true>(a, (hipStream_t)stream) : launch_stem<false, 8, 12, true>(a, (hipStream_t)stream);
}
