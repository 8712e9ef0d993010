// train.hip - HBM-bound kernels of the training step (processors/ddp_pose_resnet_solver.py:110-133):
// train-mode BatchNorm forward/backward on NHWC fp32, max-pool backward, fused Adam, weight (re)packing, layout helpers.
// All reductions accumulate in double and are deterministic (fixed grid, per-block partials, ordered final sum).
#include "sp_common.h"
#include <stdlib.h>

namespace {


typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 4 consecutive channels of an NHWC tensor, fp32 (16 B) or bf16 (8 B), as floats; i4 = index in units of 4 channels
template <bool BF16>
__device__ __forceinline__ f32x4 ld4(const void* p, long long i4) {
    if constexpr (BF16) {
        const bf16x4 v = __builtin_bit_cast(bf16x4, reinterpret_cast<const u32x2*>(p)[i4]);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return reinterpret_cast<const f32x4*>(p)[i4];
    }
}
template <bool BF16>
__device__ __forceinline__ void st4(void* p, long long i4, f32x4 v) {
    if constexpr (BF16) {
        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        reinterpret_cast<u32x2*>(p)[i4] = __builtin_bit_cast(u32x2, o);
    } else {
        reinterpret_cast<f32x4*>(p)[i4] = v;
    }
}

// BatchNorm element maps shared by the plain and the fused (fold-in-prologue) passes.  Contraction is switched off inside them: the
// compiler's choice of fused multiply-adds depends on the surrounding code, and the two ways of running a layer must give the same bits.
__device__ __forceinline__ f32x4 bn_fwd_elem(f32x4 v, const f32x4 mu, const f32x4 is, const f32x4 g, const f32x4 b) {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] - mu[e]) * is[e] * g[e] + b[e];
    return v;
}
__device__ __forceinline__ f32x4 bn_bwd_elem(const f32x4 g, const f32x4 zz, const f32x4 mu, const f32x4 is, const f32x4 ga, const f32x4 dg,
                                             const f32x4 db, const float inv_m) {
#pragma clang fp contract(off)
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = (zz[e] - mu[e]) * is[e];
        o[e] = ga[e] * is[e] * (g[e] - db[e] * inv_m - xh * dg[e] * inv_m);
    }
    return o;
}

// partial-sum workgroups of a channel reduction: enough to fill the chip (>= 4 rows per thread), bounded by the 4 MB workspace
// ([blocks][C][2] doubles)
static int red_blocks(long long rows, int c) {
    const int c4 = c / 4, lanes_c = c4 < 256 ? c4 : 256, rows_par = 256 / lanes_c;
    long long n = (rows + (long long)rows_par * 4 - 1) / ((long long)rows_par * 4);
    const long long cap = (262144 / c) < 1024 ? (262144 / c) : 1024;
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

inline int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// ---------------------------------------------------------------------------------------------------------------
// per-channel sums over the M rows of an NHWC tensor: thread = (4 channels) x (row stripe)
//   mode 0: s0 = sum z,            s1 = sum z^2                         (BN forward statistics)
//   mode 1: s0 = sum g,            s1 = sum g * xhat,  g = dy*(y>0?)    (BN backward: dbeta, dgamma)
// part: [gridDim.x][C][2] doubles
// ---------------------------------------------------------------------------------------------------------------
template <int MODE, bool BF16, bool G16 = BF16>   // BF16: dtype of z / relu_src (and of `a` in mode 0); G16: dtype of `a` in mode 1 (dy)
__global__ __launch_bounds__(256) void channel_reduce_kernel(const void* __restrict__ a, const void* __restrict__ relu_src,
                                                             const void* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int M, int C, double* __restrict__ part) {
    const int C4 = C >> 2;
    const int lanes_c = C4 < 256 ? C4 : 256;       // threads along channels
    const int rows_par = 256 / lanes_c;            // row stripes inside the block
    const int tc = threadIdx.x % lanes_c, tr = threadIdx.x / lanes_c;
    __shared__ double sm[256 * 8];
    for (int c4 = tc; c4 < C4; c4 += lanes_c) {
        double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
        f32x4 mu = {0, 0, 0, 0}, is = {0, 0, 0, 0};
        if (MODE == 1) { mu = reinterpret_cast<const f32x4*>(mean)[c4]; is = reinterpret_cast<const f32x4*>(invstd)[c4]; }
        if (tr < rows_par) {
            for (long long r = (long long)blockIdx.x * rows_par + tr; r < M; r += (long long)gridDim.x * rows_par) {
                const f32x4 v = (MODE == 0) ? ld4<BF16>(a, r * C4 + c4) : ld4<G16>(a, r * C4 + c4);
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0[e] += (double)v[e]; s1[e] += (double)v[e] * (double)v[e]; }
                } else {
                    f32x4 g = v;
                    if (relu_src) {
                        const f32x4 y = ld4<BF16>(relu_src, r * C4 + c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
                    }
                    const f32x4 zz = ld4<BF16>(z, r * C4 + c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0[e] += (double)g[e]; s1[e] += (double)g[e] * (double)((zz[e] - mu[e]) * is[e]); }
                }
            }
        }
        // fold the row stripes of this block (fixed order)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sm[threadIdx.x * 8 + e] = s0[e]; sm[threadIdx.x * 8 + 4 + e] = s1[e]; }
        __syncthreads();
        if (tr == 0) {
            for (int k = 1; k < rows_par; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) sm[tc * 8 + e] += sm[(k * lanes_c + tc) * 8 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part[((size_t)blockIdx.x * C + c4 * 4 + e) * 2 + 0] = sm[tc * 8 + e];
                part[((size_t)blockIdx.x * C + c4 * 4 + e) * 2 + 1] = sm[tc * 8 + 4 + e];
            }
        }
        __syncthreads();
    }
}

// Final fold of the per-block partials: one wave per channel (4 channels per workgroup), each lane sums nblk/64
// partials in a fixed order, then a fixed-order butterfly - deterministic and ~3 us instead of a 256-deep serial chain.
__device__ __forceinline__ void fold_partials(const double* __restrict__ part, int nblk, int C, int c, double& s0, double& s1) {
    const int lane = threadIdx.x & 63;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const f64x2* p2 = reinterpret_cast<const f64x2*>(part);       // (s0, s1) pairs: one 16-byte load each
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, d0 = 0, d1 = 0;
    int b = lane;
    for (; b + 192 < nblk; b += 256) {                             // four independent chains: four loads in flight per lane
        const f64x2 v0 = p2[(size_t)b * C + c], v1 = p2[(size_t)(b + 64) * C + c], v2 = p2[(size_t)(b + 128) * C + c],
                    v3 = p2[(size_t)(b + 192) * C + c];
        a0 += v0.x; a1 += v0.y; b0 += v1.x; b1 += v1.y; c0 += v2.x; c1 += v2.y; d0 += v3.x; d1 += v3.y;
    }
    for (; b < nblk; b += 64) { const f64x2 v = p2[(size_t)b * C + c]; a0 += v.x; a1 += v.y; }
    s0 = (a0 + b0) + (c0 + d0); s1 = (a1 + b1) + (c1 + d1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, SP_WAVE); s1 += __shfl_xor(s1, off, SP_WAVE); }
}

// BN forward finalize: mean, invstd (biased var), running stats with momentum and UNBIASED var (torch semantics)
__global__ void bn_stats_final_kernel(const double* __restrict__ part, int nblk, int C, double M, float eps, float momentum,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                      float* __restrict__ run_var) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s0, s1;
    fold_partials(part, nblk, C, c, s0, s1);
    if ((threadIdx.x & 63) != 0) return;
    const double mu = s0 / M;
    double var = s1 / M - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = M > 1 ? var * M / (M - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + (double)momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + (double)momentum * unb);
    }
}

__global__ void pair_sum_final_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ o0, float* __restrict__ o1) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s0, s1;
    fold_partials(part, nblk, C, c, s0, s1);
    if ((threadIdx.x & 63) != 0) return;
    if (o0) o0[c] = (float)s0;
    if (o1) o1[c] = (float)s1;
}

// Fold of the per-(phase, M tile) partial rows a STATS / BSTATS conv epilogue wrote ([nrows][stride] fp32, two arrays) for ONE slab of 64
// channels, by a workgroup of 1,024 threads = 16 channel quads x 64 row lanes: thread (quad q, row lane rl) adds rows rl, rl + 64, ... of
// its four channels in fp64 (two chains: even and odd trips), the 4 row lanes of a wave meet through two xor shuffles, the 16 waves
// through LDS in wave order.  Every order is fixed by the shape -> deterministic; and the SAME function is the prologue of the fused
// BatchNorm passes below and the body of the stand-alone fold kernels, so the two ways of running a layer agree bit for bit.
// Result: threads tid < 64 hold (sum0, sum1) of channel c0 + tid; returns whether that channel exists.
constexpr int SLAB = 64, FOLD_THREADS = 1024;
// QUADS channel quads x 64 row lanes: QUADS = 16 is the 64-channel slab of 1,024 threads (the fused passes below), QUADS = 4 a 16-channel slab of 256
// threads (the stand-alone folds: a layer1 / stem fold is 0.4-1.5 MB of partial rows, and ONE workgroup per 64 channels read them at one CU's
// 50-60 GB/s - 29 / 65 us for the stem's pair; four times the workgroups, the SAME sums: a channel's rows are added in the same order by the same
// tree whatever QUADS is - row lane rl takes rows rl, rl + 64, ..., groups of four row lanes combine as (t0 + t1) + (t2 + t3), the sixteen groups in order)
template <int QUADS>
__device__ __forceinline__ bool fold_slab(const float* __restrict__ ps, const float* __restrict__ pq, int nrows, int stride, int C, int c0,
                                          double& s0, double& s1) {
    constexpr int SL = 4 * QUADS;
    __shared__ double sh[16][SL][2];                    // [group of four row lanes][channel][stat]: 16 KB at QUADS = 16
    const int tid = threadIdx.x, q = tid % QUADS, rl = tid / QUADS, wave = rl >> 2;
    const int c = c0 + 4 * q;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
    if (c < C) {
        int r = rl;
        // (two trips per iteration: eight 16-byte loads in flight per thread, added in the order of the one-trip loop below - same bits; a
        // 3,072-row fold is 24 dependent trips otherwise)
        for (; r + 192 < nrows; r += 256) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(ps + (size_t)r * stride + c), u1 = *reinterpret_cast<const f32x4*>(pq + (size_t)r * stride + c);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ps + (size_t)(r + 64) * stride + c), v1 = *reinterpret_cast<const f32x4*>(pq + (size_t)(r + 64) * stride + c);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(ps + (size_t)(r + 128) * stride + c), w1 = *reinterpret_cast<const f32x4*>(pq + (size_t)(r + 128) * stride + c);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(ps + (size_t)(r + 192) * stride + c), x1 = *reinterpret_cast<const f32x4*>(pq + (size_t)(r + 192) * stride + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += (double)u0[e]; a1[e] += (double)u1[e]; b0[e] += (double)v0[e]; b1[e] += (double)v1[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += (double)w0[e]; a1[e] += (double)w1[e]; b0[e] += (double)x0[e]; b1[e] += (double)x1[e]; }
        }
        for (; r + 64 < nrows; r += 128) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(ps + (size_t)r * stride + c), u1 = *reinterpret_cast<const f32x4*>(pq + (size_t)r * stride + c);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(ps + (size_t)(r + 64) * stride + c), v1 = *reinterpret_cast<const f32x4*>(pq + (size_t)(r + 64) * stride + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += (double)u0[e]; a1[e] += (double)u1[e]; b0[e] += (double)v0[e]; b1[e] += (double)v1[e]; }
        }
        for (; r < nrows; r += 64) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(ps + (size_t)r * stride + c), u1 = *reinterpret_cast<const f32x4*>(pq + (size_t)r * stride + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += (double)u0[e]; a1[e] += (double)u1[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        double t0 = a0[e] + b0[e], t1 = a1[e] + b1[e];
        t0 += __shfl_xor(t0, QUADS, SP_WAVE); t1 += __shfl_xor(t1, QUADS, SP_WAVE);     // row lanes 4w+0/1 and 4w+2/3
        t0 += __shfl_xor(t0, 2 * QUADS, SP_WAVE); t1 += __shfl_xor(t1, 2 * QUADS, SP_WAVE);
        if ((rl & 3) == 0) { sh[wave][4 * q + e][0] = t0; sh[wave][4 * q + e][1] = t1; }
    }
    __syncthreads();
    bool have = false;
    if (tid < SL) {
        s0 = 0; s1 = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { s0 += sh[w][tid][0]; s1 += sh[w][tid][1]; }
        have = c0 + tid < C;
    }
    __syncthreads();                                    // (the slab's LDS may be folded into again: second array pair of a BSTATS2 launch)
    return have;
}

__device__ __forceinline__ bool fold_slab64(const float* __restrict__ ps, const float* __restrict__ pq, int nrows, int stride, int C, int c0,
                                            double& s0, double& s1) {
    return fold_slab<16>(ps, pq, nrows, stride, C, c0, s0, s1);
}
constexpr int SFQ = 4, SFC = 4 * SFQ, SFT = 64 * SFQ;      // the stand-alone folds: 16 channels, 256 threads per workgroup

__device__ __forceinline__ void bn_finalize(double s0, double s1, double M, float eps, float momentum, int c, float& mu_f, float& is_f,
                                            float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                            float* __restrict__ run_var, bool publish) {
    const double mu = s0 / M;
    double var = s1 / M - mu * mu;
    if (var < 0) var = 0;
    mu_f = (float)mu;
    is_f = (float)(1.0 / sqrt(var + (double)eps));
    if (publish) {
        mean[c] = mu_f;
        invstd[c] = is_f;
        if (run_mean) {
            const double unb = M > 1 ? var * M / (M - 1) : var;
            run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + (double)momentum * mu);
            run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + (double)momentum * unb);
        }
    }
}

// BN forward statistics from the partial sums of the STATS conv epilogue (fp32 sums of one M tile's rows each), then the same
// finalisation as bn_stats_final_kernel.  One workgroup per 16-channel slab.
__global__ __launch_bounds__(SFT) void bn_stats_from_conv_kernel(const float* __restrict__ ps, const float* __restrict__ pq, int nrows, int stride, int C,
                                          double M, float eps, float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                          float* __restrict__ run_mean, float* __restrict__ run_var) {
    double s0, s1;
    const int c0 = blockIdx.x * SFC;
    if (!fold_slab<SFQ>(ps, pq, nrows, stride, C, c0, s0, s1)) return;
    float mu, is;
    bn_finalize(s0, s1, M, eps, momentum, c0 + threadIdx.x, mu, is, mean, invstd, run_mean, run_var, true);
}

// SyncBatchNorm on the fused statistics: the same fold, stopped before the finalisation - this rank's (sum, sum of squares) per channel
// in fp64, the [c][2] layout sp_bn_train_finalize consumes after the cross-rank SUM
__global__ __launch_bounds__(SFT) void bn_sums_from_conv_kernel(const float* __restrict__ ps, const float* __restrict__ pq, int nrows, int stride, int C,
                                         double* __restrict__ sums) {
    double s0, s1;
    const int c0 = blockIdx.x * SFC;
    if (!fold_slab<SFQ>(ps, pq, nrows, stride, C, c0, s0, s1)) return;
    sums[2 * (c0 + threadIdx.x)] = s0;
    sums[2 * (c0 + threadIdx.x) + 1] = s1;
}

// dbeta = sum g, dgamma = sum g*xhat from the partial rows a BSTATS dgrad launch (or several: one per output phase) left
// (gridDim.y = 2: the second row of workgroups folds (ps, pq2) into (dbeta2, dgamma2) - the projection shortcut's BatchNorm, whose d beta is
// the same sum g and whose d gamma comes from the third array of the BSTATS2 epilogue: one launch for the pair)
__global__ __launch_bounds__(SFT) void bn_bwd_sums_from_conv_kernel(const float* __restrict__ ps, const float* __restrict__ pq, int nrows, int stride, int C,
                                             float* __restrict__ dbeta, float* __restrict__ dgamma, float* __restrict__ dbeta_copy,
                                             float* __restrict__ dgamma_copy, const float* __restrict__ pq2 = nullptr,
                                             float* __restrict__ dbeta2 = nullptr, float* __restrict__ dgamma2 = nullptr) {
    if (blockIdx.y == 1) { pq = pq2; dbeta = dbeta2; dgamma = dgamma2; dbeta_copy = nullptr; dgamma_copy = nullptr; }
    double s0, s1;
    const int c0 = blockIdx.x * SFC;
    if (!fold_slab<SFQ>(ps, pq, nrows, stride, C, c0, s0, s1)) return;
    const int c = c0 + threadIdx.x;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
    if (dbeta_copy) {                  // SyncBatchNorm: the same sums again where the message is assembled (the parameter gradients keep the local ones)
        dbeta_copy[c] = (float)s0;
        dgamma_copy[c] = (float)s1;
    }
}

// ---- the fold as the PROLOGUE of the consuming BatchNorm pass (round 4) -------------------------------------------------------------------
// At 32 images per GPU the step is a chain of ~400 dependent launches and the 111 stand-alone folds (5-9 us each, plus a launch boundary
// and ~10 us of host time) were a seventh of it.  With one partial row per (phase, M tile) a 64-channel slab of a layer3 / layer4 /
// deconv0 tensor has 24-96 rows of 512 bytes to fold: every workgroup of the consuming pass does that itself (same order everywhere,
// so all workgroups of a slab hold the same bits) and then streams its stripe of rows.  Grid = (slabs of 64 channels) x (row stripes);
// thread = (channel quad, row lane).  The workgroups of stripe 0 publish mean / invstd (saved for backward) and the running statistics.
// Host side: used while nrows stays small (PoseTrainer.fold_in_consumer_rows); layer1 / the stem keep the stand-alone fold.
// VW consecutive channels of an NHWC tensor as floats (iv = index in units of VW channels): 16-bit elements come in one 8- or 16-byte
// load, fp32 in VW / 4 16-byte loads.  The fused passes below walk bf16 tensors 8 channels (16 bytes) per thread: the same pass with
// 8-byte accesses reaches 3.7 TB/s on layer1's block outputs, with 16-byte accesses 4.5 (tools/bench_bn_passes.py).
template <bool B16, int VW>
__device__ __forceinline__ void ldn(const void* p, long long iv, float (&o)[VW]) {
    if constexpr (B16 && VW == 8) {
        typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
        const bf16x8_ v = __builtin_bit_cast(bf16x8_, reinterpret_cast<const u32x4*>(p)[iv]);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    } else {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const f32x4 v = ld4<B16>(p, iv * (VW / 4) + h);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * h + e] = v[e];
        }
    }
}
template <bool B16, int VW>
__device__ __forceinline__ void stn(void* p, long long iv, const float (&o)[VW]) {
    if constexpr (B16 && VW == 8) {
        typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
        bf16x8_ v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)o[e];
        reinterpret_cast<u32x4*>(p)[iv] = __builtin_bit_cast(u32x4, v);
    } else {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) st4<B16>(p, iv * (VW / 4) + h, f32x4{o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]});
    }
}

// ReLU mask of 8 consecutive channels as one byte (bit e = channel e passed): bf16 training keeps it beside y, and the backward passes (and
// the BSTATS dgrad epilogue) read this byte instead of 16 bytes of y.  y > 0 after the bf16 rounding <=> the fp32 value > 0 (a positive
// float never rounds to zero in bf16: same exponent range).
__device__ __forceinline__ unsigned char relu_bits8(const float (&x)[8]) {
    unsigned m = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) m |= (x[e] > 0.f ? 1u : 0u) << e;
    return (unsigned char)m;
}

// After the fold (1,024 threads = 16 channel quads x 64 row lanes) the threads regroup as (64 / VW channel groups) x (row lanes) with
// VW = 4 channels (fp32) or 8 (bf16): 16 bytes per access either way.
template <bool BF16>
__global__ __launch_bounds__(FOLD_THREADS) void bn_fold_apply_kernel(const void* __restrict__ z, const float* __restrict__ ps, const float* __restrict__ pq,
                                                                     int nrows, int stride, double M, float eps, float momentum,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const void* __restrict__ res, void* __restrict__ y, int C, int relu, long long rows,
                                                                     float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ run_mean,
                                                                     float* __restrict__ run_var, unsigned char* __restrict__ mask) {
    constexpr int VW = BF16 ? 8 : 4, LPS = SLAB / VW, RLN = FOLD_THREADS / LPS, U = BF16 ? 2 : 4;
    __shared__ float st[2][SLAB];
    const int slab = blockIdx.x, stripe = blockIdx.y, stripes = gridDim.y;
    const int c0 = slab * SLAB, tid = threadIdx.x, qa = tid % LPS, rl = tid / LPS;
    const int c = c0 + VW * qa, CV = C / VW, cv = c / VW;
    const bool live = c < C;
    const long long step = (long long)stripes * RLN;
    long long r = (long long)stripe * RLN + rl;
    // the first U rows of this thread (and gamma / beta) are requested BEFORE the fold: their latency and the prologue's overlap
    float v[U][VW], rr[U][VW], g[VW], b[VW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long row = r + u * step;
        const bool ok = live && row < rows;
#pragma unroll
        for (int e = 0; e < VW; ++e) { v[u][e] = 0.f; rr[u][e] = 0.f; }
        if (ok) ldn<BF16, VW>(z, row * CV + cv, v[u]);
        if (ok && res) ldn<BF16, VW>(res, row * CV + cv, rr[u]);
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) { g[e] = 0.f; b[e] = 0.f; }
    if (live) { ldn<false, VW>(gamma, cv, g); ldn<false, VW>(beta, cv, b); }
    double s0, s1;
    if (fold_slab64(ps, pq, nrows, stride, C, c0, s0, s1)) {
        float mu, is;
        bn_finalize(s0, s1, M, eps, momentum, c0 + tid, mu, is, mean, invstd, run_mean, run_var, stripe == 0);
        st[0][tid] = mu; st[1][tid] = is;
    }
    __syncthreads();
    if (!live) return;
    float mu[VW], is[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { mu[e] = st[0][VW * qa + e]; is[e] = st[1][VW * qa + e]; }
    auto one = [&](float (&x)[VW], const float (&rx)[VW], long long row) {
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const int o = 4 * h;
            const f32x4 t = bn_fwd_elem(f32x4{x[o], x[o + 1], x[o + 2], x[o + 3]}, f32x4{mu[o], mu[o + 1], mu[o + 2], mu[o + 3]},
                                        f32x4{is[o], is[o + 1], is[o + 2], is[o + 3]}, f32x4{g[o], g[o + 1], g[o + 2], g[o + 3]},
                                        f32x4{b[o], b[o + 1], b[o + 2], b[o + 3]});
#pragma unroll
            for (int e = 0; e < 4; ++e) x[o + e] = t[e];
        }
        if (res) {
#pragma unroll
            for (int e = 0; e < VW; ++e) x[e] += rx[e];
        }
        if (relu) {
            if constexpr (VW == 8) {
                if (mask) mask[row * CV + cv] = relu_bits8(x);
            }
#pragma unroll
            for (int e = 0; e < VW; ++e) x[e] = x[e] > 0.f ? x[e] : 0.f;
        }
        stn<BF16, VW>(y, row * CV + cv, x);
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (r + u * step < rows) one(v[u], rr[u], r + u * step);
    r += U * step;
    for (; r + (U - 1) * step < rows; r += U * step) {          // U rows in flight per thread
#pragma unroll
        for (int u = 0; u < U; ++u) ldn<BF16, VW>(z, (r + u * step) * CV + cv, v[u]);
        if (res) {
#pragma unroll
            for (int u = 0; u < U; ++u) ldn<BF16, VW>(res, (r + u * step) * CV + cv, rr[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(v[u], rr[u], r + u * step);
    }
    for (; r < rows; r += step) {
        ldn<BF16, VW>(z, r * CV + cv, v[0]);
        if (res) ldn<BF16, VW>(res, r * CV + cv, rr[0]);
        one(v[0], rr[0], r);
    }
}

// Backward counterpart: dbeta = sum g and dgamma = sum g * xhat folded from the BSTATS partial rows in the prologue of the pass that
// needs them (bn_bwd_apply_kernel's body follows); stripe 0 writes the two parameter gradients.  pq2 != null: the projection shortcut's
// BatchNorm shares this g (its d beta is the same sum g, its d gamma = sum g * xhat2 from the third array): stripe 0 folds and writes those too.
template <bool BF16, bool G16>
__global__ __launch_bounds__(FOLD_THREADS) void bn_fold_bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ relu_src, const void* __restrict__ z,
                                                                         const float* __restrict__ ps, const float* __restrict__ pq, const float* __restrict__ pq2,
                                                                         int nrows, int stride, const float* __restrict__ mean,
                                                                         const float* __restrict__ invstd, const float* __restrict__ gamma, float inv_m,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dgamma2,
                                                                         float* __restrict__ dbeta2, void* __restrict__ dz, void* dres, int dres_accumulate,
                                                                         int C, long long rows, int src_is_mask) {
    constexpr int VW = BF16 ? 8 : 4, LPS = SLAB / VW, RLN = FOLD_THREADS / LPS, U = BF16 ? 2 : 4;
    const unsigned char* mk = reinterpret_cast<const unsigned char*>(relu_src);      // src_is_mask (VW = 8 only): one byte per 8 channels
    __shared__ float st[2][SLAB];
    const int slab = blockIdx.x, stripe = blockIdx.y, stripes = gridDim.y;
    const int c0 = slab * SLAB, tid = threadIdx.x, qa = tid % LPS, rl = tid / LPS;
    const int c = c0 + VW * qa, CV = C / VW, cv = c / VW;
    const bool live = c < C;
    const long long step = (long long)stripes * RLN;
    long long r = (long long)stripe * RLN + rl;
    // first U rows (dy, z, ReLU source) and the per-channel constants requested before the fold
    float g4[U][VW], yy[U][VW], zz[U][VW];
    float mu[VW], is[VW], ga[VW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long row = r + u * step;
        const bool ok = live && row < rows;
        const long long i = row * CV + cv;
#pragma unroll
        for (int e = 0; e < VW; ++e) { g4[u][e] = 0.f; zz[u][e] = 0.f; yy[u][e] = 0.f; }
        if (ok) { ldn<G16, VW>(dy, i, g4[u]); ldn<BF16, VW>(z, i, zz[u]); }
        if (ok && relu_src) {
            if (src_is_mask) yy[u][0] = __uint_as_float((unsigned)mk[i]); else ldn<BF16, VW>(relu_src, i, yy[u]);
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) { mu[e] = 0.f; is[e] = 0.f; ga[e] = 0.f; }
    if (live) { ldn<false, VW>(mean, cv, mu); ldn<false, VW>(invstd, cv, is); ldn<false, VW>(gamma, cv, ga); }
    double s0, s1;
    if (fold_slab64(ps, pq, nrows, stride, C, c0, s0, s1)) {
        const float db = (float)s0, dg = (float)s1;
        st[0][tid] = db; st[1][tid] = dg;
        if (stripe == 0) { dbeta[c0 + tid] = db; dgamma[c0 + tid] = dg; }
    }
    if (pq2 && stripe == 0) {                             // (uniform per workgroup)
        double t0, t1;
        if (fold_slab64(ps, pq2, nrows, stride, C, c0, t0, t1)) { dbeta2[c0 + tid] = (float)t0; dgamma2[c0 + tid] = (float)t1; }
    }
    __syncthreads();
    if (!live) return;
    float db[VW], dg[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { db[e] = st[0][VW * qa + e]; dg[e] = st[1][VW * qa + e]; }
    auto one = [&](float (&g)[VW], const float (&ys)[VW], const float (&zs)[VW], long long row) {
        if (relu_src) {
            if (src_is_mask) {
                const unsigned m = __float_as_uint(ys[0]);
#pragma unroll
                for (int e = 0; e < VW; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) g[e] = ys[e] > 0.f ? g[e] : 0.f;
            }
        }
        float o[VW];
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const int k = 4 * h;
            const f32x4 t = bn_bwd_elem(f32x4{g[k], g[k + 1], g[k + 2], g[k + 3]}, f32x4{zs[k], zs[k + 1], zs[k + 2], zs[k + 3]},
                                        f32x4{mu[k], mu[k + 1], mu[k + 2], mu[k + 3]}, f32x4{is[k], is[k + 1], is[k + 2], is[k + 3]},
                                        f32x4{ga[k], ga[k + 1], ga[k + 2], ga[k + 3]}, f32x4{dg[k], dg[k + 1], dg[k + 2], dg[k + 3]},
                                        f32x4{db[k], db[k + 1], db[k + 2], db[k + 3]}, inv_m);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[k + e] = t[e];
        }
        stn<BF16, VW>(dz, row * CV + cv, o);
        if (dres) {
            if (dres_accumulate) {
                float t[VW];
                ldn<G16, VW>(dres, row * CV + cv, t);
#pragma unroll
                for (int e = 0; e < VW; ++e) t[e] += g[e];
                stn<G16, VW>(dres, row * CV + cv, t);
            } else stn<G16, VW>(dres, row * CV + cv, g);
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (r + u * step < rows) one(g4[u], yy[u], zz[u], r + u * step);
    r += U * step;
    for (; r + (U - 1) * step < rows; r += U * step) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = (r + u * step) * CV + cv;
            ldn<G16, VW>(dy, i, g4[u]);
            ldn<BF16, VW>(z, i, zz[u]);
            if (relu_src) {
                if (src_is_mask) yy[u][0] = __uint_as_float((unsigned)mk[i]); else ldn<BF16, VW>(relu_src, i, yy[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(g4[u], yy[u], zz[u], r + u * step);
    }
    for (; r < rows; r += step) {
        const long long i = r * CV + cv;
        ldn<G16, VW>(dy, i, g4[0]);
        ldn<BF16, VW>(z, i, zz[0]);
        if (relu_src) {
            if (src_is_mask) yy[0][0] = __uint_as_float((unsigned)mk[i]); else ldn<BF16, VW>(relu_src, i, yy[0]);
        }
        one(g4[0], yy[0], zz[0], r);
    }
}

// SyncBatchNorm halves: per-rank (sum, sum of squares) kept in fp64 so that the cross-rank SUM is order-insensitive to ~1e-16
__global__ void pair_sum_final_f64_kernel(const double* __restrict__ part, int nblk, int C, double* __restrict__ sums) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s0, s1;
    fold_partials(part, nblk, C, c, s0, s1);
    if ((threadIdx.x & 63) != 0) return;
    sums[2 * c] = s0;
    sums[2 * c + 1] = s1;
}

// y = [relu]( (z - mean) * invstd * gamma + beta [+ res] ); a thread owns VW channels of a row (bf16: 8 = 16 bytes when c % 8 == 0)
template <bool BF16, int VW>
__global__ void bn_apply_kernel(const void* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, const void* __restrict__ res,
                                void* __restrict__ y, int CV, int relu, long long total, unsigned char* __restrict__ mask) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        float mu[VW], is[VW], g[VW], b[VW], x[VW], r[VW];
        ldn<false, VW>(mean, cv, mu); ldn<false, VW>(invstd, cv, is); ldn<false, VW>(gamma, cv, g); ldn<false, VW>(beta, cv, b);
        ldn<BF16, VW>(z, i, x);
        if (res) ldn<BF16, VW>(res, i, r);
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const int o = 4 * h;
            const f32x4 t = bn_fwd_elem(f32x4{x[o], x[o + 1], x[o + 2], x[o + 3]}, f32x4{mu[o], mu[o + 1], mu[o + 2], mu[o + 3]},
                                        f32x4{is[o], is[o + 1], is[o + 2], is[o + 3]}, f32x4{g[o], g[o + 1], g[o + 2], g[o + 3]},
                                        f32x4{b[o], b[o + 1], b[o + 2], b[o + 3]});
#pragma unroll
            for (int e = 0; e < 4; ++e) x[o + e] = t[e];
        }
        if (res) {
#pragma unroll
            for (int e = 0; e < VW; ++e) x[e] += r[e];
        }
        if (relu) {
            if constexpr (VW == 8) {
                if (mask) mask[i] = relu_bits8(x);
            }
#pragma unroll
            for (int e = 0; e < VW; ++e) x[e] = x[e] > 0.f ? x[e] : 0.f;
        }
        stn<BF16, VW>(y, i, x);
    }
}

// The same with the statistics still as (sum, sum of squares) per channel in fp64 ([c][2], what the SyncBatchNorm all-reduce leaves): every
// thread finalises the 4 channels it works on exactly as bn_stats_final_kernel does (its channel chunk only changes when the grid stride
// is not a multiple of C / 4), and the threads of the first row also write mean / invstd (saved for backward) and the running statistics -
// sp_bn_train_finalize's launch on the SyncBatchNorm critical chain disappears into the consumer
template <bool BF16>
__global__ void bn_apply_sums_kernel(const void* __restrict__ z, const double* __restrict__ sums, double M, float eps, float momentum,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, const void* __restrict__ res, void* __restrict__ y,
                                     int C4, int relu, long long total, float* __restrict__ mean, float* __restrict__ invstd,
                                     float* __restrict__ run_mean, float* __restrict__ run_var) {
    int have = -1;
    f32x4 mu, is, g, b;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        if (c4 != have) {
            have = c4;
            g = reinterpret_cast<const f32x4*>(gamma)[c4];
            b = reinterpret_cast<const f32x4*>(beta)[c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * c4 + e;
                const double m = sums[2 * c] / M;
                double var = sums[2 * c + 1] / M - m * m;
                if (var < 0) var = 0;
                mu[e] = (float)m;
                is[e] = (float)(1.0 / sqrt(var + (double)eps));
                if (i < C4) {                     // the first row's thread of this chunk publishes the layer's statistics
                    mean[c] = mu[e];
                    invstd[c] = is[e];
                    if (run_mean) {
                        const double unb = M > 1 ? var * M / (M - 1) : var;
                        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + (double)momentum * m);
                        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + (double)momentum * unb);
                    }
                }
            }
        }
        f32x4 v = bn_fwd_elem(ld4<BF16>(z, i), mu, is, g, b);
        if (res) { const f32x4 r = ld4<BF16>(res, i); v += r; }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        st4<BF16>(y, i, v);
    }
}

// g = dy * (y > 0);  dz = gamma*invstd * (g - dbeta/M - xhat * dgamma/M);  dres (+)= g
template <bool BF16, bool G16, int VW>   // BF16: z, relu_src, dz;  G16: dy, dres;  VW channels per thread
__global__ void bn_bwd_apply_kernel(const void* __restrict__ dy, const void* __restrict__ relu_src, const void* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta, float inv_m, void* __restrict__ dz,
                                    void* dres, int dres_accumulate, int CV, long long total, int src_is_mask) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        float mu[VW], is[VW], ga[VW], dg[VW], db[VW], g[VW], zz[VW], o[VW];
        ldn<false, VW>(mean, cv, mu); ldn<false, VW>(invstd, cv, is); ldn<false, VW>(gamma, cv, ga);
        ldn<false, VW>(dgamma, cv, dg); ldn<false, VW>(dbeta, cv, db);
        ldn<G16, VW>(dy, i, g);
        if (relu_src && src_is_mask) {
            const unsigned m = reinterpret_cast<const unsigned char*>(relu_src)[i];
#pragma unroll
            for (int e = 0; e < VW; ++e) g[e] = ((m >> e) & 1u) ? g[e] : 0.f;
        } else if (relu_src) {
            float yv[VW];
            ldn<BF16, VW>(relu_src, i, yv);
#pragma unroll
            for (int e = 0; e < VW; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
        }
        ldn<BF16, VW>(z, i, zz);
#pragma unroll
        for (int h = 0; h < VW / 4; ++h) {
            const int k = 4 * h;
            const f32x4 t = bn_bwd_elem(f32x4{g[k], g[k + 1], g[k + 2], g[k + 3]}, f32x4{zz[k], zz[k + 1], zz[k + 2], zz[k + 3]},
                                        f32x4{mu[k], mu[k + 1], mu[k + 2], mu[k + 3]}, f32x4{is[k], is[k + 1], is[k + 2], is[k + 3]},
                                        f32x4{ga[k], ga[k + 1], ga[k + 2], ga[k + 3]}, f32x4{dg[k], dg[k + 1], dg[k + 2], dg[k + 3]},
                                        f32x4{db[k], db[k + 1], db[k + 2], db[k + 3]}, inv_m);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[k + e] = t[e];
        }
        stn<BF16, VW>(dz, i, o);
        if (dres) {
            if (dres_accumulate) {
                float t[VW];
                ldn<G16, VW>(dres, i, t);
#pragma unroll
                for (int e = 0; e < VW; ++e) t[e] += g[e];
                stn<G16, VW>(dres, i, t);
            } else stn<G16, VW>(dres, i, g);
        }
    }
}

// dx[b,iy,ix,:] = sum over the (<= 4) windows that contain (iy,ix) and whose FIRST maximum (row-major scan, as
// nn.MaxPool2d) is this pixel, of dy[window].  Gather form: deterministic, no atomics.
template <bool BF16, bool G16>   // BF16: pool input x;  G16: dy, dx
__global__ void maxpool3x3s2_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, void* __restrict__ dx, int H, int W,
                                        int C4, int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const long long b = r / H;
        const f32x4 me = ld4<BF16>(x, i);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows oy with iy in [2oy-1, 2oy+1]
        for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                // is (iy,ix) the first max of window (oy,ox)?  per channel
                bool win[4] = {true, true, true, true};
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        if (yy == iy && xx == ix) continue;
                        const f32x4 o = ld4<BF16>(x, ((b * H + yy) * W + xx) * C4 + c);
                        const bool before = (yy < iy) || (yy == iy && xx < ix);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // an earlier element wins ties; a later one must be strictly greater (NaN: torch picks NaN; ignored)
                            if (before ? (o[e] >= me[e]) : (o[e] > me[e])) win[e] = false;
                        }
                    }
                }
                const f32x4 g = ld4<G16>(dy, ((b * Ho + oy) * Wo + ox) * C4 + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (win[e]) acc[e] += g[e];
            }
        }
        st4<G16>(dx, i, acc);
    }
}

// Training-time nn.MaxPool2d(3,2,1): also records WHICH tap (ky*3+kx, first maximum in row-major scan as torch) produced each
// output, one byte per element, so that backward is a gather of <= 4 (index, dy) pairs per input pixel instead of 36 compares.
template <bool BF16>
__global__ void maxpool3x3s2_idx_kernel(const void* __restrict__ x, void* __restrict__ y, unsigned int* __restrict__ idx, int H, int W, int C4,
                                        int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned int tap[4] = {4u, 4u, 4u, 4u};                 // the centre tap is always inside the image
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = oy * 2 - 1 + ky;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = ox * 2 - 1 + kx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const f32x4 v = ld4<BF16>(x, ((b * H + yy) * W + xx) * C4 + c);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; tap[e] = (unsigned)(ky * 3 + kx); }   // torch: (val > max) || isnan(val)
            }
        }
        st4<BF16>(y, i, best);
        idx[i] = tap[0] | (tap[1] << 8) | (tap[2] << 16) | (tap[3] << 24);
    }
}

template <bool G16>
__global__ void maxpool3x3s2_bwd_idx_kernel(const unsigned int* __restrict__ idx, const void* __restrict__ dy, void* __restrict__ dx, int H, int W,
                                            int C4, int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const long long b = r / H;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {        // windows oy with iy in [2oy-1, 2oy+1]
            if (oy >= Ho) continue;
            const unsigned ky = (unsigned)(iy - (2 * oy - 1));
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const unsigned mine = ky * 3u + (unsigned)(ix - (2 * ox - 1));
                const long long o = ((b * Ho + oy) * Wo + ox) * C4 + c;
                const unsigned int t = idx[o];
                const f32x4 g = ld4<G16>(dy, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (((t >> (8 * e)) & 0xffu) == mine) acc[e] += g[e];
            }
        }
        st4<G16>(dx, i, acc);
    }
}

// ---- the ResNet stem in training: bn1 + ReLU + MaxPool2d(3,2,1) as ONE forward pass and TWO backward passes (round 4) ----------------------
// relu(bn1(z)) feeds the pooling only (pose_resnet_dconv.py:251-256): nothing reads it as a tensor, so the forward pass applies the BatchNorm
// map to the nine taps of a window, pools, and writes the pooled map + the winning tap per element (as sp_maxpool3x3s2_idx_nhwc) - the 128 x 96
// map is neither written nor read back.  Backward: a pooled gradient reaches exactly one stem pixel, so d beta = sum g and d gamma = sum g * xhat
// run over the POOLED grid (4x fewer elements; xhat and the ReLU mask are re-formed from z at the winner), and dz gathers its <= 4 windows as
// sp_maxpool3x3s2_bwd_idx_nhwc does - the pooling's input gradient is never materialised either.
template <bool BF16>
__global__ void bn_apply_maxpool_kernel(const void* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, void* __restrict__ y,
                                        unsigned int* __restrict__ idx, int H, int W, int C4, int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c], is = reinterpret_cast<const f32x4*>(invstd)[c];
        const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c], be = reinterpret_cast<const f32x4*>(beta)[c];
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned int tap[4] = {4u, 4u, 4u, 4u};                 // the centre tap is always inside the image
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = oy * 2 - 1 + ky;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = ox * 2 - 1 + kx;
                if ((unsigned)xx >= (unsigned)W) continue;
                f32x4 v = bn_fwd_elem(ld4<BF16>(z, ((b * H + yy) * W + xx) * C4 + c), mu, is, ga, be);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] > 0.f ? v[e] : 0.f;
                    if (BF16) v[e] = (float)(__bf16)v[e];       // the value nn.MaxPool2d would see: the stored bf16 activation
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; tap[e] = (unsigned)(ky * 3 + kx); }   // torch: (val > max) || isnan(val)
                }
            }
        }
        st4<BF16>(y, i, best);
        idx[i] = tap[0] | (tap[1] << 8) | (tap[2] << 16) | (tap[3] << 24);
    }
}

// sums over the pooled grid: s0 = sum g, s1 = sum g * xhat with g = dyp * (bn(z) > 0) at the window's winner.  Same block structure and
// fixed-order folds as channel_reduce_kernel (part: [gridDim.x][C][2] doubles -> pair_sum_final_kernel).
template <bool BF16, bool G16>
__global__ __launch_bounds__(256) void stem_pool_bwd_reduce_kernel(const void* __restrict__ dyp, const unsigned int* __restrict__ idx, const void* __restrict__ z,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int H, int W, int Ho,
                                                                   int Wo, int M, int C, double* __restrict__ part) {
    const int C4 = C >> 2;
    const int lanes_c = C4 < 256 ? C4 : 256;
    const int rows_par = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, tr = threadIdx.x / lanes_c;
    __shared__ double sm[256 * 8];
    for (int c4 = tc; c4 < C4; c4 += lanes_c) {
        double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
        const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
        const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4], be = reinterpret_cast<const f32x4*>(beta)[c4];
        if (tr < rows_par) {
            // four pooled rows per trip (round 5): their gradient / winner loads go out together, then the sixteen winner gathers of z, then the
            // sums in row order - the per-row chain (index load -> gather -> use) was two dependent round trips per 4 channels
            constexpr int U = 4;
            const long long stride = (long long)gridDim.x * rows_par;
            const float* zf = reinterpret_cast<const float*>(z);
            const __bf16* zh = reinterpret_cast<const __bf16*>(z);
            for (long long r0 = (long long)blockIdx.x * rows_par + tr; r0 < M; r0 += U * stride) {
                f32x4 g[U];
                unsigned t[U];
                float zz[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long long r = r0 + u * stride;
                    g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    t[u] = 0x04040404u;
                    if (r < M) { g[u] = ld4<G16>(dyp, r * C4 + c4); t[u] = idx[r * C4 + c4]; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long long r = r0 + u * stride;
                    const int ox = (int)(r % Wo);
                    const long long q = r / Wo;
                    const int oy = (int)(q % Ho);
                    const long long b = q / Ho;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int tap = (int)((t[u] >> (8 * e)) & 0xffu), ky = (tap * 11) >> 5, kx = tap - 3 * ky;   // tap / 3 for tap < 9
                        const long long zi = (((b * H + (oy * 2 - 1 + ky)) * W + (ox * 2 - 1 + kx)) * C4 + c4) * 4 + e;
                        zz[u][e] = 0.f;
                        if (r < M) zz[u][e] = BF16 ? (float)zh[zi] : zf[zi];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (r0 + u * stride >= M) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z1 = zz[u][e];
                        f32x4 v4 = {z1, z1, z1, z1};
                        v4 = bn_fwd_elem(v4, f32x4{mu[e], mu[e], mu[e], mu[e]}, f32x4{is[e], is[e], is[e], is[e]}, f32x4{ga[e], ga[e], ga[e], ga[e]},
                                         f32x4{be[e], be[e], be[e], be[e]});
                        const float ge = v4[0] > 0.f ? g[u][e] : 0.f;
                        s0[e] += (double)ge;
                        s1[e] += (double)ge * (double)((z1 - mu[e]) * is[e]);
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sm[threadIdx.x * 8 + e] = s0[e]; sm[threadIdx.x * 8 + 4 + e] = s1[e]; }
        __syncthreads();
        if (tr == 0) {
            for (int k = 1; k < rows_par; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) sm[tc * 8 + e] += sm[(k * lanes_c + tc) * 8 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part[((size_t)blockIdx.x * C + c4 * 4 + e) * 2 + 0] = sm[tc * 8 + e];
                part[((size_t)blockIdx.x * C + c4 * 4 + e) * 2 + 1] = sm[tc * 8 + 4 + e];
            }
        }
        __syncthreads();
    }
}

// dz of the stem conv: g at a stem pixel = the pooled gradients of the (<= 4) windows it won, through the ReLU mask re-formed from z.
// VW channels per thread: 4, or 8 where every operand is bf16 (round 5: 16-byte accesses - half the memory instructions per byte; the element
// maps run on the same 4-channel groups, so the bits do not move).
template <bool BF16, bool G16, int VW>
__global__ void stem_pool_bwd_apply_kernel(const void* __restrict__ dyp, const unsigned int* __restrict__ idx, const void* __restrict__ z,
                                           const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, const float* __restrict__ dgamma, const float* __restrict__ dbeta, float inv_m,
                                           void* __restrict__ dz, int H, int W, int CV, int Ho, int Wo, long long total) {
    constexpr int Q = VW / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV);
        long long r = i / CV;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const long long b = r / H;
        float zz[VW], acc[VW];
        ldn<BF16, VW>(z, i, zz);
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[e] = 0.f;
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {        // windows oy with iy in [2oy-1, 2oy+1]
            if (oy >= Ho) continue;
            const unsigned ky = (unsigned)(iy - (2 * oy - 1));
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const unsigned mine = ky * 3u + (unsigned)(ix - (2 * ox - 1));
                const long long o = ((b * Ho + oy) * Wo + ox) * CV + c;
                unsigned int t[Q];
#pragma unroll
                for (int h = 0; h < Q; ++h) t[h] = idx[o * Q + h];
                float g[VW];
                ldn<G16, VW>(dyp, o, g);
#pragma unroll
                for (int e = 0; e < VW; ++e) if (((t[e >> 2] >> (8 * (e & 3))) & 0xffu) == mine) acc[e] += g[e];
            }
        }
        float o8[VW];
#pragma unroll
        for (int h = 0; h < Q; ++h) {
            const int c4 = c * Q + h;
            const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
            const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4], be = reinterpret_cast<const f32x4*>(beta)[c4];
            const f32x4 dg = reinterpret_cast<const f32x4*>(dgamma)[c4], db = reinterpret_cast<const f32x4*>(dbeta)[c4];
            const f32x4 z4 = {zz[4 * h], zz[4 * h + 1], zz[4 * h + 2], zz[4 * h + 3]};
            f32x4 a4 = {acc[4 * h], acc[4 * h + 1], acc[4 * h + 2], acc[4 * h + 3]};
            const f32x4 v = bn_fwd_elem(z4, mu, is, ga, be);
#pragma unroll
            for (int e = 0; e < 4; ++e) a4[e] = v[e] > 0.f ? a4[e] : 0.f;
            const f32x4 d4 = bn_bwd_elem(a4, z4, mu, is, ga, dg, db, inv_m);
#pragma unroll
            for (int e = 0; e < 4; ++e) o8[4 * h + e] = d4[e];
        }
        stn<BF16, VW>(dz, i, o8);
    }
}

// [B,C,H,W] fp32 -> [B,H,W,Cpad] (fp32 or bf16), channels >= C zero: the loss gradient d loss / d heat-map in the layout (and
// K-tile padding) the final layer's dgrad / wgrad launches read.  One thread per (pixel, 4 output channels).
template <bool BF16OUT>
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, void* __restrict__ y, int C, int hw, int Cp4, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % Cp4);
        const long long r = i / Cp4;
        const long long b = r / hw;
        const int pix = (int)(r - b * hw);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 * 4 + e;
            if (c < C) v[e] = x[(b * C + c) * hw + pix];
        }
        st4<BF16OUT>(y, i, v);
    }
}

// torch.optim.Adam (no amsgrad, weight_decay 0) on flat buffers; g is multiplied by grad_scale first (1/world after a SUM all-reduce)
struct AdamScalars { float step_size, b1, b2, omb1, omb2, eps, bc2_sqrt, grad_scale; };

// DEV: the eight scalars come from device memory (sp_adam_set_scalars wrote them) instead of the kernel arguments - the launch is then
// the same every step, which is what lets a captured train step (hipGraph) be replayed
template <bool DEV>
__global__ void adam_kernel(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long long n4,
                            AdamScalars a, const AdamScalars* __restrict__ dev) {
    if constexpr (DEV) a = *dev;
    const float step_size = a.step_size, b1 = a.b1, b2 = a.b2, omb1 = a.omb1, omb2 = a.omb2, eps = a.eps, bc2_sqrt = a.bc2_sqrt, grad_scale = a.grad_scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * grad_scale;
            mm[e] = mm[e] * b1 + omb1 * gr;                    // exp_avg.mul_(beta1).add_(grad, alpha=1-beta1)
            vv[e] = vv[e] * b2 + omb2 * gr * gr;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps; // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
            pp[e] = pp[e] - step_size * (mm[e] / denom);       // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

__global__ void adam_set_scalars_kernel(AdamScalars a, AdamScalars* __restrict__ dst) { *dst = a; }

// generic 4-D gather-copy: dst[((a*D1 + b)*D2 + c)*D3 + d] = (in range) ? src[a*s0 + b*s1 + c*s2 + d*s3 + base] : 0
struct Permute4 {
    int d[4];        // destination extents
    long long s[4];  // source strides (elements) per destination index
    int lim[4];      // valid extent per index (>= -> zero fill)
    long long base;
    long long dst_off;
};
template <bool BF16OUT>
__global__ void permute4_kernel(const float* __restrict__ src, void* __restrict__ dst, const Permute4 pm, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int i3 = (int)(r % pm.d[3]); r /= pm.d[3];
        const int i2 = (int)(r % pm.d[2]); r /= pm.d[2];
        const int i1 = (int)(r % pm.d[1]);
        const int i0 = (int)(r / pm.d[1]);
        float v = 0.f;
        if (i0 < pm.lim[0] && i1 < pm.lim[1] && i2 < pm.lim[2] && i3 < pm.lim[3])
            v = src[pm.base + i0 * pm.s[0] + i1 * pm.s[1] + i2 * pm.s[2] + i3 * pm.s[3]];
        if constexpr (BF16OUT) reinterpret_cast<__bf16*>(dst)[pm.dst_off + i] = (__bf16)v;
        else reinterpret_cast<float*>(dst)[pm.dst_off + i] = v;
    }
}

// All pack jobs of a network in ONE launch: job table in device memory, blockIdx.y = job, grid-stride inside the job.
struct PackJobDev {
    int d[4];
    long long s[4];
    int lim[4];
    long long base;      // element offset into src
    long long dst_ptr;   // device address of the destination buffer (already offset)
    long long total;
    int dst_bf16;
    int pad;
    int tap0;            // walk 3: offset (<= 0) from the job's `base` to the first tap of an (i0, i3) pair's source block
    int tile0;           // walk 3: i0 values per LDS tile (32 i3 values x tile0 blocks of s[0] floats fit the 40 KB buffer)
};
// `pad` of a job selects how its elements are walked (the result is the same gather-copy; only the access pattern differs):
//   0  destination order, one element per thread: fine when the fastest destination index is also contiguous in the source;
//   1  multi-tap filters ([o][c][taps] sources): a thread takes one (i0, i3) pair and loops over the (i1, i2) taps - its source elements
//      are one run of taps (36 / 64 bytes: a sector or two), and consecutive threads write consecutive destination elements per tap.
//      Destination order would read 4 bytes per 36 / 64 / 2,304-byte stride: measured 567 MB of HBM traffic per repack launch for
//      ~70 MB of compulsory bytes (2.3 GB of the bf16 train step's 20.5);
//   2  1x1 filters packed transposed ([o][c] -> [c][o]: d1 = d2 = 1, the source contiguous along i0): 32x32 tiles through LDS, both
//      the loads and the stores are whole 128-byte rows;
//   3  (round 5) multi-tap filters whose FASTEST destination index is the source's slowest (the dgrad packs of k > 1 convs,
//      Wd[c][taps][o] <- W[o][c][taps], and the four phase packs of a transposed conv): walk 1 gives every thread a 36 / 64-byte run in a
//      row of its own - 2.3-3.5x the compulsory sectors (the repack moved 680 MB per step for 408 MB of compulsory bytes).  Here a tile of
//      32 i3 values x tile0 i0 values is read as 32 contiguous runs of tile0 * s[0] floats (the (i0, taps) blocks of one i3 are adjacent
//      in the source), parked in LDS with an odd row pitch, and written with i3 fastest: 64 / 128-byte runs both ways.
// Walk 3 lives in a kernel of its own (round 6): its 40 KB tile buffer would otherwise be reserved by EVERY repack launch - 3 workgroups
// per CU instead of 8 for the bandwidth-bound walks 0-2 that the default repack uses.  Jobs of other walks in a tiled launch, and walk-3
// jobs in a plain launch (treated as walk 1: the same gather-copy), stay correct.
template <bool TILED>
__global__ void permute4_batched_kernel(const float* __restrict__ src, const PackJobDev* __restrict__ jobs) {
    const PackJobDev pm = jobs[blockIdx.y];
    auto put = [&](long long i, float v) __attribute__((always_inline)) {
        if (pm.dst_bf16) reinterpret_cast<__bf16*>(pm.dst_ptr)[i] = (__bf16)v;
        else reinterpret_cast<float*>(pm.dst_ptr)[i] = v;
    };
    if constexpr (TILED) {
      if (pm.pad == 3) {
          constexpr int T3 = 32, BUF = 10240;
          __shared__ float buf[BUF];
          __shared__ int tap_src[32], tap_dst[32];           // per destination tap: offset inside the source block (-1: zero fill), (i1 * d2 + i2) * d3
          const int span = (int)pm.s[0], T0 = pm.tile0, run = T0 * span, pitch = run | 1;     // (host: 32 * (T0 * span | 1) <= BUF)
          const int taps = pm.d[1] * pm.d[2];                 // (host: <= 32)
          if (threadIdx.x < taps) {
              const int i1 = threadIdx.x / pm.d[2], i2 = threadIdx.x - i1 * pm.d[2];
              tap_src[threadIdx.x] = (i1 < pm.lim[1] && i2 < pm.lim[2]) ? (int)(i1 * pm.s[1] + i2 * pm.s[2]) - pm.tap0 : -1;
              tap_dst[threadIdx.x] = (i1 * pm.d[2] + i2) * pm.d[3];
          }
          const int t0n = (pm.d[0] + T0 - 1) / T0, t3n = (pm.d[3] + T3 - 1) / T3;
          const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
          const int r3s = threadIdx.x & (T3 - 1), sub = threadIdx.x >> 5, nsub = blockDim.x >> 5;
          for (int tl = blockIdx.x; tl < t0n * t3n; tl += gridDim.x) {
              const int a0 = (tl % t0n) * T0, a3 = (tl / t0n) * T3;
              // load: wave w takes rows w, w + 4, ...; a row is one contiguous run of the source (no per-element index arithmetic)
              const int kmax = max(0, min(T0, pm.lim[0] - a0)) * span;      // floats of the run that belong to valid i0
              for (int r3 = wave; r3 < T3; r3 += nw) {
                  const bool row_ok = a3 + r3 < pm.lim[3];
                  const float* row = src + pm.base + pm.tap0 + (long long)a0 * pm.s[0] + (long long)(a3 + r3) * pm.s[3];
                  for (int k = lane; k < run; k += 64) buf[r3 * pitch + k] = (row_ok && k < kmax) ? row[k] : 0.f;
              }
              __syncthreads();
              // store: thread = (i3 lane, one of 8 (i0, tap) walkers); (cl, tp) advance without divisions
              const int i3 = a3 + r3s;
              int cl = 0, tp = sub;
              while (tp >= taps) { tp -= taps; ++cl; }
              for (; cl < T0; ) {
                  const int i0 = a0 + cl;
                  if (i0 < pm.d[0] && i3 < pm.d[3]) {
                      const int so = tap_src[tp];
                      const float v = (so >= 0 && i0 < pm.lim[0] && i3 < pm.lim[3]) ? buf[r3s * pitch + cl * span + so] : 0.f;
                      put((long long)i0 * taps * pm.d[3] + tap_dst[tp] + i3, v);
                  }
                  tp += nsub;
                  while (tp >= taps) { tp -= taps; ++cl; }
              }
              __syncthreads();
          }
        return;
      }
    }
    if (pm.pad == 1 || pm.pad == 3) {
        const long long pairs = (long long)pm.d[0] * pm.d[3];
        const int taps = pm.d[1] * pm.d[2];
        for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < pairs; t += (long long)gridDim.x * blockDim.x) {
            const int i3 = (int)(t % pm.d[3]), i0 = (int)(t / pm.d[3]);
            const bool ok03 = i0 < pm.lim[0] && i3 < pm.lim[3];
            const long long sb = pm.base + i0 * pm.s[0] + i3 * pm.s[3];
            for (int tp = 0; tp < taps; ++tp) {
                const int i1 = tp / pm.d[2], i2 = tp - i1 * pm.d[2];
                float v = 0.f;
                if (ok03 && i1 < pm.lim[1] && i2 < pm.lim[2]) v = src[sb + i1 * pm.s[1] + i2 * pm.s[2]];
                put(((long long)(i0 * pm.d[1] + i1) * pm.d[2] + i2) * pm.d[3] + i3, v);
            }
        }
        return;
    }
    if (pm.pad == 2) {
        __shared__ float tile[32][33];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 256 threads: 8 rows of 32
        const int t0n = (pm.d[0] + 31) / 32, t3n = (pm.d[3] + 31) / 32;
        for (int tl = blockIdx.x; tl < t0n * t3n; tl += gridDim.x) {
            const int a0 = (tl % t0n) * 32, a3 = (tl / t0n) * 32;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                   // load: tx along i0 (contiguous in the source), rows along i3
                const int i3 = a3 + ty + 8 * r, i0 = a0 + tx;
                float v = 0.f;
                if (i0 < pm.lim[0] && i3 < pm.lim[3] && 0 < pm.lim[1] && 0 < pm.lim[2]) v = src[pm.base + i0 * pm.s[0] + i3 * pm.s[3]];
                tile[ty + 8 * r][tx] = v;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                   // store: tx along i3 (contiguous in the destination)
                const int i0 = a0 + ty + 8 * r, i3 = a3 + tx;
                if (i0 < pm.d[0] && i3 < pm.d[3]) put((long long)i0 * pm.d[3] + i3, tile[tx][ty + 8 * r]);
            }
            __syncthreads();
        }
        return;
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pm.total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int i3 = (int)(r % pm.d[3]); r /= pm.d[3];
        const int i2 = (int)(r % pm.d[2]); r /= pm.d[2];
        const int i1 = (int)(r % pm.d[1]);
        const int i0 = (int)(r / pm.d[1]);
        float v = 0.f;
        if (i0 < pm.lim[0] && i1 < pm.lim[1] && i2 < pm.lim[2] && i3 < pm.lim[3])
            v = src[pm.base + i0 * pm.s[0] + i1 * pm.s[1] + i2 * pm.s[2] + i3 * pm.s[3]];
        put(i, v);
    }
}

}  // namespace

extern "C" int sp_permute4_batched(const float* src, const void* jobs_device, int n_jobs, int blocks_per_job, void* stream) {
    SP_REQUIRE(src && jobs_device && n_jobs > 0 && n_jobs <= 65535 && blocks_per_job > 0, "sp_permute4_batched: bad argument");
#ifdef SP_REPACK_AB_DIAG   // DIAGNOSTIC BUILD ONLY (same-box A/B of round 5's launch, whose every repack reserved walk 3's 40 KB tile; never the shipped library)
    hipLaunchKernelGGL(permute4_batched_kernel<true>, dim3(blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const PackJobDev*>(jobs_device));
#else
    hipLaunchKernelGGL(permute4_batched_kernel<false>, dim3(blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const PackJobDev*>(jobs_device));
#endif
    return sp_check_launch("permute4_batched_kernel");
}

extern "C" int sp_permute4_batched_tiled(const float* src, const void* jobs_device, int n_jobs, int blocks_per_job, void* stream) {
    SP_REQUIRE(src && jobs_device && n_jobs > 0 && n_jobs <= 65535 && blocks_per_job > 0, "sp_permute4_batched_tiled: bad argument");
    hipLaunchKernelGGL(permute4_batched_kernel<true>, dim3(blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const PackJobDev*>(jobs_device));
    return sp_check_launch("permute4_batched_kernel<tiled>");
}

extern "C" int sp_bn_train_stats_nhwc(const void* z, int bf16, int64_t rows, int c, float eps, float momentum, float* mean, float* invstd,
                                      float* running_mean, float* running_var, void* workspace, void* stream) {
    SP_REQUIRE(z && mean && invstd && workspace, "sp_bn_train_stats_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && c > 0 && c % 4 == 0 && rows < (1ll << 31) && ((c / 4) <= 256 ? 256 % 1 == 0 : (c / 4) % 256 == 0),
               "sp_bn_train_stats_nhwc: bad shape rows=%lld c=%d (c/4 must be <= 256 or a multiple of 256)", (long long)rows, c);
    SP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sp_bn_train_stats_nhwc: running stats come in pairs");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    if (bf16 & 1) hipLaunchKernelGGL((channel_reduce_kernel<0, true>), dim3(red_blocks(rows, c)), dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, (int)rows, c, part);
    else hipLaunchKernelGGL((channel_reduce_kernel<0, false>), dim3(red_blocks(rows, c)), dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, (int)rows, c, part);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, red_blocks(rows, c), c, (double)rows, eps, momentum, mean,
                       invstd, running_mean, running_var);
    return sp_check_launch("bn_train_stats");
}

extern "C" int sp_bn_train_partial_nhwc(const void* z, int bf16, int64_t rows, int c, double* sums, void* workspace, void* stream) {
    SP_REQUIRE(z && sums && workspace, "sp_bn_train_partial_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && c > 0 && c % 4 == 0 && rows < (1ll << 31), "sp_bn_train_partial_nhwc: bad shape rows=%lld c=%d", (long long)rows, c);
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    if (bf16 & 1) hipLaunchKernelGGL((channel_reduce_kernel<0, true>), dim3(red_blocks(rows, c)), dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, (int)rows, c, part);
    else hipLaunchKernelGGL((channel_reduce_kernel<0, false>), dim3(red_blocks(rows, c)), dim3(256), 0, s, z, nullptr, nullptr, nullptr, nullptr, (int)rows, c, part);
    hipLaunchKernelGGL(pair_sum_final_f64_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, red_blocks(rows, c), c, sums);
    return sp_check_launch("bn_train_partial");
}

extern "C" int sp_bn_train_finalize(const double* sums, int64_t total_rows, int c, float eps, float momentum, float* mean, float* invstd,
                                    float* running_mean, float* running_var, void* stream) {
    SP_REQUIRE(sums && mean && invstd, "sp_bn_train_finalize: null pointer");
    SP_REQUIRE(total_rows > 0 && c > 0, "sp_bn_train_finalize: bad shape");
    SP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sp_bn_train_finalize: running stats come in pairs");
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((c + 3) / 4), dim3(256), 0, (hipStream_t)stream, sums, 1, c, (double)total_rows, eps, momentum,
                       mean, invstd, running_mean, running_var);
    return sp_check_launch("bn_train_finalize");
}

extern "C" int sp_bn_train_stats_from_conv(const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride, int64_t rows, int c,
                                           float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                           void* stream) {
    SP_REQUIRE(stats_sum && stats_sumsq && mean && invstd, "sp_bn_train_stats_from_conv: null pointer");
    SP_REQUIRE(partial_rows > 0 && stride >= c && c > 0 && rows > 0, "sp_bn_train_stats_from_conv: bad shape");
    SP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sp_bn_train_stats_from_conv: running stats come in pairs");
    hipLaunchKernelGGL(bn_stats_from_conv_kernel, dim3((c + SFC - 1) / SFC), dim3(SFT), 0, (hipStream_t)stream, stats_sum, stats_sumsq, partial_rows, stride,
                       c, (double)rows, eps, momentum, mean, invstd, running_mean, running_var);
    return sp_check_launch("bn_stats_from_conv_kernel");
}

extern "C" int sp_bn_sums_from_conv(const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride, int c, double* sums,
                                    void* stream) {
    SP_REQUIRE(stats_sum && stats_sumsq && sums && partial_rows > 0 && stride >= c && c > 0, "sp_bn_sums_from_conv: bad argument");
    hipLaunchKernelGGL(bn_sums_from_conv_kernel, dim3((c + SFC - 1) / SFC), dim3(SFT), 0, (hipStream_t)stream, stats_sum, stats_sumsq, partial_rows, stride,
                       c, sums);
    return sp_check_launch("bn_sums_from_conv_kernel");
}

extern "C" int sp_bn_bwd_sums_from_conv(const float* sum_g, const float* sum_g_xhat, int partial_rows, int stride, int c, float* dgamma,
                                        float* dbeta, void* stream) {
    SP_REQUIRE(sum_g && sum_g_xhat && dgamma && dbeta && partial_rows > 0 && stride >= c && c > 0, "sp_bn_bwd_sums_from_conv: bad argument");
    hipLaunchKernelGGL(bn_bwd_sums_from_conv_kernel, dim3((c + SFC - 1) / SFC), dim3(SFT), 0, (hipStream_t)stream, sum_g, sum_g_xhat, partial_rows, stride,
                       c, dbeta, dgamma, nullptr, nullptr);
    return sp_check_launch("bn_bwd_sums_from_conv_kernel");
}

extern "C" int sp_bn_bwd_sums_from_conv_pair(const float* sum_g, const float* sum_g_xhat, const float* sum_g_xhat2, int partial_rows, int stride, int c,
                                             float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, void* stream) {
    SP_REQUIRE(sum_g && sum_g_xhat && sum_g_xhat2 && dgamma && dbeta && dgamma2 && dbeta2 && partial_rows > 0 && stride >= c && c > 0,
               "sp_bn_bwd_sums_from_conv_pair: bad argument");
    hipLaunchKernelGGL(bn_bwd_sums_from_conv_kernel, dim3((c + SFC - 1) / SFC, 2), dim3(SFT), 0, (hipStream_t)stream, sum_g, sum_g_xhat, partial_rows,
                       stride, c, dbeta, dgamma, nullptr, nullptr, sum_g_xhat2, dbeta2, dgamma2);
    return sp_check_launch("bn_bwd_sums_from_conv_kernel");
}

extern "C" int sp_bn_bwd_sums_from_conv2(const float* sum_g, const float* sum_g_xhat, int partial_rows, int stride, int c, float* dgamma, float* dbeta,
                                         float* dgamma_copy, float* dbeta_copy, void* stream) {
    SP_REQUIRE(sum_g && sum_g_xhat && dgamma && dbeta && dgamma_copy && dbeta_copy && partial_rows > 0 && stride >= c && c > 0,
               "sp_bn_bwd_sums_from_conv2: bad argument");
    hipLaunchKernelGGL(bn_bwd_sums_from_conv_kernel, dim3((c + SFC - 1) / SFC), dim3(SFT), 0, (hipStream_t)stream, sum_g, sum_g_xhat, partial_rows, stride,
                       c, dbeta, dgamma, dbeta_copy, dgamma_copy);
    return sp_check_launch("bn_bwd_sums_from_conv_kernel");
}

// Row stripes of the fused passes: enough workgroups of 1,024 threads to fill the chip twice over, each with >= 4 x 64 rows to stream (the
// prologue's fold is repeated per workgroup)
static int fold_stripes(long long rows, int slabs, int partial_rows) {
    long long want = (512 + slabs - 1) / slabs;
    static const long long min_rows = getenv("SP_FOLD_MIN_ROWS") ? atoll(getenv("SP_FOLD_MIN_ROWS")) : 256;          // (env: development knob)
    const long long cap = (rows + min_rows - 1) / min_rows;
    if (want > cap) want = cap;
    // every workgroup repeats the fold: (workgroups) x (partial rows) x 512 B of L2 reads.  Bounded to ~64 MB per launch: tensors with many
    // partial rows get fewer, fatter workgroups (each thread keeps four rows of every operand in flight, so a few hundred workgroups of
    // 1,024 threads still cover the HBM latency)
    static const long long budget = getenv("SP_FOLD_BUDGET") ? atoll(getenv("SP_FOLD_BUDGET")) : 131072;      // (workgroups x partial rows; env: development knob)
    const long long capf = budget / ((long long)partial_rows * slabs);
    if (want > capf) want = capf;
    return (int)(want < 1 ? 1 : want);
}

extern "C" int sp_bn_fold_apply_nhwc(const void* z, int bf16, const float* stats_sum, const float* stats_sumsq, int partial_rows, int stride,
                                     int64_t total_rows, float eps, float momentum, const float* gamma, const float* beta, const void* residual,
                                     void* y, int64_t rows, int c, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                                     void* relu_mask, void* stream) {
    SP_REQUIRE(z && stats_sum && stats_sumsq && gamma && beta && y && mean && invstd, "sp_bn_fold_apply_nhwc: null pointer");
    SP_REQUIRE(!relu_mask || ((bf16 & 1) && relu), "sp_bn_fold_apply_nhwc: the ReLU bit mask exists for bf16 tensors behind a ReLU");
    SP_REQUIRE(rows > 0 && total_rows >= rows && c > 0 && c % 4 == 0 && partial_rows > 0 && stride >= c && stride % 4 == 0 && rows < (1ll << 31),
               "sp_bn_fold_apply_nhwc: bad shape");
    SP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sp_bn_fold_apply_nhwc: running stats come in pairs");
    SP_REQUIRE(!(bf16 & 1) || c % 8 == 0, "sp_bn_fold_apply_nhwc: bf16 tensors are walked 8 channels at a time (c %% 8 == 0)");
    const int slabs = (c + SLAB - 1) / SLAB;
    const dim3 grid(slabs, fold_stripes(rows, slabs, partial_rows));
    if (bf16 & 1) hipLaunchKernelGGL(bn_fold_apply_kernel<true>, grid, dim3(FOLD_THREADS), 0, (hipStream_t)stream, z, stats_sum, stats_sumsq, partial_rows, stride,
                                     (double)total_rows, eps, momentum, gamma, beta, residual, y, c, relu, (long long)rows, mean, invstd, running_mean, running_var,
                                     reinterpret_cast<unsigned char*>(relu_mask));
    else hipLaunchKernelGGL(bn_fold_apply_kernel<false>, grid, dim3(FOLD_THREADS), 0, (hipStream_t)stream, z, stats_sum, stats_sumsq, partial_rows, stride,
                            (double)total_rows, eps, momentum, gamma, beta, residual, y, c, relu, (long long)rows, mean, invstd, running_mean, running_var,
                            (unsigned char*)nullptr);
    return sp_check_launch("bn_fold_apply_kernel");
}

extern "C" int sp_bn_fold_bwd_apply_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* sum_g, const float* sum_g_xhat,
                                         const float* sum_g_xhat2, int partial_rows, int stride, const float* mean, const float* invstd,
                                         const float* gamma, int64_t total_rows, int64_t rows, int c, float* dgamma, float* dbeta, float* dgamma2,
                                         float* dbeta2, void* dz, void* dres, int dres_accumulate, void* stream) {
    SP_REQUIRE(dy && z && sum_g && sum_g_xhat && mean && invstd && gamma && dgamma && dbeta && dz, "sp_bn_fold_bwd_apply_nhwc: null pointer");
    SP_REQUIRE(!sum_g_xhat2 || (dgamma2 && dbeta2), "sp_bn_fold_bwd_apply_nhwc: the second BatchNorm's gradients are missing");
    SP_REQUIRE(rows > 0 && total_rows >= rows && c > 0 && c % 4 == 0 && partial_rows > 0 && stride >= c && stride % 4 == 0 && rows < (1ll << 31),
               "sp_bn_fold_bwd_apply_nhwc: bad shape");
    const bool a16 = bf16 & 1, g16 = bf16 & 2;
    SP_REQUIRE(a16 || !g16, "sp_bn_fold_bwd_apply_nhwc: bf16 gradients with fp32 activations is not a supported mix");
    SP_REQUIRE(!a16 || c % 8 == 0, "sp_bn_fold_bwd_apply_nhwc: bf16 tensors are walked 8 channels at a time (c %% 8 == 0)");
    const int src_is_mask = (bf16 & 4) ? 1 : 0;
    SP_REQUIRE(!src_is_mask || (a16 && relu_src), "sp_bn_fold_bwd_apply_nhwc: a ReLU bit mask goes with bf16 activations");
    const int slabs = (c + SLAB - 1) / SLAB;
    const dim3 grid(slabs, fold_stripes(rows, slabs, partial_rows));
    hipStream_t s = (hipStream_t)stream;
#define SP_FBA(A, G)                                                                                                                         \
    hipLaunchKernelGGL((bn_fold_bwd_apply_kernel<A, G>), grid, dim3(FOLD_THREADS), 0, s, dy, relu_src, z, sum_g, sum_g_xhat, sum_g_xhat2, partial_rows, \
                       stride, mean, invstd, gamma, (float)(1.0 / (double)total_rows), dgamma, dbeta, dgamma2, dbeta2, dz, dres, dres_accumulate, c,  \
                       (long long)rows, src_is_mask)
    if (a16 && g16) SP_FBA(true, true);
    else if (a16) SP_FBA(true, false);
    else SP_FBA(false, false);
#undef SP_FBA
    return sp_check_launch("bn_fold_bwd_apply_kernel");
}

extern "C" int sp_bn_apply_nhwc(const void* z, int bf16, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const void* residual, void* y, int64_t rows, int c, int relu, void* relu_mask, void* stream) {
    SP_REQUIRE(!relu_mask || ((bf16 & 1) && relu && c % 8 == 0), "sp_bn_apply_nhwc: the ReLU bit mask exists for bf16 tensors (c %% 8 == 0) behind a ReLU");
    SP_REQUIRE(z && mean && invstd && gamma && beta && y, "sp_bn_apply_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && c > 0 && c % 4 == 0, "sp_bn_apply_nhwc: bad shape");
    if ((bf16 & 1) && c % 8 == 0) {
        const long long total = rows * (c / 8);
        hipLaunchKernelGGL((bn_apply_kernel<true, 8>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta, residual, y,
                           c / 8, relu, total, reinterpret_cast<unsigned char*>(relu_mask));
        return sp_check_launch("bn_apply_kernel");
    }
    const long long total = rows * (c / 4);
    if (bf16 & 1) hipLaunchKernelGGL((bn_apply_kernel<true, 4>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta,
                                     residual, y, c / 4, relu, total, (unsigned char*)nullptr);
    else hipLaunchKernelGGL((bn_apply_kernel<false, 4>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta,
                            residual, y, c / 4, relu, total, (unsigned char*)nullptr);
    return sp_check_launch("bn_apply_kernel");
}

extern "C" int sp_bn_apply_sums_nhwc(const void* z, int bf16, const double* sums, int64_t total_rows, float eps, float momentum, const float* gamma,
                                     const float* beta, const void* residual, void* y, int64_t rows, int c, int relu, float* mean, float* invstd,
                                     float* running_mean, float* running_var, void* stream) {
    SP_REQUIRE(z && sums && gamma && beta && y && mean && invstd, "sp_bn_apply_sums_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && total_rows > 0 && c > 0 && c % 4 == 0, "sp_bn_apply_sums_nhwc: bad shape");
    SP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sp_bn_apply_sums_nhwc: running stats come in pairs");
    const long long total = rows * (c / 4);
    if (bf16 & 1) hipLaunchKernelGGL(bn_apply_sums_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, sums, (double)total_rows, eps,
                                     momentum, gamma, beta, residual, y, c / 4, relu, total, mean, invstd, running_mean, running_var);
    else hipLaunchKernelGGL(bn_apply_sums_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, sums, (double)total_rows, eps,
                            momentum, gamma, beta, residual, y, c / 4, relu, total, mean, invstd, running_mean, running_var);
    return sp_check_launch("bn_apply_sums_kernel");
}

extern "C" int sp_bn_train_bwd_reduce_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                                           int64_t rows, int c, float* dgamma, float* dbeta, void* workspace, void* stream) {
    SP_REQUIRE(dy && z && mean && invstd && dgamma && dbeta && workspace, "sp_bn_train_bwd_reduce_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && c > 0 && c % 4 == 0 && rows < (1ll << 31), "sp_bn_train_bwd_reduce_nhwc: bad shape");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    const bool a16 = bf16 & 1, g16 = bf16 & 2;
    SP_REQUIRE(a16 || !g16, "sp_bn_train_bwd_reduce_nhwc: bf16 gradients with fp32 activations is not a supported mix");
    if (a16 && g16) hipLaunchKernelGGL((channel_reduce_kernel<1, true, true>), dim3(red_blocks(rows, c)), dim3(256), 0, s, dy, relu_src, z, mean, invstd, (int)rows, c, part);
    else if (a16) hipLaunchKernelGGL((channel_reduce_kernel<1, true, false>), dim3(red_blocks(rows, c)), dim3(256), 0, s, dy, relu_src, z, mean, invstd, (int)rows, c, part);
    else hipLaunchKernelGGL((channel_reduce_kernel<1, false, false>), dim3(red_blocks(rows, c)), dim3(256), 0, s, dy, relu_src, z, mean, invstd, (int)rows, c, part);
    hipLaunchKernelGGL(pair_sum_final_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, red_blocks(rows, c), c, dbeta, dgamma);
    return sp_check_launch("bn_train_bwd_reduce");
}

extern "C" int sp_bn_train_bwd_apply_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                                          const float* gamma, const float* sum_dgamma, const float* sum_dbeta, int64_t total_rows, int64_t rows,
                                          int c, void* dz, void* dres, int dres_accumulate, void* stream) {
    SP_REQUIRE(dy && z && mean && invstd && gamma && dz && sum_dgamma && sum_dbeta, "sp_bn_train_bwd_apply_nhwc: null pointer");
    SP_REQUIRE(rows > 0 && total_rows >= rows && c > 0 && c % 4 == 0 && rows < (1ll << 31), "sp_bn_train_bwd_apply_nhwc: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const bool a16 = bf16 & 1, g16 = bf16 & 2;
    SP_REQUIRE(a16 || !g16, "sp_bn_train_bwd_apply_nhwc: bf16 gradients with fp32 activations is not a supported mix");
    const int vw = (a16 && c % 8 == 0) ? 8 : 4;
    const int src_is_mask = (bf16 & 4) ? 1 : 0;
    SP_REQUIRE(!src_is_mask || (vw == 8 && relu_src), "sp_bn_train_bwd_apply_nhwc: a ReLU bit mask goes with bf16 activations (c %% 8 == 0)");
    const long long total = rows * (c / vw);
#define SP_BWD_APPLY(A, G, V)                                                                                                                \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<A, G, V>), dim3(grid_for(total, 256)), dim3(256), 0, s, dy, relu_src, z, mean, invstd, gamma, sum_dgamma, \
                       sum_dbeta, (float)(1.0 / (double)total_rows), dz, dres, dres_accumulate, c / V, total, src_is_mask)
    if (a16 && g16 && vw == 8) SP_BWD_APPLY(true, true, 8);
    else if (a16 && g16) SP_BWD_APPLY(true, true, 4);
    else if (a16 && vw == 8) SP_BWD_APPLY(true, false, 8);
    else if (a16) SP_BWD_APPLY(true, false, 4);
    else SP_BWD_APPLY(false, false, 4);
#undef SP_BWD_APPLY
    return sp_check_launch("bn_train_bwd_apply");
}

extern "C" int sp_bn_train_bwd_nhwc(const void* dy, int bf16, const void* relu_src, const void* z, const float* mean, const float* invstd,
                                    const float* gamma, int64_t rows, int c, void* dz, float* dgamma, float* dbeta, void* dres,
                                    int dres_accumulate, void* workspace, void* stream) {
    const int rc = sp_bn_train_bwd_reduce_nhwc(dy, bf16, relu_src, z, mean, invstd, rows, c, dgamma, dbeta, workspace, stream);
    if (rc) return rc;
    return sp_bn_train_bwd_apply_nhwc(dy, bf16, relu_src, z, mean, invstd, gamma, dgamma, dbeta, rows, rows, c, dz, dres, dres_accumulate, stream);
}

// sum over batch and pixels of an NCHW tensor, one 1024-thread workgroup per channel (the final layer's bias gradient straight from
// d loss / d heat maps: 17 channels of 32 x 3072 floats).  Only 17 workgroups exist, so the time is one workgroup's load latency chain:
// the (image, pixel-quad) space is walked with float4 loads, eight independent ones in flight per thread.  Thread t owns quads t,
// t + 1024, ... in fp64, then a fixed-order tree - deterministic.  VEC = 1 is the any-size fallback.
template <int VEC>
__global__ __launch_bounds__(1024) void channel_sum_nchw_kernel(const float* __restrict__ x, int batch, int channels, int hw, float* __restrict__ out) {
    __shared__ double sm[1024];
    const int c = blockIdx.x;
    const int per = hw / VEC;                              // vectors per image plane
    const int total = batch * per;
    double acc = 0;
    for (int i0 = threadIdx.x; i0 < total; i0 += 8 * 1024) {
        float v[8][VEC];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            const int b = i / per, q = i - b * per;
            const float* src = x + ((size_t)b * channels + c) * hw + (size_t)q * VEC;
            if (i < total) {
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                } else {
                    v[u][0] = src[0];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc += (double)v[u][e];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)sm[0];
}

extern "C" int sp_channel_sum_nchw(const float* x, int batch, int channels, int hw, float* sum, void* stream) {
    SP_REQUIRE(x && sum && batch > 0 && channels > 0 && hw > 0 && (int64_t)batch * hw < (1ll << 31), "sp_channel_sum_nchw: bad argument");
    if (hw % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        hipLaunchKernelGGL(channel_sum_nchw_kernel<4>, dim3(channels), dim3(1024), 0, (hipStream_t)stream, x, batch, channels, hw, sum);
    else
        hipLaunchKernelGGL(channel_sum_nchw_kernel<1>, dim3(channels), dim3(1024), 0, (hipStream_t)stream, x, batch, channels, hw, sum);
    return sp_check_launch("channel_sum_nchw_kernel");
}

extern "C" int sp_channel_sum_nhwc(const float* a, int64_t rows, int c, float* sum, void* workspace, void* stream) {
    SP_REQUIRE(a && sum && workspace && rows > 0 && c > 0 && c % 4 == 0 && rows < (1ll << 31), "sp_channel_sum_nhwc: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL((channel_reduce_kernel<0, false>), dim3(red_blocks(rows, c)), dim3(256), 0, s, a, nullptr, nullptr, nullptr, nullptr, (int)rows, c, part);
    hipLaunchKernelGGL(pair_sum_final_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, red_blocks(rows, c), c, sum, nullptr);
    return sp_check_launch("channel_sum");
}

extern "C" int sp_maxpool3x3s2_bwd_nhwc(const void* x, int bf16, const void* dy, void* dx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && dy && dx, "sp_maxpool3x3s2_bwd_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_maxpool3x3s2_bwd_nhwc: bad shape");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * h * w * (c / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_maxpool3x3s2_bwd_nhwc: tensor too large");
    const bool a16 = bf16 & 1, g16 = bf16 & 2;
#define SP_POOL_BWD(A, G) \
    hipLaunchKernelGGL((maxpool3x3s2_bwd_kernel<A, G>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, h, w, c / 4, ho, wo, total)
    if (a16 && g16) SP_POOL_BWD(true, true);
    else if (a16) SP_POOL_BWD(true, false);
    else SP_POOL_BWD(false, false);
#undef SP_POOL_BWD
    return sp_check_launch("maxpool3x3s2_bwd_kernel");
}

extern "C" int sp_maxpool3x3s2_idx_nhwc(const void* x, int bf16, void* y, void* idx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(x && y && idx, "sp_maxpool3x3s2_idx_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_maxpool3x3s2_idx_nhwc: bad shape");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * ho * wo * (c / 4);
    SP_REQUIRE((long long)batch * h * w * c < (1ll << 31), "sp_maxpool3x3s2_idx_nhwc: tensor too large");
    if (bf16 & 1) hipLaunchKernelGGL(maxpool3x3s2_idx_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                                 reinterpret_cast<unsigned int*>(idx), h, w, c / 4, ho, wo, total);
    else hipLaunchKernelGGL(maxpool3x3s2_idx_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                            reinterpret_cast<unsigned int*>(idx), h, w, c / 4, ho, wo, total);
    return sp_check_launch("maxpool3x3s2_idx_kernel");
}

extern "C" int sp_maxpool3x3s2_bwd_idx_nhwc(const void* idx, const void* dy, int bf16, void* dx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(idx && dy && dx, "sp_maxpool3x3s2_bwd_idx_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_maxpool3x3s2_bwd_idx_nhwc: bad shape");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * h * w * (c / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_maxpool3x3s2_bwd_idx_nhwc: tensor too large");
    if (bf16 & 2) hipLaunchKernelGGL(maxpool3x3s2_bwd_idx_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                 reinterpret_cast<const unsigned int*>(idx), dy, dx, h, w, c / 4, ho, wo, total);
    else hipLaunchKernelGGL(maxpool3x3s2_bwd_idx_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                            reinterpret_cast<const unsigned int*>(idx), dy, dx, h, w, c / 4, ho, wo, total);
    return sp_check_launch("maxpool3x3s2_bwd_idx_kernel");
}

extern "C" int sp_bn_apply_maxpool_nhwc(const void* z, int bf16, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y,
                                       void* idx, int batch, int h, int w, int c, void* stream) {
    SP_REQUIRE(z && mean && invstd && gamma && beta && y && idx, "sp_bn_apply_maxpool_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_bn_apply_maxpool_nhwc: bad shape");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)batch * ho * wo * (c / 4);
    SP_REQUIRE((long long)batch * h * w * c < (1ll << 31), "sp_bn_apply_maxpool_nhwc: tensor too large");
    if (bf16 & 1) hipLaunchKernelGGL(bn_apply_maxpool_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta, y,
                                     reinterpret_cast<unsigned int*>(idx), h, w, c / 4, ho, wo, total);
    else hipLaunchKernelGGL(bn_apply_maxpool_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, z, mean, invstd, gamma, beta, y,
                            reinterpret_cast<unsigned int*>(idx), h, w, c / 4, ho, wo, total);
    return sp_check_launch("bn_apply_maxpool_kernel");
}

extern "C" int sp_bn_maxpool_bwd_nhwc(const void* dy_pooled, int bf16, const void* idx, const void* z, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int batch, int h, int w, int c, float* dgamma, float* dbeta, void* dz,
                                      void* workspace, void* stream) {
    SP_REQUIRE(dy_pooled && idx && z && mean && invstd && gamma && beta && dgamma && dbeta && dz && workspace, "sp_bn_maxpool_bwd_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "sp_bn_maxpool_bwd_nhwc: bad shape");
    const bool a16 = bf16 & 1, g16 = bf16 & 2;
    SP_REQUIRE(a16 || !g16, "sp_bn_maxpool_bwd_nhwc: bf16 gradients with fp32 activations is not a supported mix");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long rows = (long long)batch * h * w, prows = (long long)batch * ho * wo, total = rows * (c / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_bn_maxpool_bwd_nhwc: tensor too large");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    // (512 partial blocks at most: with four rows in flight per thread two workgroups per CU cover the latency, and the fixed-order fold of the
    // partials - a launch on the chain - reads half the rows the generic reduction's 1,024 would give it)
    static const bool wide = !(getenv("SP_STEM_BWD_NARROW") && atoi(getenv("SP_STEM_BWD_NARROW")));   // (env: development knob - the round-4 geometry for same-box A/Bs)
    int nblk = red_blocks(prows, c);
    if (wide && nblk > 512) nblk = 512;
    const unsigned int* ix = reinterpret_cast<const unsigned int*>(idx);
#define SP_SPR(A, G) hipLaunchKernelGGL((stem_pool_bwd_reduce_kernel<A, G>), dim3(nblk), dim3(256), 0, s, dy_pooled, ix, z, mean, invstd, gamma, beta, h, w, ho, wo, (int)prows, c, part)
    if (a16 && g16) SP_SPR(true, true); else if (a16) SP_SPR(true, false); else SP_SPR(false, false);
#undef SP_SPR
    hipLaunchKernelGGL(pair_sum_final_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, nblk, c, dbeta, dgamma);
    const float inv_m = (float)(1.0 / (double)rows);
#define SP_SPA(A, G, V) hipLaunchKernelGGL((stem_pool_bwd_apply_kernel<A, G, V>), dim3(grid_for(total * 4 / V, 256)), dim3(256), 0, s, dy_pooled, ix, z, mean, invstd, gamma, beta, dgamma, dbeta, inv_m, dz, h, w, c / V, ho, wo, total * 4 / V)
    if (a16 && g16 && c % 8 == 0 && wide) SP_SPA(true, true, 8); else if (a16 && g16) SP_SPA(true, true, 4); else if (a16) SP_SPA(true, false, 4); else SP_SPA(false, false, 4);
#undef SP_SPA
    return sp_check_launch("sp_bn_maxpool_bwd_nhwc");
}

extern "C" int sp_nchw_to_nhwc_pad(const float* x, void* y, int y_bf16, int batch, int channels, int h, int w, int c_pad, void* stream) {
    SP_REQUIRE(x && y, "sp_nchw_to_nhwc_pad: null pointer");
    SP_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && c_pad >= channels && c_pad % 4 == 0, "sp_nchw_to_nhwc_pad: bad shape C=%d c_pad=%d",
               channels, c_pad);
    const long long total = (long long)batch * h * w * (c_pad / 4);
    SP_REQUIRE(total * 4 < (1ll << 31), "sp_nchw_to_nhwc_pad: tensor too large");
    if (y_bf16) hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, channels, h * w,
                               c_pad / 4, total);
    else hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, channels, h * w,
                            c_pad / 4, total);
    return sp_check_launch("nchw_to_nhwc_pad_kernel");
}

// Hyper-parameters arrive as doubles: torch forms 1 - beta, 1 - beta^step and lr / (1 - beta1^step) in Python doubles and only then hands
// them to its fp32 kernels; forming them from fp32 betas is a 1.3e-5 relative error in 1 - beta2 (0.999f = 0.99900001287).
static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, int step, float grad_scale) {
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    return AdamScalars{(float)(lr / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2), grad_scale};
}

extern "C" int sp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                            double beta2, double eps, int step, float grad_scale, void* stream) {
    SP_REQUIRE(param && grad && exp_avg && exp_avg_sq, "sp_adam_step: null pointer");
    SP_REQUIRE(n > 0 && n % 4 == 0 && step >= 1, "sp_adam_step: n=%lld must be a positive multiple of 4 and step >= 1", (long long)n);
    hipLaunchKernelGGL(adam_kernel<false>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4*>(param),
                       reinterpret_cast<const f32x4*>(grad), reinterpret_cast<f32x4*>(exp_avg), reinterpret_cast<f32x4*>(exp_avg_sq), n / 4,
                       adam_scalars(lr, beta1, beta2, eps, step, grad_scale), nullptr);
    return sp_check_launch("adam_kernel");
}

// The same update with the step's scalars in device memory: sp_adam_set_scalars (one tiny launch per step, the values formed on the host
// exactly as sp_adam_step forms them) + any number of sp_adam_step_dev launches whose arguments never change - a captured step replays
extern "C" int sp_adam_set_scalars(double lr, double beta1, double beta2, double eps, int step, float grad_scale, float* scalars8, void* stream) {
    SP_REQUIRE(scalars8 && step >= 1, "sp_adam_set_scalars: null pointer or step < 1");
    hipLaunchKernelGGL(adam_set_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, adam_scalars(lr, beta1, beta2, eps, step, grad_scale),
                       reinterpret_cast<AdamScalars*>(scalars8));
    return sp_check_launch("adam_set_scalars_kernel");
}

extern "C" int sp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* scalars8, void* stream) {
    SP_REQUIRE(param && grad && exp_avg && exp_avg_sq && scalars8, "sp_adam_step_dev: null pointer");
    SP_REQUIRE(n > 0 && n % 4 == 0, "sp_adam_step_dev: n=%lld must be a positive multiple of 4", (long long)n);
    hipLaunchKernelGGL(adam_kernel<true>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4*>(param),
                       reinterpret_cast<const f32x4*>(grad), reinterpret_cast<f32x4*>(exp_avg), reinterpret_cast<f32x4*>(exp_avg_sq), n / 4,
                       AdamScalars{}, reinterpret_cast<const AdamScalars*>(scalars8));
    return sp_check_launch("adam_kernel");
}

// ---- SELayer backward (nets/commons.py:4-18 inside Bottleneck.forward, pose_resnet_dconv.py:124-131):
//   forward  u = bn3(conv3(t)); s = mean_hw(u); h = relu(fc0 s + b0); g = fc2 h + b2; y = relu(u * sigmoid(g) + identity)
//   backward dr = dy (y > 0);  d identity += dr;  da[b,c] = sum_hw dr u;  dg = da a (1 - a);  (fc2, relu, fc0 through the conv kernels)
//            du = dr a + ds / HW
// The two FC layers are 1x1 convolutions on a [B,1,1,C] map and go through the conv kernels (forward, dgrad, wgrad); the four small
// kernels below are what is specific to the gate.  Not on any BASELINE config's path: written for clarity, deterministic (fixed-order
// sums in fp64), coalesced along the channels.
namespace {

template <bool BF16> __device__ __forceinline__ float ld_act(const void* p, size_t i) {
    if constexpr (BF16) return (float)reinterpret_cast<const __bf16*>(p)[i];
    else return reinterpret_cast<const float*>(p)[i];
}
template <bool BF16> __device__ __forceinline__ void st_act(void* p, size_t i, float v) {
    if constexpr (BF16) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v;
    else reinterpret_cast<float*>(p)[i] = v;
}

// da[b][c] = sum over pixels of (y > 0 ? dy : 0) * u; workgroup = (64 channels, image b), 4 row lanes per channel
template <bool BF16>
__global__ __launch_bounds__(256) void se_gate_bwd_reduce_kernel(const float* __restrict__ dy, const void* __restrict__ y, const void* __restrict__ u,
                                                                 int HW, int C, float* __restrict__ da) {
    __shared__ double sm[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.y;
    double acc = 0;
    if (c < C)
        for (int r = rl; r < HW; r += 4) {
            const size_t i = ((size_t)b * HW + r) * C + c;
            const float dr = ld_act<BF16>(y, i) > 0.f ? dy[i] : 0.f;
            acc += (double)dr * (double)ld_act<BF16>(u, i);
        }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0 && c < C) da[(size_t)b * C + c] = (float)(((sm[threadIdx.x] + sm[threadIdx.x + 64]) + sm[threadIdx.x + 128]) + sm[threadIdx.x + 192]);
}

// MODE 0: out = da * a (1 - a) with a = sigmoid(gate logit);  MODE 1: out = dh (h > 0).  bias gradient db[c] = sum_b out[b][c] either way
template <bool BF16, int MODE>
__global__ void se_rows_bwd_kernel(const float* __restrict__ grad, const void* __restrict__ saved, int B, int C, void* __restrict__ out, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0;
    for (int b = 0; b < B; ++b) {
        const size_t i = (size_t)b * C + c;
        const float x = ld_act<BF16>(saved, i);
        float v;
        if constexpr (MODE == 0) {
            const float a = 1.f / (1.f + expf(-x));
            v = grad[i] * a * (1.f - a);
        } else {
            v = x > 0.f ? grad[i] : 0.f;
        }
        st_act<BF16>(out, i, v);
        s += (double)v;
    }
    if (db) db[c] = (float)s;
}

template <bool BF16>
__global__ void se_gate_bwd_apply_kernel(const float* __restrict__ dy, const void* __restrict__ y, const void* __restrict__ g, const float* __restrict__ ds,
                                         int HW, int C, float inv_hw, float* __restrict__ du, float* __restrict__ dres, int accumulate, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long b = i / ((long long)HW * C);
        const float dr = ld_act<BF16>(y, (size_t)i) > 0.f ? dy[i] : 0.f;
        const float a = 1.f / (1.f + expf(-ld_act<BF16>(g, (size_t)(b * C + c))));
        du[i] = dr * a + ds[b * C + c] * inv_hw;
        dres[i] = accumulate ? dres[i] + dr : dr;
    }
}

}  // namespace

extern "C" int sp_se_gate_bwd_reduce(const float* dy, int bf16, const void* y, const void* u, int batch, int hw, int c, float* da, void* stream) {
    SP_REQUIRE(dy && y && u && da && batch > 0 && hw > 0 && c > 0, "sp_se_gate_bwd_reduce: bad argument");
    const dim3 grid((c + 63) / 64, batch);
    if (bf16) hipLaunchKernelGGL(se_gate_bwd_reduce_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, u, hw, c, da);
    else hipLaunchKernelGGL(se_gate_bwd_reduce_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, u, hw, c, da);
    return sp_check_launch("se_gate_bwd_reduce_kernel");
}

extern "C" int sp_se_sigmoid_bwd(const float* da, int bf16, const void* gate_logits, int batch, int c, void* dg, float* dbias, void* stream) {
    SP_REQUIRE(da && gate_logits && dg && batch > 0 && c > 0, "sp_se_sigmoid_bwd: bad argument");
    const dim3 grid((c + 255) / 256);
    if (bf16) hipLaunchKernelGGL((se_rows_bwd_kernel<true, 0>), grid, dim3(256), 0, (hipStream_t)stream, da, gate_logits, batch, c, dg, dbias);
    else hipLaunchKernelGGL((se_rows_bwd_kernel<false, 0>), grid, dim3(256), 0, (hipStream_t)stream, da, gate_logits, batch, c, dg, dbias);
    return sp_check_launch("se_rows_bwd_kernel");
}

extern "C" int sp_relu_bwd_rows(const float* dh, int bf16, const void* h, int batch, int c, void* out, float* dbias, void* stream) {
    SP_REQUIRE(dh && h && out && batch > 0 && c > 0, "sp_relu_bwd_rows: bad argument");
    const dim3 grid((c + 255) / 256);
    if (bf16) hipLaunchKernelGGL((se_rows_bwd_kernel<true, 1>), grid, dim3(256), 0, (hipStream_t)stream, dh, h, batch, c, out, dbias);
    else hipLaunchKernelGGL((se_rows_bwd_kernel<false, 1>), grid, dim3(256), 0, (hipStream_t)stream, dh, h, batch, c, out, dbias);
    return sp_check_launch("se_rows_bwd_kernel");
}

extern "C" int sp_se_gate_bwd_apply(const float* dy, int bf16, const void* y, const void* gate_logits, const float* ds, int batch, int hw, int c,
                                    float* du, float* dres, int dres_accumulate, void* stream) {
    SP_REQUIRE(dy && y && gate_logits && ds && du && dres && batch > 0 && hw > 0 && c > 0, "sp_se_gate_bwd_apply: bad argument");
    const long long total = (long long)batch * hw * c;
    SP_REQUIRE(total < (1ll << 31), "sp_se_gate_bwd_apply: tensor too large");
    if (bf16) hipLaunchKernelGGL(se_gate_bwd_apply_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, gate_logits, ds, hw,
                                 c, 1.f / (float)hw, du, dres, dres_accumulate, total);
    else hipLaunchKernelGGL(se_gate_bwd_apply_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, gate_logits, ds, hw, c,
                            1.f / (float)hw, du, dres, dres_accumulate, total);
    return sp_check_launch("se_gate_bwd_apply_kernel");
}

// Backward of y = [relu](base + nearest_upsample(x, f)) (HRNet fuse layers, nets/pose_hrnet.py:192-202,250-257; f = 1: a plain add):
// dr = relu ? dy (y > 0) : dy;  d base (+)= dr;  d x (+)= sum of dr over the f x f block above each low-resolution pixel.  One thread per
// (low-resolution pixel, 4 channels); fixed summation order.
template <bool BF16>
__global__ void upsample_add_bwd_kernel(const f32x4* __restrict__ dy, const void* __restrict__ y, int h, int w, int C4, int f, f32x4* __restrict__ dbase,
                                        int base_acc, f32x4* __restrict__ dx, int x_acc, long long total) {
    const int W = w * f, H = h * f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int X0 = (int)(r % w) * f; r /= w;
        const int Y0 = (int)(r % h) * f;
        const long long b = r / h;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int yy = 0; yy < f; ++yy)
            for (int xx = 0; xx < f; ++xx) {
                const long long I = ((b * H + Y0 + yy) * W + X0 + xx) * C4 + c;
                f32x4 g = dy[I];
                if (y) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float yv = BF16 ? (float)reinterpret_cast<const __bf16*>(y)[I * 4 + e] : reinterpret_cast<const float*>(y)[I * 4 + e];
                        g[e] = yv > 0.f ? g[e] : 0.f;
                    }
                }
                if (base_acc) { const f32x4 o = dbase[I]; dbase[I] = f32x4{o[0] + g[0], o[1] + g[1], o[2] + g[2], o[3] + g[3]}; }
                else dbase[I] = g;
                acc[0] += g[0]; acc[1] += g[1]; acc[2] += g[2]; acc[3] += g[3];
            }
        if (x_acc) { const f32x4 o = dx[i]; dx[i] = f32x4{o[0] + acc[0], o[1] + acc[1], o[2] + acc[2], o[3] + acc[3]}; }
        else dx[i] = acc;
    }
}

extern "C" int sp_upsample_add_bwd_nhwc(const float* dy, int bf16, const void* y_relu_src, int batch, int h, int w, int c, int factor, float* dbase,
                                        int dbase_accumulate, float* dx, int dx_accumulate, void* stream) {
    SP_REQUIRE(dy && dbase && dx, "sp_upsample_add_bwd_nhwc: null pointer");
    SP_REQUIRE(batch > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && factor >= 1, "sp_upsample_add_bwd_nhwc: bad shape");
    SP_REQUIRE(dbase != dx, "sp_upsample_add_bwd_nhwc: the two gradients must be different tensors");
    const long long total = (long long)batch * h * w * (c / 4);
    SP_REQUIRE(total * factor * factor * 4 < (1ll << 31), "sp_upsample_add_bwd_nhwc: tensor too large");
    if (bf16) hipLaunchKernelGGL(upsample_add_bwd_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                 reinterpret_cast<const f32x4*>(dy), y_relu_src, h, w, c / 4, factor, reinterpret_cast<f32x4*>(dbase), dbase_accumulate,
                                 reinterpret_cast<f32x4*>(dx), dx_accumulate, total);
    else hipLaunchKernelGGL(upsample_add_bwd_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                            reinterpret_cast<const f32x4*>(dy), y_relu_src, h, w, c / 4, factor, reinterpret_cast<f32x4*>(dbase), dbase_accumulate,
                            reinterpret_cast<f32x4*>(dx), dx_accumulate, total);
    return sp_check_launch("upsample_add_bwd_kernel");
}

// Measurement aid (bench.py --sync-bn-latency-us): keep `stream` busy for `us` microseconds of the 100 MHz constant clock - a stand-in
// for the latency of a small cross-GPU message on a box with one GPU.  One wave, no memory traffic.
__global__ void stream_delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int sp_stream_delay_us(double us, void* stream) {
    SP_REQUIRE(us >= 0 && us <= 1e6, "sp_stream_delay_us: 0 .. 1e6 us");
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(us * 100.0));
    return sp_check_launch("stream_delay_kernel");
}

extern "C" int sp_permute4_f32(const float* src, void* dst, int dst_bf16, const int32_t* dst_dims, const int64_t* src_strides, const int32_t* valid,
                               int64_t src_base, int64_t dst_offset, void* stream) {
    SP_REQUIRE(src && dst && dst_dims && src_strides && valid, "sp_permute4_f32: null pointer");
    Permute4 pm;
    long long total = 1;
    for (int i = 0; i < 4; ++i) {
        SP_REQUIRE(dst_dims[i] > 0, "sp_permute4_f32: bad extent");
        pm.d[i] = dst_dims[i]; pm.s[i] = src_strides[i]; pm.lim[i] = valid[i];
        total *= dst_dims[i];
    }
    pm.base = src_base; pm.dst_off = dst_offset;
    if (dst_bf16) hipLaunchKernelGGL(permute4_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, pm, total);
    else hipLaunchKernelGGL(permute4_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, pm, total);
    return sp_check_launch("permute4_kernel");
}
