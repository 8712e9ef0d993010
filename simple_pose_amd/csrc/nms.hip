// nms.hip - per-image pose rescoring + greedy OKS-NMS after decode, and the result-score rule.
// Replaces eval.py:153-197 (temp_read_in_and_filter), datasets/naive_data.py:120-173 (oks_iou, oks_nms) and the score rule of
// metrics/pose_metrics.py:172-179 (kps_to_dict_).  The reference runs these in numpy float64 on the host after a JSON round
// trip; here the decoded key points never leave the device.  One workgroup per image; all arithmetic in fp64, operation by
// operation as numpy evaluates it (contraction OFF; numpy's pairwise add.reduce order for the 17-term sums).
#include "sp_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int NMS_MAX_JOINTS = 64;
constexpr int NMS_MAX_GROUP = 2048;

struct NmsVar { double v[NMS_MAX_JOINTS]; };

// numpy float64 add.reduce over a contiguous run: 8 interleaved accumulators over the multiple-of-8 prefix, combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail in order; n < 8: in order.
__device__ double np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// eval.py:166-174: score = box_score * mean(kpt_scores[kpt_scores > in_vis_thre]) (0 without a visible joint); also widens the
// fp32 decoder output to the float64 the reference's JSON round trip produces.
__global__ void pose_rescore_kernel(const float* __restrict__ kps, const double* __restrict__ box_score, int P, int J, double vis,
                                    double* __restrict__ kps64, double* __restrict__ score) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double buf[NMS_MAX_JOINTS];
    int k = 0;
    for (int j = 0; j < J; ++j) {
        const float* s = kps + ((size_t)p * J + j) * 3;
        const double x = (double)s[0], y = (double)s[1], c = (double)s[2];
        if (kps64) {
            double* d = kps64 + ((size_t)p * J + j) * 3;
            d[0] = x; d[1] = y; d[2] = c;
        }
        if (c > vis) buf[k++] = c;
    }
    const double m = k > 0 ? np_pairwise_sum(buf, k) / (double)k : 0.0;
    score[p] = box_score[p] * m;
}

// oks_iou (naive_data.py:120-150) of the pick against one candidate
__device__ double oks_one(const double* pick /* LDS [J][3] */, const double* __restrict__ cand, double pick_area, double cand_area,
                          const NmsVar& var, int J, double vis_thresh) {
    double term[NMS_MAX_JOINTS];
    float vis_sum = 0.f;
    const double denom = (pick_area + cand_area) / 2 + 1e-12;
    for (int j = 0; j < J; ++j) {
        const double dx = cand[j * 3] - pick[j * 3], dy = cand[j * 3 + 1] - pick[j * 3 + 1];
        const double e = (dx * dx + dy * dy) / var.v[j] / denom / 2;
        float vis = 1.f;
        if (vis_thresh >= 0) vis = (cand[j * 3 + 2] > vis_thresh && pick[j * 3 + 2] > vis_thresh) ? 1.f : 0.f;
        term[j] = exp(-e) * (double)vis;
        vis_sum += vis;
    }
    const float den = vis_sum + (float)1e-12;            // float32 + weak python float stays float32
    return np_pairwise_sum(term, J) / (double)den;
}

// oks_nms (naive_data.py:153-173), one workgroup per image: rank sort (descending score, ties: higher index first), then the
// greedy loop - picks are sequential, the candidates of one pick are tested in parallel.
__global__ __launch_bounds__(256) void oks_nms_kernel(const double* __restrict__ kps, const double* __restrict__ scores,
                                                      const double* __restrict__ areas, const int* __restrict__ seg, int J,
                                                      const NmsVar var, double thresh, double vis_thresh, int* __restrict__ keep,
                                                      int* __restrict__ keep_count) {
    __shared__ int order[NMS_MAX_GROUP];
    __shared__ unsigned char alive[NMS_MAX_GROUP];
    __shared__ double pick[NMS_MAX_JOINTS * 3];
    __shared__ int n_keep;
    const int g = blockIdx.x, lo = seg[g], N = seg[g + 1] - lo, tid = threadIdx.x;
    if (N <= 0 || N > NMS_MAX_GROUP) {
        if (tid == 0) keep_count[g] = N <= 0 ? 0 : -1;
        return;
    }
    const double* sc = scores + lo;
    for (int i = tid; i < N; i += 256) {
        const double si = sc[i];
        const bool ni = si != si;
        int rank = 0;
        for (int j = 0; j < N; ++j) {                      // a total order even with NaN scores (NaN first, as argsort()[::-1])
            const double sj = sc[j];
            const bool nj = sj != sj;
            const bool before = (nj || ni) ? (nj && (!ni || j > i)) : (sj > si || (sj == si && j > i));
            rank += before ? 1 : 0;
        }
        order[rank] = i;
        alive[i] = 1;
        keep[lo + i] = -1;
    }
    if (tid == 0) n_keep = 0;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const int p = order[k];
        if (!alive[p]) continue;                           // uniform: LDS state is identical for every thread after the barrier
        for (int i = tid; i < J * 3; i += 256) pick[i] = kps[(size_t)(lo + p) * J * 3 + i];
        if (tid == 0) { keep[lo + n_keep] = lo + p; ++n_keep; }
        __syncthreads();
        const double pa = areas[lo + p];
        for (int q = k + 1 + tid; q < N; q += 256) {
            const int c = order[q];
            if (!alive[c]) continue;
            const double o = oks_one(pick, kps + (size_t)(lo + c) * J * 3, pa, areas[lo + c], var, J, vis_thresh);
            if (!(o <= thresh)) alive[c] = 0;              // order = order[oks_ovr <= thresh]
        }
        __syncthreads();
    }
    if (tid == 0) keep_count[g] = n_keep;
}

// kps_to_dict_: score = sc.mean() + sc.max() over the J per-joint maxima (fp32)
__global__ void pose_score_kernel(const float* __restrict__ max_val, int B, int J, float* __restrict__ score) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    float m = max_val[(size_t)b * J];
    for (int j = 0; j < J; ++j) {
        const float v = max_val[(size_t)b * J + j];
        s += (double)v;
        if (v > m) m = v;
    }
    score[b] = (float)s / (float)J + m;
}

}  // namespace

extern "C" int sp_pose_rescore(const float* kps, const double* box_score, int persons, int joints, double in_vis_thre, double* kps64,
                               double* score, void* stream) {
    SP_REQUIRE(kps && box_score && score, "sp_pose_rescore: null pointer");
    SP_REQUIRE(persons > 0 && joints > 0 && joints <= NMS_MAX_JOINTS, "sp_pose_rescore: persons=%d joints=%d (joints <= %d)", persons, joints,
               NMS_MAX_JOINTS);
    hipLaunchKernelGGL(pose_rescore_kernel, dim3(sp_ceil_div(persons, 64)), dim3(64), 0, (hipStream_t)stream, kps, box_score, persons, joints,
                       in_vis_thre, kps64, score);
    return sp_check_launch("pose_rescore_kernel");
}

extern "C" int sp_oks_nms(const double* kps, const double* scores, const double* areas, const int32_t* seg, int groups, int max_group,
                          int joints, const double* sigmas_host, double thresh, double vis_thresh, int32_t* keep, int32_t* keep_count,
                          void* stream) {
    static const double coco[17] = {.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89};
    SP_REQUIRE(kps && scores && areas && seg && keep && keep_count, "sp_oks_nms: null pointer");
    SP_REQUIRE(groups > 0 && joints > 0 && joints <= NMS_MAX_JOINTS, "sp_oks_nms: groups=%d joints=%d", groups, joints);
    SP_REQUIRE(sigmas_host || joints == 17, "sp_oks_nms: the default sigmas are COCO's 17; pass sigmas for %d joints", joints);
    SP_REQUIRE(max_group >= 0 && max_group <= NMS_MAX_GROUP, "sp_oks_nms: %d persons in one image (limit %d)", max_group, NMS_MAX_GROUP);
    NmsVar var;
    for (int j = 0; j < joints; ++j) {
        const double s = sigmas_host ? sigmas_host[j] : coco[j] / 10.0;   // naive_data.py:131-133
        var.v[j] = (s * 2) * (s * 2);
    }
    hipLaunchKernelGGL(oks_nms_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, kps, scores, areas, seg, joints, var, thresh, vis_thresh,
                       keep, keep_count);
    return sp_check_launch("oks_nms_kernel");
}

extern "C" int sp_pose_score(const float* max_val, int batch, int joints, float* score, void* stream) {
    SP_REQUIRE(max_val && score && batch > 0 && joints > 0, "sp_pose_score: bad argument");
    hipLaunchKernelGGL(pose_score_kernel, dim3(sp_ceil_div(batch, 64)), dim3(64), 0, (hipStream_t)stream, max_val, batch, joints, score);
    return sp_check_launch("pose_score_kernel");
}
