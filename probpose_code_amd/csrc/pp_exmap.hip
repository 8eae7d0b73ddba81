// Ex-mAP after the similarities: the greedy detection <-> instance matching of every (image, visibility level, area range)
// cell at every similarity threshold (COCOeval.evaluateImg, mmpose/evaluation/metrics/_cocoeval.py:709-887, the
// return_matching=False branch) and the precision / recall / score tables over the whole dataset (COCOeval.accumulate,
// :889-1009). Integer / index work: results are bit-identical to the numpy original (the precision values are quotients
// of exactly representable counts, computed in float64 like numpy does).
#include "pp_common.h"

namespace pp {

struct MatchParams {
    const double* ious;            // per cell a (L, Dc, Gc) block at cell_iou_off[c]; detections in evaluation order
    const int* cell_gt_off;        // (n_cells + 1)
    const int* cell_dt_off;        // (n_cells + 1)
    const long long* cell_iou_off; // (n_cells)
    const unsigned char* gt_ignore;  // (N_gt, L)  gt["ignore"][level]
    const unsigned char* gt_iscrowd; // (N_gt)
    const double* gt_area;         // (N_gt) the area the range test uses (:729-733)
    const double* gt_bbox;         // (N_gt, 4) xywh - match_by_bbox only
    const double* dt_area;         // (N_dt)
    const double* dt_bbox;         // (N_dt, 4) xywh - match_by_bbox only
    const double* area_rng;        // (A, 2)
    const double* iou_thrs;        // (T)
    int* dt_match;                 // (L, A, T, N_dt) global instance index or -1
    unsigned char* dt_ignore;      // (L, A, T, N_dt)
    int* gt_match;                 // (L, A, T, N_gt) global detection index or -1
    unsigned char* gt_ignore_out;  // (L, A, N_gt)  the cell's _ignore flag of each instance
    double* sim_sum;               // (L, A, n_cells) sum of the similarities of the matches made (all thresholds)
    int* sim_cnt;                  // (L, A, n_cells)
    int n_cells, N_gt, N_dt, L, A, T, match_by_bbox;
};

// One workgroup per (cell, level x area range); thread t < T walks the detections of the cell in score order for threshold t.
// LDS: order[G] (instances, not-ignored first, stable), flag[G] in that order, taken[T][G].
__global__ void exoks_match_kernel(const MatchParams p) {
    extern __shared__ int lds[];
    const int c = blockIdx.x, lvl = blockIdx.y / p.A, a = blockIdx.y % p.A;
    const int g0 = p.cell_gt_off[c], G = p.cell_gt_off[c + 1] - g0, d0 = p.cell_dt_off[c], D = p.cell_dt_off[c + 1] - d0;
    int* order = lds;
    int* flag = lds + G;
    int* taken = lds + 2 * G;  // [T][G] detection index + 1, 0 = free
    __shared__ int n_keep;
    __shared__ double s_sum[64];
    __shared__ int s_cnt[64];
    const double lo = p.area_rng[2 * a], hi = p.area_rng[2 * a + 1];
    const size_t row = (size_t)lvl * p.A + a;
    if (threadIdx.x == 0) {  // stable partition: np.argsort(_ignore, kind="mergesort") (:738)
        int n = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int g = 0; g < G; ++g) {
                const double ar = p.gt_area[g0 + g];
                const int ig = (p.gt_ignore[(size_t)(g0 + g) * p.L + lvl] || ar < lo || ar > hi) ? 1 : 0;
                if (pass == 0) p.gt_ignore_out[row * p.N_gt + g0 + g] = (unsigned char)ig;
                if (ig == pass) {
                    order[n] = g;
                    flag[n] = ig;
                    ++n;
                }
            }
        int keep = 0;
        for (int g = 0; g < G; ++g) keep += flag[g] == 0;
        n_keep = keep;
    }
    for (int i = threadIdx.x; i < p.T * G; i += blockDim.x) taken[i] = 0;
    __syncthreads();
    const int t = threadIdx.x;
    double sum = 0.0;
    int cnt = 0;
    if (t < p.T) {
        const double thr = p.iou_thrs[t];
        const double* iou = p.ious + p.cell_iou_off[c] + (size_t)lvl * D * G;
        int* mine = taken + t * G;
        const size_t out = (row * p.T + t);
        const bool all_ignored = n_keep == 0;  // np.all(gtIg), true for an image without instances as well (:866)
        for (int d = 0; d < D; ++d) {
            int m = -1;
            double best = fmin(thr, 1 - 1e-10);
            if (G > 0) {
                double cxd = 0, cyd = 0, nearest = 20;
                if (p.match_by_bbox) {
                    const double* b = p.dt_bbox + 4 * (size_t)(d0 + d);
                    cxd = b[0] + b[2] / 2;
                    cyd = b[1] + b[3] / 2;
                }
                for (int gi = 0; gi < G; ++gi) {
                    const int g = order[gi];
                    if (mine[gi] && !p.gt_iscrowd[g0 + g]) continue;  // already matched, and not a crowd
                    if (m > -1 && flag[m] == 0 && flag[gi] == 1) break;  // matched a regular instance: the rest are ignored ones
                    const double v = iou[(size_t)d * G + g];
                    if (p.match_by_bbox) {
                        if (v < thr) continue;
                        const double* b = p.gt_bbox + 4 * (size_t)(g0 + g);
                        const double dist = fabs(cxd - (b[0] + b[2] / 2)) + fabs(cyd - (b[1] + b[3] / 2));
                        if (dist < nearest) {
                            nearest = dist;
                            m = gi;
                            best = v;
                        }
                    } else {
                        if (v < best) continue;
                        best = v;
                        m = gi;
                    }
                }
            }
            int ig = 0;
            if (m >= 0) {
                mine[m] = d + 1;
                ig = flag[m];
                sum += best;
                ++cnt;
            } else {
                const double ar = p.dt_area[d0 + d];
                ig = (ar < lo || ar > hi) ? 1 : 0;  // unmatched detections outside the area range (:862-864)
            }
            if (all_ignored) ig = 1;
            p.dt_match[out * p.N_dt + d0 + d] = m >= 0 ? g0 + order[m] : -1;
            p.dt_ignore[out * p.N_dt + d0 + d] = (unsigned char)ig;
        }
        for (int gi = 0; gi < G; ++gi) p.gt_match[out * p.N_gt + g0 + order[gi]] = mine[gi] ? d0 + mine[gi] - 1 : -1;
    }
    s_sum[threadIdx.x] = sum;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        int n = 0;
        for (int i = 0; i < p.T; ++i) {
            s += s_sum[i];
            n += s_cnt[i];
        }
        p.sim_sum[row * p.n_cells + c] = s;
        p.sim_cnt[row * p.n_cells + c] = n;
    }
}

struct AccParams {
    const int* dt_match;            // (L, A, T, N_dt)
    const unsigned char* dt_ignore; // (L, A, T, N_dt)
    const unsigned char* gt_ignore; // (L, A, N_gt)
    const int* order;               // (N_dt) detections of the dataset by descending score, stable
    const double* dt_score;         // (N_dt)
    const double* rec_thrs;         // (R)
    double* precision;              // (T, L, R, A)   [T x V x R x K=1 x A x M=1]
    double* recall;                 // (T, L, A)
    double* scores;                 // (T, L, R, A)
    int* chunk_base;                // (L * A * T, n_chunks, 2) scratch: true / false positives before each chunk
    int N_gt, N_dt, L, A, T, R, n_chunks, n_cells;
};

constexpr int ACC_THREADS = 256;

// inclusive scan of one int per thread over the workgroup (4 waves)
__device__ int block_scan_incl(int v, int* wave_tot) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wv] = v;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wave_tot[w];
    return v + base;
}

// One workgroup per (level, area range, threshold). Forward sweep: running true / false positive counts, the recall, and
// for every recall threshold the first position that reaches it (np.searchsorted(rc, recThrs, "left"), :989). Backward
// sweep: the precision made non-increasing from the right (:984-986) picked up at those positions.
__global__ void exmap_accumulate_kernel(const AccParams p) {
    __shared__ int wave_tot[4];
    __shared__ int need[128];   // smallest true-positive count whose recall reaches threshold r
    __shared__ int first[128];  // position where that count is reached (N_dt = never)
    __shared__ double env[ACC_THREADS];
    __shared__ double wave_max[4];
    __shared__ int s_npig;
    const int row = blockIdx.x;  // (lvl * A + a) * T + t
    const int t = row % p.T, la = row / p.T, a = la % p.A, lvl = la / p.A;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double eps = 2.220446049250313e-16;  // np.spacing(1)

    // instances that count (:954)
    int local = 0;
    for (int g = tid; g < p.N_gt; g += ACC_THREADS) local += p.gt_ignore[(size_t)la * p.N_gt + g] == 0;
    const int npig_incl = block_scan_incl(local, wave_tot);
    if (tid == ACC_THREADS - 1) s_npig = npig_incl;
    __syncthreads();
    const int npig = s_npig;
    if (p.n_cells == 0 || npig == 0) return;  // the tables keep their -1 (:941-943, :955-956)
    const double dn = (double)npig;
    for (int r = tid; r < p.R; r += ACC_THREADS) {
        const double thr = p.rec_thrs[r];
        long long n = (long long)ceil(thr * dn);
        if (n < 0) n = 0;
        while (n > 0 && (double)(n - 1) / dn >= thr) --n;
        while ((double)n / dn < thr) ++n;  // (past npig: never reached)
        need[r] = n > 0x7fffffff ? 0x7fffffff : (int)n;
        first[r] = need[r] == 0 && p.N_dt > 0 ? 0 : p.N_dt;
    }
    __syncthreads();
    const int* match = p.dt_match + (size_t)row * p.N_dt;
    const unsigned char* ign = p.dt_ignore + (size_t)row * p.N_dt;
    int* base = p.chunk_base + (size_t)row * p.n_chunks * 2;
    int tp_run = 0, fp_run = 0;
    for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int i = ch * ACC_THREADS + tid;
        int tp = 0, fp = 0;
        if (i < p.N_dt) {
            const int j = p.order[i];
            const bool m = match[j] >= 0, ig = ign[j] != 0;
            tp = m && !ig;
            fp = !m && !ig;
        }
        if (tid == 0) {
            base[2 * ch] = tp_run;
            base[2 * ch + 1] = fp_run;
        }
        const int packed = block_scan_incl(tp | (fp << 16), wave_tot);  // ACC_THREADS < 65536: the halves cannot carry
        const int tp_c = tp_run + (packed & 0xffff);
        if (tp) {  // this is the tp_c-th true positive: the answer for every threshold that needs exactly tp_c
            int lo = 0, hi = p.R;  // need[] is non-decreasing: first r with need[r] >= tp_c
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (need[mid] < tp_c) lo = mid + 1;
                else hi = mid;
            }
            for (int r = lo; r < p.R && need[r] == tp_c; ++r) first[r] = i;
        }
        __syncthreads();
        if (tid == ACC_THREADS - 1) wave_tot[0] = packed;  // (block_scan_incl is done with wave_tot)
        __syncthreads();
        const int tot = wave_tot[0];
        __syncthreads();
        tp_run += tot & 0xffff;
        fp_run += tot >> 16;
    }
    if (tid == 0) p.recall[((size_t)t * p.L + lvl) * p.A + a] = p.N_dt ? (double)tp_run / dn : 0.0;
    for (int r = tid; r < p.R; r += ACC_THREADS) {  // thresholds never reached stay 0 (:990-995)
        const size_t o = (((size_t)t * p.L + lvl) * p.R + r) * p.A + a;
        p.precision[o] = 0.0;
        p.scores[o] = first[r] < p.N_dt ? p.dt_score[p.order[first[r]]] : 0.0;
    }
    __syncthreads();
    double carry = -1.0;  // max of the precision over all later chunks (precision >= 0)
    for (int ch = p.n_chunks - 1; ch >= 0; --ch) {
        const int i = ch * ACC_THREADS + tid;
        int tp = 0, fp = 0;
        if (i < p.N_dt) {
            const int j = p.order[i];
            const bool m = match[j] >= 0, ig = ign[j] != 0;
            tp = m && !ig;
            fp = !m && !ig;
        }
        const int packed = block_scan_incl(tp | (fp << 16), wave_tot);
        const double tps = (double)(base[2 * ch] + (packed & 0xffff)), fps = (double)(base[2 * ch + 1] + (packed >> 16));
        double v = i < p.N_dt ? tps / (fps + tps + eps) : -1.0;
        // suffix max inside the chunk
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_down(v, o, 64);
            if (lane + o < 64) v = fmax(v, u);
        }
        __syncthreads();
        if (lane == 0) wave_max[wv] = v;
        __syncthreads();
        for (int w = wv + 1; w < 4; ++w) v = fmax(v, wave_max[w]);
        v = fmax(v, carry);
        env[tid] = v;
        __syncthreads();
        for (int r = tid; r < p.R; r += ACC_THREADS) {
            const int f = first[r];
            if (f < p.N_dt && f / ACC_THREADS == ch) p.precision[(((size_t)t * p.L + lvl) * p.R + r) * p.A + a] = env[f % ACC_THREADS];
        }
        carry = env[0];
        __syncthreads();
    }
}

}  // namespace pp

extern "C" int pp_exoks_match(const double* ious, const int* cell_gt_off, const int* cell_dt_off, const long long* cell_iou_off,
                              const unsigned char* gt_ignore, const unsigned char* gt_iscrowd, const double* gt_area,
                              const double* gt_bbox, const double* dt_area, const double* dt_bbox, const double* area_rng,
                              const double* iou_thrs, int n_cells, int max_gt_per_cell, int N_gt, int N_dt, int L, int A, int T,
                              int match_by_bbox, int* dt_match, unsigned char* dt_ignore, int* gt_match,
                              unsigned char* gt_ignore_out, double* sim_sum, int* sim_cnt, void* stream) {
    using namespace pp;
    if (n_cells == 0) return PP_OK;
    PP_REQUIRE(cell_gt_off && cell_dt_off && cell_iou_off && area_rng && iou_thrs && sim_sum && sim_cnt, PP_ERR_INVALID_ARG,
               "pp_exoks_match: NULL argument");
    PP_REQUIRE(N_gt == 0 || (gt_ignore && gt_iscrowd && gt_area && gt_bbox && gt_match && gt_ignore_out), PP_ERR_INVALID_ARG,
               "pp_exoks_match: NULL instance argument");
    PP_REQUIRE(N_dt == 0 || (dt_area && dt_bbox && dt_match && dt_ignore), PP_ERR_INVALID_ARG, "pp_exoks_match: NULL detection argument");
    PP_REQUIRE((N_gt == 0 || N_dt == 0) || ious, PP_ERR_INVALID_ARG, "pp_exoks_match: NULL similarities");
    PP_REQUIRE(n_cells > 0 && L > 0 && A > 0 && T > 0 && T <= 64 && max_gt_per_cell >= 0, PP_ERR_INVALID_ARG, "pp_exoks_match: bad shape");
    const size_t lds = (size_t)(2 + T) * max_gt_per_cell * sizeof(int);
    PP_REQUIRE(lds <= 60 * 1024, PP_ERR_UNSUPPORTED, "pp_exoks_match: too many instances in one image");
    MatchParams p{ious, cell_gt_off, cell_dt_off, cell_iou_off, gt_ignore, gt_iscrowd, gt_area, gt_bbox, dt_area, dt_bbox, area_rng,
                  iou_thrs, dt_match, dt_ignore, gt_match, gt_ignore_out, sim_sum, sim_cnt, n_cells, N_gt, N_dt, L, A, T, match_by_bbox};
    hipLaunchKernelGGL(exoks_match_kernel, dim3(n_cells, L * A), dim3(64), lds, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_exmap_accumulate(const int* dt_match, const unsigned char* dt_ignore, const unsigned char* gt_ignore,
                                   const int* order, const double* dt_score, const double* rec_thrs, int n_cells, int N_gt,
                                   int N_dt, int L, int A, int T, int R, int* chunk_scratch, double* precision, double* recall,
                                   double* scores, void* stream) {
    using namespace pp;
    PP_REQUIRE(rec_thrs && precision && recall && scores, PP_ERR_INVALID_ARG, "pp_exmap_accumulate: NULL argument");
    PP_REQUIRE(N_dt == 0 || (dt_match && dt_ignore && order && dt_score && chunk_scratch), PP_ERR_INVALID_ARG,
               "pp_exmap_accumulate: NULL detection argument");
    PP_REQUIRE(N_gt == 0 || gt_ignore, PP_ERR_INVALID_ARG, "pp_exmap_accumulate: NULL instance argument");
    PP_REQUIRE(L > 0 && A > 0 && T > 0 && R > 0 && R <= 128 && N_gt >= 0 && N_dt >= 0, PP_ERR_INVALID_ARG, "pp_exmap_accumulate: bad shape");
    const int n_chunks = (N_dt + ACC_THREADS - 1) / ACC_THREADS;
    AccParams p{dt_match, dt_ignore, gt_ignore, order, dt_score, rec_thrs, precision, recall, scores, chunk_scratch,
                N_gt, N_dt, L, A, T, R, n_chunks, n_cells};
    hipLaunchKernelGGL(exmap_accumulate_kernel, dim3(L * A * T), dim3(ACC_THREADS), 0, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
