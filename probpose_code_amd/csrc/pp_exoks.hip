// Ex-OKS similarity between the detections and the ground-truth instances of one (image, category) cell - the quantity the
// reference's evaluator matches on (COCOeval.computeExtendedOks, mmpose/evaluation/metrics/_cocoeval.py:540-707, with
// fix_bbox_aspect_ratio, mmpose/structures/keypoint/keypoints_min_padding.py:68-133). float64 like the numpy original.
// One thread per (visibility level, detection, instance); K (17) keypoints are a serial loop - the cell is tiny, the
// kernel exists so that keypoints and presence probabilities can stay on the device through evaluation.
// pp_exoks_cells is the same arithmetic for every cell of a dataset in one launch (one workgroup per cell, CSR offsets).
#include "pp_common.h"

namespace pp {

struct ExOksParams {
    const double* gt_kpts;   // (G, K, 3)  x, y, v   (v = 3: outside the activation window)
    const double* gt_bbox;   // (G, 4)     x, y, w, h
    const double* gt_area;   // (G)
    const double* dt_kpts;   // (D, K, 3)  x, y, presence probability; already in evaluation order
    const double* sigmas;    // (K)
    const int* gt_vis;       // (n_vis) visibility value of level 1 .. n_vis (level 0 is v > 0)
    double* out;             // (n_vis + 1, D, G)
    int G, D, K, n_vis;
    double confidence_thr;   // NaN: presence probabilities are used clipped to [0, 1] but not binarised
    double padding;
    int use_area, original;
    // dataset form (pp_exoks_cells): instances / detections of cell c are [cell_gt_off[c], cell_gt_off[c + 1]) and
    // [cell_dt_off[c], cell_dt_off[c + 1]); its (n_vis + 1, Dc, Gc) block starts at out + cell_out_off[c]
    const int* cell_gt_off;
    const int* cell_dt_off;
    const long long* cell_out_off;
};

// similarity of detection d (global index) to instance g (global index) at visibility level lvl
__device__ double exoks_pair(const ExOksParams& p, int g, int d, int lvl) {
    const double* gk = p.gt_kpts + (size_t)g * p.K * 3;
    const double* dk = p.dt_kpts + (size_t)d * p.K * 3;
    const double bx = p.gt_bbox[4 * g], by = p.gt_bbox[4 * g + 1], bw = p.gt_bbox[4 * g + 2], bh = p.gt_bbox[4 * g + 3];
    double x0, y0, x1, y1;
    if (p.original) {  // the classic "double the box" ignore region
        x0 = bx - bw;
        x1 = bx + bw * 2;
        y0 = by - bh;
        y1 = by + bh * 2;
    } else {  // fix_bbox_aspect_ratio(xyxy, padding): centre kept, 3:4 aspect, sizes through float32 as the reference's astype
        const double ax1 = bx + bw, ay1 = by + bh;
        const double cx = bx + (ax1 - bx) / 2, cy = by + (ay1 - by) / 2;
        double w = ax1 - bx, h = ay1 - by;
        float nw = (float)w, nh = (float)h;
        if (w == 0) w = 1.0;
        if (h == 0) h = 1.0;
        if (w / h > 0.75) nh = (float)(w / 0.75);
        else nw = (float)(h * 0.75);
        nw *= (float)p.padding;
        nh *= (float)p.padding;
        x0 = cx - (double)(nw / 2.0f);
        x1 = cx + (double)(nw / 2.0f);
        y0 = cy - (double)(nh / 2.0f);
        y1 = cy + (double)(nh / 2.0f);
    }
    const double area = (p.use_area ? p.gt_area[g] : bh * bw * 0.53) + 2.220446049250313e-16;  // + np.spacing(1)
    const bool binarise = p.confidence_thr == p.confidence_thr;
    const int want = lvl == 0 ? -1 : p.gt_vis[lvl - 1];
    int k1 = 0;
    for (int k = 0; k < p.K; ++k) {
        const double v = gk[3 * k + 2];
        k1 += (lvl == 0 ? v > 0 : v == (double)want) ? 1 : 0;
    }
    double sum = 0.0;
    for (int k = 0; k < p.K; ++k) {
        const double xg = gk[3 * k], yg = gk[3 * k + 1], vg = gk[3 * k + 2];
        const double xd = dk[3 * k], yd = dk[3 * k + 1];
        double cd = fmin(fmax(dk[3 * k + 2], 0.0), 1.0);
        if (binarise) cd = cd >= p.confidence_thr ? 1.0 : 0.0;
        const double var = (p.sigmas[k] * 2) * (p.sigmas[k] * 2);
        double dist;
        if (k1 > 0) {
            if (!(lvl == 0 ? vg > 0 : vg == (double)want)) continue;
            const double dx = xd - xg, dy = yd - yg;
            dist = dx * dx + dy * dy;
            if (!p.original) {
                const bool gt_in = vg < 3, pred_in = cd == 1.0, pred_out = cd == 0.0;
                if (!gt_in && pred_in) {  // prediction inside the window, ground truth outside: distance to the window edge
                    const double ex = fmin(xd - x0, x1 - xd), ey = fmin(yd - y0, y1 - yd);
                    dist = ex * ex + ey * ey;
                }
                if (gt_in && pred_out) {  // the other way round: distance of the ground truth to the edge
                    const double ex = fmin(xg - x0, x1 - xg), ey = fmin(yg - y0, y1 - yg);
                    dist = ex * ex + ey * ey;
                }
                if (!gt_in && pred_out) dist = 0.0;  // both outside
            }
        } else {  // no keypoint of this level: distance of the prediction to the (extended) box
            const double dx = fmax(0.0, x0 - xd) + fmax(0.0, xd - x1);
            const double dy = fmax(0.0, y0 - yd) + fmax(0.0, yd - y1);
            dist = dx * dx + dy * dy;
        }
        sum += exp(-(dist / var / area / 2));
    }
    return sum / (double)(k1 > 0 ? k1 : p.K);
}

__global__ void extended_oks_kernel(const ExOksParams p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = p.n_vis + 1;
    if (idx >= L * p.D * p.G) return;
    p.out[idx] = exoks_pair(p, idx % p.G, (idx / p.G) % p.D, idx / (p.G * p.D));
}

__global__ void extended_oks_cells_kernel(const ExOksParams p) {
    const int c = blockIdx.x;
    const int g0 = p.cell_gt_off[c], G = p.cell_gt_off[c + 1] - g0, d0 = p.cell_dt_off[c], D = p.cell_dt_off[c + 1] - d0;
    double* out = p.out + p.cell_out_off[c];
    for (int idx = threadIdx.x; idx < (p.n_vis + 1) * D * G; idx += blockDim.x)
        out[idx] = exoks_pair(p, g0 + idx % G, d0 + (idx / G) % D, idx / (G * D));
}

}  // namespace pp

extern "C" int pp_extended_oks(const double* gt_kpts, const double* gt_bbox, const double* gt_area, const double* dt_kpts,
                               const double* sigmas, const int* gt_visibilities, int G, int D, int K, int n_vis,
                               double confidence_thr, double padding, int use_area, int original, double* out,
                               void* stream) {
    using namespace pp;
    if (G == 0 || D == 0) return PP_OK;
    PP_REQUIRE(gt_kpts && gt_bbox && gt_area && dt_kpts && sigmas && out, PP_ERR_INVALID_ARG, "pp_extended_oks: NULL argument");
    PP_REQUIRE(G > 0 && D > 0 && K > 0 && n_vis >= 0 && (n_vis == 0 || gt_visibilities), PP_ERR_INVALID_ARG,
               "pp_extended_oks: bad shape");
    PP_REQUIRE(padding >= 1.0, PP_ERR_INVALID_ARG, "pp_extended_oks: padding must be >= 1.0");  // _cocoeval.py:560
    ExOksParams p{gt_kpts, gt_bbox, gt_area, dt_kpts, sigmas, gt_visibilities, out, G, D, K, n_vis, confidence_thr, padding,
                  use_area, original, nullptr, nullptr, nullptr};
    const int total = (n_vis + 1) * D * G;
    hipLaunchKernelGGL(extended_oks_kernel, dim3((total + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_exoks_cells(const double* gt_kpts, const double* gt_bbox, const double* gt_area, const double* dt_kpts,
                              const double* sigmas, const int* gt_visibilities, const int* cell_gt_off, const int* cell_dt_off,
                              const long long* cell_out_off, int n_cells, int K, int n_vis, double confidence_thr,
                              double padding, int use_area, int original, double* out, void* stream) {
    using namespace pp;
    if (n_cells == 0) return PP_OK;
    PP_REQUIRE(gt_kpts && gt_bbox && gt_area && dt_kpts && sigmas && cell_gt_off && cell_dt_off && cell_out_off && out,
               PP_ERR_INVALID_ARG, "pp_exoks_cells: NULL argument");
    PP_REQUIRE(n_cells > 0 && K > 0 && n_vis >= 0 && (n_vis == 0 || gt_visibilities), PP_ERR_INVALID_ARG, "pp_exoks_cells: bad shape");
    PP_REQUIRE(padding >= 1.0, PP_ERR_INVALID_ARG, "pp_exoks_cells: padding must be >= 1.0");  // _cocoeval.py:560
    ExOksParams p{gt_kpts, gt_bbox, gt_area, dt_kpts, sigmas, gt_visibilities, out, 0, 0, K, n_vis, confidence_thr, padding,
                  use_area, original, cell_gt_off, cell_dt_off, cell_out_off};
    hipLaunchKernelGGL(extended_oks_cells_kernel, dim3(n_cells), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), p);
    PP_LAUNCH_CHECK();
    return PP_OK;
}
