// YOLOX head tail: box decode, SimOTA label assignment and the detection losses, batched over the images of a step with no host
// synchronisation.  Reference: models/detection/yolox/models/yolo_head.py
//   :165-246  forward (per-level predictions -> [B][A][5+nc], decode)          -> yolox_decode_kernel
//   :248-267  get_output_and_grid / :269-290 decode_outputs                     -> yolox_decode_kernel, yolox_decode_bwd_kernel
//   :291-443  get_losses: a Python loop over the images, int(nlabel[b]) (:325), torch.cuda.empty_cache (:383)
//   :453-541  get_assignments: candidate anchors, pairwise IoU, class cost, cost matrix  -> simota_cost_kernel
//   :543-575  get_geometry_constraint (centre radius 1.5 strides)                         -> simota_cost_kernel
//   :577-606  simota_matching: dynamic k from the top-10 IoUs, a per-ground-truth topk loop (:580-584), .item() (:596)
//                                                                               -> simota_select_kernel, simota_resolve_kernel
//   losses.py:9-52 IOUloss (loss_type "iou": 1 - iou^2), BCEWithLogits for objectness and class -> yolox_loss_kernel (+ gradient)
// All of it is small integer / float work on [B][G][A] (A = 5040 anchors at 384x640): a handful of launches per step, HBM/L2 resident.
// Ordering rules where the reference leaves them to torch.topk / torch.min: ties go to the lower anchor / ground-truth index.
#pragma once
#include "common.hpp"

namespace rvt {

struct YoloLevels {                       // anchors are ordered level by level, row-major inside a level (yolo_head.py:236-241)
    int n;
    int h[8], w[8], stride[8], a0[9];     // a0[l] = first anchor of level l, a0[n] = A
};

__device__ __forceinline__ void yolo_anchor(const YoloLevels& lv, int a, float& xs, float& ys, float& st) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) l += (i < lv.n && a >= lv.a0[i]) ? 1 : 0;
    const int r = a - lv.a0[l];
    xs = (float)(r % lv.w[l]);
    ys = (float)(r / lv.w[l]);
    st = (float)lv.stride[l];
}

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
// BCEWithLogitsLoss, the numerically stable form torch uses: max(x, 0) - x t + log(1 + exp(-|x|))
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

// ------------------------------------------------------------------------------------------------------------------ decode
// One level: reg_obj [B*H*W][ld_ro] (columns 0-3 box, 4 objectness; the rest padding of the 8-aligned prediction GEMM),
// cls [B*H*W][ld_cls] -> rows a0 .. a0+H*W of pred_train (decoded box, raw logits: the loss input, yolo_head.py:248-267)
// and of pred_infer (decoded box, sigmoid scores: the returned detections, :211-214 and :269-290).
template <class T>
__global__ void __launch_bounds__(256)
yolox_decode_kernel(const T* __restrict__ ro, const T* __restrict__ cl, int ld_ro, int ld_cls, int B, int H, int W, float stride,
                    int nc, int a0, int A, float* __restrict__ pred_train, float* __restrict__ pred_infer) {
    const int hw = H * W, NO = 5 + nc;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < B * hw; r += gridDim.x * 256) {
        const int b = r / hw, p = r % hw;
        const T* rr = ro + (size_t)r * ld_ro;
        const T* cc = cl + (size_t)r * ld_cls;
        const float gx = (float)(p % W), gy = (float)(p / W);
        const float bx = ((float)rr[0] + gx) * stride, by = ((float)rr[1] + gy) * stride;
        const float bw = expf((float)rr[2]) * stride, bh = expf((float)rr[3]) * stride;
        const float ob = (float)rr[4];
        const size_t o = ((size_t)b * A + a0 + p) * NO;
        if (pred_train != nullptr) {
            float* d = pred_train + o;
            d[0] = bx; d[1] = by; d[2] = bw; d[3] = bh; d[4] = ob;
            for (int c = 0; c < nc; c++) d[5 + c] = (float)cc[c];
        }
        if (pred_infer != nullptr) {
            float* d = pred_infer + o;
            d[0] = bx; d[1] = by; d[2] = bw; d[3] = bh; d[4] = sigmoid_exact(ob);
            for (int c = 0; c < nc; c++) d[5 + c] = sigmoid_exact((float)cc[c]);
        }
    }
}

// gradient of pred_train back to the level's prediction maps (padding columns get zeros)
template <class T>
__global__ void __launch_bounds__(256)
yolox_decode_bwd_kernel(const float* __restrict__ g_pred, const float* __restrict__ pred_train, const float* __restrict__ col_scale,
                        T* __restrict__ d_ro, T* __restrict__ d_cl, int ld_ro, int ld_cls, int B, int H, int W, float stride,
                        int nc, int a0, int A) {
    const int hw = H * W, NO = 5 + nc;
    const float s_box = col_scale[0], s_obj = col_scale[1], s_cls = col_scale[2];
    for (int r = blockIdx.x * 256 + threadIdx.x; r < B * hw; r += gridDim.x * 256) {
        const int b = r / hw, p = r % hw;
        const size_t o = ((size_t)b * A + a0 + p) * NO;
        const float* g = g_pred + o;
        const float* d = pred_train + o;
        T* rr = d_ro + (size_t)r * ld_ro;
        T* cc = d_cl + (size_t)r * ld_cls;
        rr[0] = (T)(g[0] * s_box * stride);
        rr[1] = (T)(g[1] * s_box * stride);
        rr[2] = (T)(g[2] * s_box * d[2]);             // d/dt exp(t) stride = the decoded extent
        rr[3] = (T)(g[3] * s_box * d[3]);
        rr[4] = (T)(g[4] * s_obj);
        for (int c = 5; c < ld_ro; c++) rr[c] = (T)0.f;
        for (int c = 0; c < nc; c++) cc[c] = (T)(g[5 + c] * s_cls);
        for (int c = nc; c < ld_cls; c++) cc[c] = (T)0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------ SimOTA
// nlabel[b] = number of label rows with a positive sum; the first nlabel rows are the ground truths (yolo_head.py:309, :336-337).
// meta[0] = sum of nlabel, meta[1] = number of foreground anchors (filled by simota_resolve_kernel).
__global__ void __launch_bounds__(256)
simota_count_labels_kernel(const float* __restrict__ labels, int B, int G, int* __restrict__ nlabel, int* __restrict__ meta) {
    __shared__ int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 256) {
        int n = 0;
        for (int g = 0; g < G; g++) {
            const float* l = labels + ((size_t)b * G + g) * 5;
            n += (l[0] + l[1] + l[2] + l[3] + l[4]) > 0.f ? 1 : 0;
        }
        nlabel[b] = n;
        atomicAdd(&tot, n);
    }
    __syncthreads();
    if (threadIdx.x == 0) { meta[0] = tot; meta[1] = 0; }
}

// One thread per (image, anchor): candidate test (centre inside the 1.5-stride box of ANY ground truth, :543-575), then for every
// ground truth the pairwise IoU (utils/boxes.py:79-102, cxcywh form), and
//   cost = BCE(sqrt(sigmoid(cls) sigmoid(obj)), onehot).sum + 3 (-log(iou + 1e-8)) + 1e6 [outside this ground truth's centre box]
// (:483-506).  Non-candidate anchors get cost +inf and IoU 0: they are never selected and add nothing to the dynamic-k sums.
__global__ void __launch_bounds__(256)
simota_cost_kernel(const float* __restrict__ pred, const float* __restrict__ labels, const int* __restrict__ nlabel, YoloLevels lv,
                   int B, int G, int A, int nc, float* __restrict__ cost, float* __restrict__ iou, int* __restrict__ count) {
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int ng = nlabel[b], NO = 5 + nc;
    count[(size_t)b * A + a] = 0;
    if (ng == 0) return;
    float xs, ys, st;
    yolo_anchor(lv, a, xs, ys, st);
    const float xc = (xs + 0.5f) * st, yc = (ys + 0.5f) * st, rad = st * 1.5f;
    const float* lb = labels + (size_t)b * G * 5;
    bool cand = false;
    for (int g = 0; g < ng; g++) {
        const float gx = lb[g * 5 + 1], gy = lb[g * 5 + 2];
        const float m = fminf(fminf(xc - (gx - rad), yc - (gy - rad)), fminf((gx + rad) - xc, (gy + rad) - yc));
        cand = cand || m > 0.f;
    }
    const float* p = pred + ((size_t)b * A + a) * NO;
    const float px = p[0], py = p[1], pw = p[2], ph = p[3];
    const float so = sigmoid_exact(p[4]);
    for (int g = 0; g < ng; g++) {
        const size_t o = ((size_t)b * G + g) * A + a;
        if (!cand) { cost[o] = INFINITY; iou[o] = 0.f; continue; }
        const float gc = lb[g * 5 + 0], gx = lb[g * 5 + 1], gy = lb[g * 5 + 2], gw = lb[g * 5 + 3], gh = lb[g * 5 + 4];
        const float m = fminf(fminf(xc - (gx - rad), yc - (gy - rad)), fminf((gx + rad) - xc, (gy + rad) - yc));
        const float tlx = fmaxf(gx - gw / 2, px - pw / 2), tly = fmaxf(gy - gh / 2, py - ph / 2);
        const float brx = fminf(gx + gw / 2, px + pw / 2), bry = fminf(gy + gh / 2, py + ph / 2);
        const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
        const float ai = (brx - tlx) * (bry - tly) * en;
        const float v = ai / (gw * gh + pw * ph - ai);
        float cls = 0.f;
        const int gci = (int)gc;
        for (int c = 0; c < nc; c++) {
            const float q = sqrtf(sigmoid_exact(p[5 + c]) * so);
            cls += c == gci ? -fmaxf(logf(q), -100.f) : -fmaxf(logf(1.f - q), -100.f);     // F.binary_cross_entropy clamps at -100
        }
        cost[o] = cls + 3.0f * -logf(v + 1e-8f) + (m > 0.f ? 0.f : 1e6f);
        iou[o] = v;
    }
}

// The workgroup's next element of row[0..A) in the order (key descending, index ascending) strictly after (pk, pi);
// key = sign * row[a].  Returns index -1 when the row is exhausted.
__device__ __forceinline__ void block_next_in_order(const float* __restrict__ row, int A, float sign, float pk, int pi, float& ok, int& oi) {
    __shared__ float sk[4];
    __shared__ int si[4];
    float bk = -INFINITY;
    int bi = -1;
    for (int a = threadIdx.x; a < A; a += 256) {
        const float k = sign * row[a];
        const bool after = pi < 0 || k < pk || (k == pk && a > pi);
        const bool better = bi < 0 || k > bk;                           // ascending a inside a thread: ties keep the lower index
        if (after && better && k == k) { bk = k; bi = a; }
    }
    for (int m = 1; m < 64; m <<= 1) {
        const float k2 = __shfl_xor(bk, m);
        const int i2 = __shfl_xor(bi, m);
        if (i2 >= 0 && (bi < 0 || k2 > bk || (k2 == bk && i2 < bi))) { bk = k2; bi = i2; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sk[threadIdx.x >> 6] = bk; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bk = sk[0]; bi = si[0];
    for (int w = 1; w < 4; w++)
        if (si[w] >= 0 && (bi < 0 || sk[w] > bk || (sk[w] == bk && si[w] < bi))) { bk = sk[w]; bi = si[w]; }
    ok = bk; oi = bi;
}

// One workgroup per (ground truth, image): dynamic k = clamp(int(sum of the 10 largest IoUs), min 1), then the k cheapest
// anchors vote for this ground truth (yolo_head.py:577-586).
__global__ void __launch_bounds__(256)
simota_select_kernel(const float* __restrict__ cost, const float* __restrict__ iou, const int* __restrict__ nlabel, int G, int A,
                     int* __restrict__ count, int* __restrict__ cand_g) {
    const int g = blockIdx.x, b = blockIdx.y;
    if (g >= nlabel[b]) return;
    const float* ci = iou + ((size_t)b * G + g) * A;
    const float* cc = cost + ((size_t)b * G + g) * A;
    float pk = 0.f, s = 0.f;
    int pi = -1;
    for (int r = 0; r < 10 && r < A; r++) {
        float k; int i;
        block_next_in_order(ci, A, 1.f, pk, pi, k, i);
        if (i < 0) break;
        s += k; pk = k; pi = i;
    }
    const int dyn_k = ((int)s > 1 ? (int)s : 1);
    pk = 0.f; pi = -1;
    for (int r = 0; r < dyn_k; r++) {
        float k; int i;
        block_next_in_order(cc, A, -1.f, pk, pi, k, i);
        if (i < 0 || k == -INFINITY) break;                 // no candidate anchors left (the reference's topk would raise here)
        if (threadIdx.x == 0) {
            atomicAdd(count + (size_t)b * A + i, 1);
            cand_g[(size_t)b * A + i] = g;                  // only read back where exactly one ground truth voted
        }
        pk = k; pi = i;
    }
}

// One thread per (image, anchor): anchors with several votes go to the ground truth of least cost (:588-594); writes the matched
// ground-truth index (-1 = background) and the IoU with it (:598-605), counts the foreground anchors.
__global__ void __launch_bounds__(256)
simota_resolve_kernel(const float* __restrict__ cost, const float* __restrict__ iou, const int* __restrict__ nlabel,
                      const int* __restrict__ count, const int* __restrict__ cand_g, int G, int A, int* __restrict__ match,
                      float* __restrict__ piou, int* __restrict__ meta) {
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    int fg = 0;
    if (a < A) {
        const size_t o = (size_t)b * A + a;
        const int n = count[o];
        int g = -1;
        if (n == 1) g = cand_g[o];
        else if (n > 1) {
            float bc = INFINITY;
            for (int j = 0; j < nlabel[b]; j++) {
                const float c = cost[((size_t)b * G + j) * A + a];
                if (g < 0 || c < bc) { bc = c; g = j; }
            }
        }
        match[o] = g;
        piou[o] = g >= 0 ? iou[((size_t)b * G + g) * A + a] : 0.f;
        fg = g >= 0;
    }
    __shared__ int tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    if (fg) atomicAdd(&tot, 1);
    __syncthreads();
    if (threadIdx.x == 0 && tot) atomicAdd(meta + 1, tot);
}

// ------------------------------------------------------------------------------------------------------------------ losses
// One thread per (image, anchor): objectness BCE against the foreground flag for every anchor; for foreground anchors the IoU loss
// 1 - iou^2 against the matched box (losses.py:17-35) and the class BCE against onehot * matched IoU (yolo_head.py:386-388).
// Per-workgroup partial sums (fixed order: reproducible), and - when g_pred is given - the gradient of
// (sum iou, sum obj, sum cls) / max(num_fg, 1) with respect to pred_train, each loss in its own columns (box / objectness / class).
__global__ void __launch_bounds__(256)
yolox_loss_kernel(const float* __restrict__ pred, const float* __restrict__ labels, const int* __restrict__ match,
                  const float* __restrict__ piou, const int* __restrict__ meta, int G, int A, int nc, float* __restrict__ partial,
                  float* __restrict__ g_pred) {
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x, NO = 5 + nc;
    const float inv_nf = 1.0f / (float)(meta[1] > 1 ? meta[1] : 1);
    float l_iou = 0.f, l_obj = 0.f, l_cls = 0.f;
    if (a < A) {
        const size_t o = (size_t)b * A + a;
        const float* p = pred + o * NO;
        float* gp = g_pred != nullptr ? g_pred + o * NO : nullptr;
        const int g = match[o];
        const float t_obj = g >= 0 ? 1.f : 0.f;
        l_obj = bce_logits(p[4], t_obj);
        if (gp) gp[4] = (sigmoid_exact(p[4]) - t_obj) * inv_nf;
        if (g >= 0) {
            const float* l = labels + ((size_t)b * G + g) * 5;
            const float px = p[0], py = p[1], pw = p[2], ph = p[3], tx = l[1], ty = l[2], tw = l[3], th = l[4];
            const float plx = px - pw / 2, ply = py - ph / 2, prx = px + pw / 2, pry = py + ph / 2;
            const float tlx = fmaxf(plx, tx - tw / 2), tly = fmaxf(ply, ty - th / 2);
            const float brx = fminf(prx, tx + tw / 2), bry = fminf(pry, ty + th / 2);
            const bool en = tlx < brx && tly < bry;
            const float wi = brx - tlx, hi = bry - tly;
            const float ai = en ? wi * hi : 0.f;
            const float au = pw * ph + tw * th - ai + 1e-16f;
            const float v = ai / au;
            l_iou = 1.f - v * v;
            if (gp) {
                // d iou = dI (1/U + I/U^2) - dP I/U^2 with U = P + T - I;  dI through whichever corner the prediction supplies
                const float dv = -2.f * v * inv_nf;
                const float k_i = en ? dv * (1.f / au + ai / (au * au)) : 0.f, k_p = -dv * ai / (au * au);
                const float s_l = plx > tx - tw / 2 ? 1.f : 0.f, s_r = prx < tx + tw / 2 ? 1.f : 0.f;   // tl / br taken from the prediction
                const float s_t = ply > ty - th / 2 ? 1.f : 0.f, s_b = pry < ty + th / 2 ? 1.f : 0.f;
                gp[0] = k_i * hi * (s_r - s_l);
                gp[1] = k_i * wi * (s_b - s_t);
                gp[2] = k_i * hi * 0.5f * (s_r + s_l) + k_p * ph;
                gp[3] = k_i * wi * 0.5f * (s_b + s_t) + k_p * pw;
            }
            const int gc = (int)l[0];
            const float pv = piou[o];
            for (int c = 0; c < nc; c++) {
                const float t = c == gc ? pv : 0.f;
                l_cls += bce_logits(p[5 + c], t);
                if (gp) gp[5 + c] = (sigmoid_exact(p[5 + c]) - t) * inv_nf;
            }
        } else if (gp) {
            gp[0] = gp[1] = gp[2] = gp[3] = 0.f;
            for (int c = 0; c < nc; c++) gp[5 + c] = 0.f;
        }
    }
    __shared__ float red[3][4];
    float v3[3] = {l_iou, l_obj, l_cls};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float s = v3[i];
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// losses[0..4] = loss, 5 * iou loss, objectness loss, class loss, num_fg / max(num_gts, 1)   (yolo_head.py:432-443)
__global__ void __launch_bounds__(256)
yolox_loss_finalize_kernel(const float* __restrict__ partial, int nblk, const int* __restrict__ meta, float* __restrict__ losses) {
    __shared__ double red[3][256];                               // fixed summation order: bit-reproducible from run to run
    double s[3] = {0.0, 0.0, 0.0};
    for (int k = threadIdx.x; k < nblk; k += 256)
        for (int i = 0; i < 3; i++) s[i] += (double)partial[(size_t)k * 3 + i];
    for (int i = 0; i < 3; i++) red[i][threadIdx.x] = s[i];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int i = 0; i < 3; i++) red[i][threadIdx.x] += red[i][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float nf = (float)(meta[1] > 1 ? meta[1] : 1);
    float v[3];
    for (int i = 0; i < 3; i++) {
        v[i] = (float)red[i][0] / nf * (i == 0 ? 5.0f : 1.0f);
        losses[1 + i] = v[i];
    }
    losses[0] = v[0] + v[1] + v[2];
    losses[4] = nf / (float)(meta[0] > 1 ? meta[0] : 1);
}

}  // namespace rvt
