// extern "C" entry points, part 10: the YOLOX head's tail — box decode, batched on-device SimOTA
// assignment and the detection losses with their gradient (simota.hpp; reference models/detection/yolox/models/yolo_head.py).
#include "host.hpp"
#include "simota.hpp"

using namespace rvt;

namespace {
static bool fill_levels(YoloLevels& lv, const int* level_hw, const int* level_stride, int L, int A) {
    if (L < 1 || L > 8) return false;
    lv.n = L;
    int a = 0;
    for (int l = 0; l < 8; l++) {
        lv.h[l] = l < L ? level_hw[2 * l] : 1;
        lv.w[l] = l < L ? level_hw[2 * l + 1] : 1;
        lv.stride[l] = l < L ? level_stride[l] : 1;
        lv.a0[l] = a;
        if (l < L) { if (lv.h[l] < 1 || lv.w[l] < 1 || lv.stride[l] < 1) return false; a += lv.h[l] * lv.w[l]; }
    }
    lv.a0[8] = a;
    for (int l = L; l < 8; l++) lv.a0[l] = a;
    return a == A;
}
struct SimotaWs {
    float *cost, *iou, *piou, *partial;
    int *nlabel, *meta, *count, *cand, *match;
    size_t bytes;
};
static SimotaWs carve_simota(char* base, int B, int G, int A) {
    SimotaWs w;
    size_t off = 0;
    auto take = [&](size_t n) { char* p = base ? base + off : nullptr; off += (n + 255) & ~(size_t)255; return p; };
    const size_t bga = (size_t)B * imax(G, 1) * A, ba = (size_t)B * A;
    w.cost = (float*)take(bga * 4);
    w.iou = (float*)take(bga * 4);
    w.piou = (float*)take(ba * 4);
    w.partial = (float*)take((size_t)B * ((A + 255) / 256) * 3 * 4);
    w.nlabel = (int*)take((size_t)B * 4);
    w.meta = (int*)take(16);
    w.count = (int*)take(ba * 4);
    w.cand = (int*)take(ba * 4);
    w.match = (int*)take(ba * 4);
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" {

int rvt_yolox_decode(const void* reg_obj, const void* cls, int ld_ro, int ld_cls, int dtype, int B, int H, int W, int stride,
                     int num_classes, int anchor_offset, int A, float* pred_train, float* pred_infer, void* stream) {
    RVT_CHECK(reg_obj && cls && ld_ro >= 5 && ld_cls >= num_classes && num_classes >= 1, "yolox_decode: reg_obj needs >= 5 columns, cls >= %d", num_classes);
    RVT_CHECK(B >= 1 && H >= 1 && W >= 1 && stride >= 1 && anchor_offset >= 0 && anchor_offset + H * W <= A, "yolox_decode: level does not fit the anchor range");
    RVT_CHECK(pred_train || pred_infer, "yolox_decode: nothing to write");
    const int rows = B * H * W;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((yolox_decode_kernel<T>), dim3(grid_for((size_t)rows, 4096)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)reg_obj, (const T*)cls, ld_ro, ld_cls, B, H, W, (float)stride, num_classes,
                                             anchor_offset, A, pred_train, pred_infer));
    return check_launch("yolox_decode");
}

int rvt_yolox_decode_bwd(const float* g_pred, const float* pred_train, const float* col_scale, void* d_reg_obj, void* d_cls, int ld_ro,
                         int ld_cls, int dtype, int B, int H, int W, int stride, int num_classes, int anchor_offset, int A, void* stream) {
    RVT_CHECK(g_pred && pred_train && col_scale && d_reg_obj && d_cls && ld_ro >= 5 && ld_cls >= num_classes, "yolox_decode_bwd: bad arguments");
    RVT_CHECK(B >= 1 && H >= 1 && W >= 1 && anchor_offset >= 0 && anchor_offset + H * W <= A, "yolox_decode_bwd: level does not fit the anchor range");
    const int rows = B * H * W;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((yolox_decode_bwd_kernel<T>), dim3(grid_for((size_t)rows, 4096)), dim3(256), 0, (hipStream_t)stream,
                                             g_pred, pred_train, col_scale, (T*)d_reg_obj, (T*)d_cls, ld_ro, ld_cls, B, H, W, (float)stride,
                                             num_classes, anchor_offset, A));
    return check_launch("yolox_decode_bwd");
}

size_t rvt_simota_ws_bytes(int B, int G, int A) {
    if (B < 1 || G < 0 || A < 1) return 0;
    return carve_simota(nullptr, B, G, A).bytes;
}

int rvt_simota_loss(const float* pred_train, const float* labels, const int* level_hw, const int* level_stride, int L, int B, int G, int A,
                    int num_classes, float* losses, float* g_pred, int* match_out, float* piou_out, void* ws, size_t ws_bytes,
                    void* stream) {
    RVT_CHECK(pred_train && losses && ws && (labels || G == 0), "simota_loss: null argument");
    RVT_CHECK(B >= 1 && B <= 65535 && G >= 0 && G <= 65535 && A >= 1 && num_classes >= 1, "simota_loss: B=%d G=%d A=%d out of range", B, G, A);
    YoloLevels lv;
    RVT_CHECK(fill_levels(lv, level_hw, level_stride, L, A), "simota_loss: the %d levels do not add up to A=%d anchors", L, A);
    const SimotaWs w = carve_simota((char*)ws, B, G, A);
    RVT_CHECK(ws_bytes >= w.bytes, "simota_loss: workspace %zu < %zu bytes", ws_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const dim3 ga((A + 255) / 256, B);
    int* match = match_out ? match_out : w.match;
    float* piou = piou_out ? piou_out : w.piou;
    hipLaunchKernelGGL(simota_count_labels_kernel, dim3(1), dim3(256), 0, st, labels, B, G, w.nlabel, w.meta);
    hipLaunchKernelGGL(simota_cost_kernel, ga, dim3(256), 0, st, pred_train, labels, w.nlabel, lv, B, G, A, num_classes, w.cost, w.iou, w.count);
    if (G > 0) hipLaunchKernelGGL(simota_select_kernel, dim3(G, B), dim3(256), 0, st, w.cost, w.iou, w.nlabel, G, A, w.count, w.cand);
    hipLaunchKernelGGL(simota_resolve_kernel, ga, dim3(256), 0, st, w.cost, w.iou, w.nlabel, w.count, w.cand, G, A, match, piou, w.meta);
    hipLaunchKernelGGL(yolox_loss_kernel, ga, dim3(256), 0, st, pred_train, labels, match, piou, w.meta, G, A, num_classes, w.partial, g_pred);
    hipLaunchKernelGGL(yolox_loss_finalize_kernel, dim3(1), dim3(256), 0, st, w.partial, (int)(ga.x * ga.y), w.meta, losses);
    return check_launch("simota_loss");
}

}  // extern "C"
