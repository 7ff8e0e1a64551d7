/* ORACLE (test infrastructure; never linked or loaded by radet_amd/).
 *
 * Plain-C restatement of the reference's native clustering / voting NMS:
 *   vote_nms, vote_single_dim   radet/ops/vote/vote_ext.cpp:8-35, 70-207
 *   global_vote_nms             radet/ops/vote/vote_ext.cpp:210-353
 *   cluster_nms                 radet/ops/cluster/cluster_ext.cpp:4-87
 * plus class-aware hard NMS with the semantics of mmcv 1.3.18 ops.batched_nms
 * (third-party, not in /root/reference; call sites radet_head.py:160,
 * core/post_processing/bbox_nms.py:69) -- "parity unpinned" for that one entry.
 *
 * All arithmetic is fp32 in the reference's operation order; build with
 * -ffp-contract=off.  Sort: descending score, ties broken by ascending input
 * index (the reference's torch::sort is unstable; fixtures avoid ties).
 * Pinned against oracle/_ref (the reference's own .cpp compiled here) and
 * tests/golden/nms.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int64_t i; } key_t_;

static int cmp_desc(const void *a, const void *b) {
    const key_t_ *x = (const key_t_ *)a, *y = (const key_t_ *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

static int64_t *sorted_order(const float *scores, int64_t n) {
    key_t_ *k = (key_t_ *)malloc(sizeof(key_t_) * (size_t)(n > 0 ? n : 1));
    int64_t *o = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) { k[i].s = scores[i]; k[i].i = i; }
    qsort(k, (size_t)n, sizeof(key_t_), cmp_desc);
    for (int64_t i = 0; i < n; i++) o[i] = k[i].i;
    free(k);
    return o;
}

static float iou_pair(const float *bi, float area_i, const float *bj) {
    float xl = fmaxf(bj[0], bi[0]), yt = fmaxf(bj[1], bi[1]);
    float xr = fminf(bj[2], bi[2]), yb = fminf(bj[3], bi[3]);
    float iw = fmaxf(0.0f, xr - xl), ih = fmaxf(0.0f, yb - yt);
    float inter = iw * ih;
    float area_j = (bj[2] - bj[0]) * (bj[3] - bj[1]);
    return inter / (area_j + area_i - inter);
}

/* score-weighted mean -> weighted sigma -> re-average members within +-1 sigma */
static float vote_dim(const float *s, const float *x, int n) {
    float ssum = 0.0f, v = 0.0f;
    for (int i = 0; i < n; i++) { ssum += s[i]; v += s[i] * x[i]; }
    v = v / ssum;
    float sig = 0.0f;
    for (int i = 0; i < n; i++) sig += s[i] * (x[i] - v) * (x[i] - v);
    sig = sqrtf(sig / ssum);
    float fs = 0.0f, fv = 0.0f;
    for (int i = 0; i < n; i++)
        if ((v - sig <= x[i]) & (x[i] <= v + sig)) { fv += s[i] * x[i]; fs += s[i]; }
    return fv / fs;
}

/* boxes [n,4]; out_boxes [n,4], out_labels [n], out_scores [n] (capacity n). Returns K. */
static int64_t vote_impl(const float *boxes, const float *cluster_scores, const float *vote_scores,
                         const int64_t *labels, int64_t n, float thr, int iou_enable, float sigma,
                         int global, float *out_boxes, int64_t *out_labels, float *out_scores) {
    int64_t *order = sorted_order(cluster_scores, n);
    unsigned char *sup = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    float *cx[4], *cs, *cc;
    for (int d = 0; d < 4; d++) cx[d] = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    cs = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    cc = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t max_label = -1;
    for (int64_t i = 0; i < n; i++) if (labels[i] > max_label) max_label = labels[i];
    unsigned char *label_done = (unsigned char *)calloc((size_t)(max_label + 2), 1);
    int64_t K = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t a = order[i];
        if (sup[a]) continue;
        int64_t la = labels[a];
        if (global) {
            if (la >= 0 && label_done[la]) { sup[a] = 1; continue; }
        }
        const float *bi = boxes + 4 * a;
        float area_i = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        sup[a] = 1;
        if (global && la >= 0) label_done[la] = 1;
        int m = 0;
        for (int d = 0; d < 4; d++) cx[d][m] = bi[d];
        cs[m] = vote_scores[a]; cc[m] = cluster_scores[a]; m++;
        for (int64_t j = i + 1; j < n; j++) {
            int64_t b = order[j];
            if (labels[b] != la || sup[b]) continue;
            const float *bj = boxes + 4 * b;
            float iou = iou_pair(bi, area_i, bj);
            float vs = vote_scores[b];
            if (iou_enable) {
                /* reference: the float overload of exp is the one in scope (checked against oracle/_ref) */
                float f = -(1 - iou) * (1 - iou) / sigma;
                vs = vs * expf(f);
            }
            if (iou > thr) {
                sup[b] = 1;
                for (int d = 0; d < 4; d++) cx[d][m] = bj[d];
                cs[m] = vs; cc[m] = cluster_scores[b]; m++;
            }
        }
        for (int d = 0; d < 4; d++) out_boxes[4 * K + d] = vote_dim(cs, cx[d], m);
        float mx = cc[0];
        for (int t = 1; t < m; t++) if (cc[t] > mx) mx = cc[t];
        out_scores[K] = mx;
        out_labels[K] = la;
        K++;
    }
    for (int d = 0; d < 4; d++) free(cx[d]);
    free(cs); free(cc); free(sup); free(order); free(label_done);
    return K;
}

int64_t oracle_vote_nms(const float *boxes, const float *cluster_scores, const float *vote_scores,
                        const int64_t *labels, int64_t n, float thr, int iou_enable, float sigma,
                        float *out_boxes, int64_t *out_labels, float *out_scores) {
    return vote_impl(boxes, cluster_scores, vote_scores, labels, n, thr, iou_enable, sigma, 0,
                     out_boxes, out_labels, out_scores);
}

int64_t oracle_global_vote_nms(const float *boxes, const float *cluster_scores, const float *vote_scores,
                               const int64_t *labels, int64_t n, float thr, int iou_enable, float sigma,
                               float *out_boxes, int64_t *out_labels, float *out_scores) {
    return vote_impl(boxes, cluster_scores, vote_scores, labels, n, thr, iou_enable, sigma, 1,
                     out_boxes, out_labels, out_scores);
}

/* instance_id[n], cluster_num[n] (size at the head's index, 0 elsewhere) */
void oracle_cluster_nms(const float *boxes, const float *scores, const int64_t *labels, int64_t n,
                        float thr, int64_t *instance_id, int64_t *cluster_num) {
    int64_t *order = sorted_order(scores, n);
    unsigned char *sup = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    memset(instance_id, 0, sizeof(int64_t) * (size_t)n);
    memset(cluster_num, 0, sizeof(int64_t) * (size_t)n);
    int64_t id = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t a = order[i];
        if (sup[a]) continue;
        const float *bi = boxes + 4 * a;
        float area_i = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        int64_t cnt = 1;
        sup[a] = 1;
        for (int64_t j = i + 1; j < n; j++) {
            int64_t b = order[j];
            if (labels[b] != labels[a] || sup[b]) continue;
            if (iou_pair(bi, area_i, boxes + 4 * b) > thr) { instance_id[b] = id; sup[b] = 1; cnt++; }
        }
        instance_id[a] = id;
        cluster_num[a] = cnt;
        id++;
    }
    free(sup); free(order);
}

/* mmcv-style batched hard NMS: boxes offset by label*(max_coord+1) (fp32), greedy, IoU > thr
 * suppresses.  keep[] receives kept input indices in descending-score order. Returns count.
 *
 * PARITY UNPINNED (mmcv 1.3.18 is neither in /root/reference nor in this image).  Knife-edges this
 * restatement had to decide without a reference to check against:
 *   1. the comparison at exact equality: a box is suppressed when IoU > thr (strict), the rule of
 *      mmcv's CPU `nms` (nms_cpu: `if (ovr > iou_threshold) suppressed = 1`), which is the path the
 *      north star's "PyTorch-CPU" reference takes.  mmcv's CUDA kernel (nms_cuda: `devIoU(...) >
 *      threshold` on a bit-mask tile) is strict as well in 1.3.x to the builder's recollection, but
 *      the two were written independently and nobody can run either here: IoU == thr exactly is
 *      undecided by any fixture.  tests/test_gpu_kernels.py::test_nms_ops_bit_exact holds the HIP
 *      kernel to THIS rule (oracle and kernel agree on ties by construction);
 *   2. the IoU denominator: area_a + area_b - inter with offset = 0 (no "+1" pixel convention),
 *      no epsilon; a zero-area pair gives 0/0 = NaN, which compares false (kept);
 *   3. the class offset is added in fp32 before the IoU (as mmcv does), so coordinates above 2^24 /
 *      (labels + 1) lose low bits exactly as in mmcv;
 *   4. score ties: descending score, then ascending input index (torch.sort's order for equal keys
 *      is unspecified; fixtures avoid ties). */
int64_t oracle_batched_nms(const float *boxes, const float *scores, const int64_t *labels, int64_t n,
                           float thr, int class_agnostic, int64_t *keep) {
    float *ob = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float mx = -INFINITY;
    for (int64_t i = 0; i < 4 * n; i++) if (boxes[i] > mx) mx = boxes[i];
    for (int64_t i = 0; i < n; i++) {
        float off = class_agnostic ? 0.0f : (float)labels[i] * (mx + 1.0f);
        for (int d = 0; d < 4; d++) ob[4 * i + d] = boxes[4 * i + d] + off;
    }
    int64_t *order = sorted_order(scores, n);
    unsigned char *sup = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int64_t K = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t a = order[i];
        if (sup[a]) continue;
        keep[K++] = a;
        const float *bi = ob + 4 * a;
        float area_i = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int64_t j = i + 1; j < n; j++) {
            int64_t b = order[j];
            if (sup[b]) continue;
            if (iou_pair(bi, area_i, ob + 4 * b) > thr) sup[b] = 1;
        }
    }
    free(ob); free(order); free(sup);
    return K;
}
