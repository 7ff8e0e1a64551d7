"""ORACLE (test infrastructure, never imported by radet_amd/): NumPy restatement of the
reference's visibility-guided positive-sample assigner.

Follows radet/datasets/pipelines/label_assignment.py:
  * candidate test            :57-76   (generate_candidate_cell)
  * mask lookup at the centre :78-86   (cal_sample_pro)
  * per-gt sampling           :88-131  (adapt_cal_k, random_sample: balance_sample,
                                        multiply_samplepro_for_weight and adapt_positive_num
                                        on or off; random_sample_by_distance=True only)
  * ascending-area visit, 'min_area' ambiguity rule, scatter  :136-201 (__call__)
Anchor centres follow core/anchor/anchor_generator.py:206-271 with center_offset=0:
one square anchor per cell centred at (j*stride, i*stride), levels concatenated,
row-major inside a level.

Pinned by tests/golden/assigner_*.npz (outputs of the reference itself run in the
build container, numpy 2.2 semantics for the `0.2 * max` threshold).

Also exposes `legacy_choice`, a transparent restatement of numpy's legacy
RandomState.choice(p=...) that takes the uniforms explicitly; the HIP assigner
implements exactly this algorithm and the tests check both against numpy itself.
"""
import math

import numpy as np

INF = 1e8
EPS = 1e-8
DEFAULT_STRIDES = (8, 16, 32, 64, 128)
DEFAULT_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))


def point_grid(img_h, img_w, strides=DEFAULT_STRIDES):
    """Centres (x, y) float32, anchor side float32 and level id of every point."""
    xs, ys, sizes, lvls = [], [], [], []
    for l, s in enumerate(strides):
        fh, fw = math.ceil(img_h / s), math.ceil(img_w / s)
        jj, ii = np.meshgrid(np.arange(fw), np.arange(fh))
        xs.append((jj.reshape(-1) * s).astype(np.float32))
        ys.append((ii.reshape(-1) * s).astype(np.float32))
        sizes.append(np.full(fh * fw, 8 * s, np.float32))
        lvls.append(np.full(fh * fw, l, np.int64))
    return (np.concatenate(xs), np.concatenate(ys), np.concatenate(sizes), np.concatenate(lvls))


def legacy_choice(p32, size, replace, uniforms):
    """numpy legacy `RandomState.choice(n, size, p=p32, replace=...)` with the uniform
    stream given explicitly (`uniforms` = successive `random_sample()` outputs).
    Returns (indices, n_uniforms_consumed)."""
    p = np.asarray(p32, dtype=np.float64).copy()
    used = 0
    if replace:
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        u = uniforms[used:used + size]
        used += size
        return np.searchsorted(cdf, u, side="right").astype(np.int64), used
    found = np.zeros(size, np.int64)
    n_uniq = 0
    while n_uniq < size:
        k = size - n_uniq
        x = uniforms[used:used + k]
        assert x.shape[0] == k, "uniform stream exhausted"
        used += k
        if n_uniq > 0:
            p[found[:n_uniq]] = 0
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        new = np.searchsorted(cdf, x, side="right")
        seen = set()
        for v in new:                       # first occurrences, in draw order
            if int(v) not in seen:
                seen.add(int(v))
                found[n_uniq] = v
                n_uniq += 1
    return found, used


def assign_points(gt_bboxes, gt_labels, masks, img_shape, rng=None, strides=DEFAULT_STRIDES,
                  regress_ranges=DEFAULT_RANGES, positive_num=10, neg_threshold=0.2, balance_sample=True,
                  multiply_samplepro_for_weight=False, adapt_positive_num=False, random_sample_by_distance=True):
    """Returns (points_to_gt_index int64[N], points_weight float32[N]).

    gt_bboxes f32[G,4], gt_labels i64[G] (unused by the arithmetic), masks [G,H,W] (0/1),
    rng: np.random.RandomState or None (None = the global np.random, like the reference).
    """
    rng = np.random if rng is None else rng
    img_h, img_w = int(img_shape[0]), int(img_shape[1])
    gt_bboxes = np.asarray(gt_bboxes, np.float32).reshape(-1, 4)
    G = gt_bboxes.shape[0]
    cx, cy, anchor_size, lvl = point_grid(img_h, img_w, strides)
    N = cx.shape[0]
    p2g = np.full(N, -1, np.int64)
    wts = np.ones(N, np.float32)
    if G == 0:
        return p2g, wts
    lo = np.asarray([r[0] for r in regress_ranges], np.float32)[lvl][:, None]
    hi = np.asarray([r[1] for r in regress_ranges], np.float32)[lvl][:, None]
    left = cx[:, None] - gt_bboxes[None, :, 0]
    right = gt_bboxes[None, :, 2] - cx[:, None]
    top = cy[:, None] - gt_bboxes[None, :, 1]
    bottom = gt_bboxes[None, :, 3] - cy[:, None]
    min_side = np.minimum(np.minimum(left, top), np.minimum(right, bottom))
    max_side = np.maximum(np.maximum(left, top), np.maximum(right, bottom))
    cand = (min_side > 0.01) & (max_side >= lo) & (max_side <= hi)          # [N, G]
    xi, yi = cx.astype(np.int64), cy.astype(np.int64)
    prob = np.asarray(masks)[:, yi, xi].astype(np.float32).T              # [N, G]
    areas = (gt_bboxes[:, 2] - gt_bboxes[:, 0]) * (gt_bboxes[:, 3] - gt_bboxes[:, 1])
    order = sorted(range(G), key=lambda k: areas[k])                       # stable, like the reference
    for g in order:
        idx = np.nonzero(cand[:, g] & (p2g == -1))[0]
        if idx.shape[0] == 0:
            continue
        p = np.clip(prob[idx, g], np.float32(EPS), None)
        keep = p > (neg_threshold * np.max(p))
        nn_idx = idx[keep]
        nn_p = p[keep]
        n = nn_idx.shape[0]
        sample_p = nn_p / np.sum(nn_p)
        k = positive_num
        if adapt_positive_num:                 # adapt_cal_k (:88-95): sizes of ALL candidate cells, object size = max(w, h)
            sz, cnt_l = np.unique(anchor_size[idx], return_counts=True)
            obj = max(gt_bboxes[g, 2] - gt_bboxes[g, 0], gt_bboxes[g, 3] - gt_bboxes[g, 1])
            dk = ((cnt_l / idx.shape[0]) * np.exp((obj - sz) / (2 * sz))).sum()
            k = int(positive_num * dk + 0.5)
        if n < k and not balance_sample:       # (:112-113) all of them, no draw
            chosen = np.arange(0, n)
        elif random_sample_by_distance:
            chosen = rng.choice(a=n, size=k, p=sample_p, replace=bool(n < k))
        else:                                  # (:111, :118) uniform draw: randint / permutation inside numpy
            chosen = rng.choice(a=n, size=k, replace=bool(n < k))
        uniq, cnt = np.unique(chosen, return_counts=True)
        weight = cnt.astype(np.float32)
        if multiply_samplepro_for_weight:      # (:127-128) the clipped map value, not the normalised probability
            weight *= nn_p[uniq]
        p2g[nn_idx] = 0
        wts[nn_idx] = 0.0
        p2g[nn_idx[uniq]] = g + 1
        wts[nn_idx[uniq]] = weight
    return p2g, wts


def uniforms_from_words(words):
    """RandomState.random_sample() values from consecutive pairs of raw MT19937 outputs (genrand_res53)"""
    w = np.asarray(words, np.uint64)
    n = w.shape[0] // 2
    return ((w[0:2 * n:2] >> np.uint64(5)).astype(np.float64) * 67108864.0 + (w[1:2 * n:2] >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def legacy_bounded(words, pos, rng):
    """numpy legacy bounded integer in [0, rng] (random_interval; the masked path of randint): raw 32-bit outputs & mask until
    <= rng.  Returns (value, new position)."""
    if rng == 0:
        return 0, pos
    mask = rng
    for sh in (1, 2, 4, 8, 16):
        mask |= mask >> sh
    while True:
        v = int(words[pos]) & mask
        pos += 1
        if v <= rng:
            return v, pos


def legacy_choice_uniform(n, size, replace, words):
    """numpy legacy `RandomState.choice(n, size, replace=...)` WITHOUT p, with the stream of raw 32-bit outputs given explicitly:
    randint(0, n, size) with replacement, permutation(n)[:size] (Fisher-Yates from the top) without.  Returns (indices, words
    consumed); checked against numpy itself in tests/test_oracle.py."""
    pos = 0
    if replace:
        out = np.zeros(size, np.int64)
        for k in range(size):
            out[k], pos = legacy_bounded(words, pos, n - 1)
        return out, pos
    perm = np.arange(n, dtype=np.int64)
    for i in range(n - 1, 0, -1):
        v, pos = legacy_bounded(words, pos, i)
        perm[i], perm[v] = perm[v], perm[i]
    return perm[:size].copy(), pos


def assign_points_explicit(gt_bboxes, gt_labels, masks, img_shape, uniforms=None, words=None, **kw):
    """Same as assign_points but driven by an explicit random stream through `legacy_choice` / `legacy_choice_uniform` (the
    algorithms the HIP kernel implements): `uniforms` = successive random_sample() outputs (weighted draws only), or `words` =
    the RandomState's raw 32-bit outputs (what the kernel is handed; both kinds of draw).  Returns (p2g, weights, uniforms
    resp. words consumed)."""

    class _Stream:
        def __init__(self, u, w):
            self.u, self.w, self.used = (None if u is None else np.asarray(u, np.float64)), w, 0

        def choice(self, a, size, p=None, replace=True):
            if p is None:
                idx, k = legacy_choice_uniform(a, size, replace, self.w[self.used:])
            elif self.w is not None:           # uniforms of the weighted draw from the word stream: two words each
                idx, k = legacy_choice(p, size, replace, uniforms_from_words(self.w[self.used:self.used + 16384]))
                k *= 2
            else:
                idx, k = legacy_choice(p, size, replace, self.u[self.used:])
            self.used += k
            return idx

    s = _Stream(uniforms, words)
    p2g, w = assign_points(gt_bboxes, gt_labels, masks, img_shape, rng=s, **kw)
    return p2g, w, s.used
