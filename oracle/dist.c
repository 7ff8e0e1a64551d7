/* ORACLE (test infrastructure, never linked into radet_amd/): plain-C restatement of the reference's
 * box-to-distance raster scans, radet/ops/bbox2distance/bbox2distance_ext.cpp:
 *   oracle_mbd  follows FastMBD (:7-124)              -- minimum-barrier distance, `niter` alternating raster passes
 *   oracle_gdt  follows GDT + GeodesicDistanceTransform (:136-236) -- two-pass chamfer geodesic distance
 * Pinned by tests/test_oracle.py against oracle/_ref/ref_bbox2distance_ext.so = that file compiled in place.
 * Arithmetic notes kept from the reference: size_factor uses INTEGER division w*h/(base*base); factor =
 * (float)(alpha*alpha) / size_factor in double; costs in double (MBD) / float (GDT); neighbours visited in the
 * reference's order, each UPDATE seeing the previous one's result. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_mbd(const uint8_t* image, int h, int w, const int64_t* seeds_x, const int64_t* seeds_y, int nseeds,
                float alpha, int niter, int base_size, double* dmap) {
    double size_factor;
    if (h * w < base_size * base_size) size_factor = 400.;
    else size_factor = 400. * (w * h / (base_size * base_size));
    const double factor = alpha * alpha / size_factor;
    int32_t* label = (int32_t*)malloc(sizeof(int32_t) * h * w);
    uint8_t* H = (uint8_t*)malloc((size_t)h * w * 3);
    uint8_t* L = (uint8_t*)malloc((size_t)h * w * 3);
    for (int i = 0; i < h * w; ++i) { label[i] = -1; dmap[i] = 255; }
    memcpy(H, image, (size_t)h * w * 3);
    memcpy(L, image, (size_t)h * w * 3);
    for (int s = 0; s < nseeds; ++s) { label[seeds_y[s] * w + seeds_x[s]] = s; dmap[seeds_y[s] * w + seeds_x[s]] = 0; }
    for (int it = 0; it < niter; ++it) {
        const int fwd = (it % 2 == 0);
        const int step = fwd ? 1 : -1;
        const int ox[2] = {0, fwd ? -1 : 1}, oy[2] = {fwd ? -1 : 1, 0};
        for (int y = fwd ? 0 : h - 1; y != (fwd ? h : -1); y += step)
            for (int x = fwd ? 0 : w - 1; x != (fwd ? w : -1); x += step) {
                const uint8_t* cur = image + ((size_t)y * w + x) * 3;
                for (int k = 0; k < 2; ++k) {
                    const int nx = x + ox[k], ny = y + oy[k];
                    if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                    const int nl = label[ny * w + nx];
                    if (nl < 0) continue;
                    const uint8_t* hh = H + ((size_t)ny * w + nx) * 3;
                    const uint8_t* ll = L + ((size_t)ny * w + nx) * 3;
                    uint8_t mx[3], mn[3];
                    int cc[3];
                    for (int c = 0; c < 3; ++c) {
                        mx[c] = hh[c] > cur[c] ? hh[c] : cur[c];
                        mn[c] = ll[c] < cur[c] ? ll[c] : cur[c];
                        cc[c] = mx[c] - mn[c];
                    }
                    int m = cc[0] > cc[1] ? cc[0] : cc[1];
                    m = m > cc[2] ? m : cc[2];
                    double cost = 0;
                    cost += m / 255.;
                    cost *= cost;
                    const int64_t sy = seeds_y[nl], sx = seeds_x[nl];
                    cost += factor * ((sy - y) * (sy - y) + (sx - x) * (sx - x));
                    if (cost < dmap[y * w + x]) {
                        dmap[y * w + x] = cost;
                        label[y * w + x] = nl;
                        memcpy(H + ((size_t)y * w + x) * 3, mx, 3);
                        memcpy(L + ((size_t)y * w + x) * 3, mn, 3);
                    }
                }
            }
    }
    free(label); free(H); free(L);
}

#define UPD(ci, pi, coef) do { float d_ = dist[pi] + (coef) * (cost[ci] + cost[pi]); \
                               if (dist[ci] > d_) { dist[ci] = d_; label[ci] = label[pi]; } } while (0)

void oracle_gdt(const float* cost, int h, int w, const int64_t* seeds_x, const int64_t* seeds_y, int nseeds, float* dist) {
    const float c1 = 1.0f / 2.0f, c2 = sqrtf(2.0f) / 2.0f;
    int32_t* label = (int32_t*)malloc(sizeof(int32_t) * h * w);
    for (int i = 0; i < h * w; ++i) { label[i] = -1; dist[i] = 255.f; }
    for (int s = 0; s < nseeds; ++s) {
        const int p = (int)(seeds_y[s] * w + seeds_x[s]);
        label[p] = s;
        dist[p] = cost[p];
    }
    for (int j = 1; j < w; ++j) UPD(j, j - 1, c1);
    for (int i = 1; i < h; ++i) {
        const int r = i * w, q = (i - 1) * w;
        int j = 0;
        UPD(r + j, q + j, c1);
        UPD(r + j, q + j + 1, c2);
        for (j = 1; j < w - 1; ++j) {
            UPD(r + j, r + j - 1, c1);
            UPD(r + j, q + j - 1, c2);
            UPD(r + j, q + j, c1);
            UPD(r + j, q + j + 1, c2);
        }
        UPD(r + j, r + j - 1, c1);
        UPD(r + j, q + j - 1, c2);
        UPD(r + j, q + j, c1);
    }
    for (int j = w - 2; j >= 0; --j) UPD((h - 1) * w + j, (h - 1) * w + j + 1, c1);
    for (int i = h - 2; i >= 0; --i) {
        const int r = i * w, q = (i + 1) * w;
        int j = w - 1;
        UPD(r + j, q + j, c1);
        UPD(r + j, q + j - 1, c2);
        for (j = w - 2; j > 0; --j) {
            UPD(r + j, r + j + 1, c1);
            UPD(r + j, q + j + 1, c2);
            UPD(r + j, q + j, c1);
            UPD(r + j, q + j - 1, c2);
        }
        UPD(r + j, r + j + 1, c1);
        UPD(r + j, q + j + 1, c2);
        UPD(r + j, q + j, c1);
    }
    free(label);
}
