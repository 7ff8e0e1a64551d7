// Box-to-distance raster scans of the mask-free sampler (GenerateDistanceMap(with_gt_mask=False)) on the GPU:
//   radet_mbd : minimum-barrier distance, `niter` alternating raster passes  (reference: FastMBD,
//               radet/ops/bbox2distance/bbox2distance_ext.cpp:7-124, a single-threaded host loop)
//   radet_gdt : two-pass chamfer geodesic distance                          (GeodesicDistanceTransform + GDT, :136-236)
// A raster scan is a recurrence: pixel (y, x) needs its already-updated neighbours.  MBD uses (y-1, x) and (y, x-1), so
// all pixels of an anti-diagonal x + y = t are independent; GDT also uses (y-1, x+1), so the independent sets are the
// skewed diagonals x + 2y = t.  One workgroup per box crop walks the diagonals with one barrier per step (latency
// bound: 10^3 steps of <= 300 pixels), the crops of a batch run on different CUs.  Every pixel performs the
// reference's neighbour updates in the reference's order and arithmetic (double for MBD, float for GDT, no FMA
// contraction), so the result is bit-identical to the sequential scan.
#include "common.h"
#include "radet_hip.h"

struct DistImg { int px_off, h, w, seed_off, nseeds; };

__global__ __launch_bounds__(1024) void mbd_kernel(const uint8_t* __restrict__ images, const int* __restrict__ desc,
                                                   const int* __restrict__ seeds_x, const int* __restrict__ seeds_y,
                                                   double factor_num, int niter, int base_size, double* dmap_all,
                                                   int* label_all, uint8_t* H_all, uint8_t* L_all) {
    const int* d = desc + blockIdx.x * 5;
    const int px_off = d[0], h = d[1], w = d[2], nseeds = d[4];
    const int* sx = seeds_x + d[3];
    const int* sy = seeds_y + d[3];
    const uint8_t* img = images + (size_t)px_off * 3;
    double* dmap = dmap_all + px_off;
    int* label = label_all + px_off;
    uint8_t* H = H_all + (size_t)px_off * 3;
    uint8_t* L = L_all + (size_t)px_off * 3;
    const int tid = threadIdx.x, n = h * w;
    double size_factor;                                      // integer division, as in the reference
    if (h * w < base_size * base_size) size_factor = 400.;
    else size_factor = 400. * (w * h / (base_size * base_size));
    const double factor = factor_num / size_factor;
    for (int p = tid; p < n; p += 1024) { label[p] = -1; dmap[p] = 255.0; }
    for (int p = tid; p < n * 3; p += 1024) { H[p] = img[p]; L[p] = img[p]; }
    __syncthreads();
    if (tid == 0)
        for (int s = 0; s < nseeds; ++s) { label[sy[s] * w + sx[s]] = s; dmap[sy[s] * w + sx[s]] = 0.0; }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const bool fwd = (it & 1) == 0;
        for (int t = 0; t < w + h - 1; ++t) {
            const int ylo = t - (w - 1) > 0 ? t - (w - 1) : 0, yhi = t < h - 1 ? t : h - 1;
            for (int yy = ylo + tid; yy <= yhi; yy += 1024) {
                const int xx = t - yy;
                const int y = fwd ? yy : h - 1 - yy, x = fwd ? xx : w - 1 - xx;
                const int p = y * w + x;
                const uint8_t c0 = img[p * 3], c1 = img[p * 3 + 1], c2 = img[p * 3 + 2];
                double dm = dmap[p];
                bool upd = false;
                int nlab = 0;
                uint8_t mxo[3], mno[3];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nx = x + (k == 0 ? 0 : (fwd ? -1 : 1)), ny = y + (k == 0 ? (fwd ? -1 : 1) : 0);
                    if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                    const int q = ny * w + nx;
                    const int nl = label[q];
                    if (nl < 0) continue;
                    const uint8_t h0 = H[q * 3], h1 = H[q * 3 + 1], h2 = H[q * 3 + 2];
                    const uint8_t l0 = L[q * 3], l1 = L[q * 3 + 1], l2 = L[q * 3 + 2];
                    const uint8_t m0 = h0 > c0 ? h0 : c0, m1 = h1 > c1 ? h1 : c1, m2 = h2 > c2 ? h2 : c2;
                    const uint8_t n0 = l0 < c0 ? l0 : c0, n1 = l1 < c1 ? l1 : c1, n2 = l2 < c2 ? l2 : c2;
                    const int e0 = m0 - n0, e1 = m1 - n1, e2 = m2 - n2;
                    int m = e0 > e1 ? e0 : e1;
                    m = m > e2 ? m : e2;
                    double cost = 0;
                    cost += m / 255.;
                    cost *= cost;
                    const long long syv = sy[nl], sxv = sx[nl];
                    cost += factor * (double)((syv - y) * (syv - y) + (sxv - x) * (sxv - x));
                    if (cost < dm) {
                        dm = cost;
                        upd = true;
                        nlab = nl;
                        mxo[0] = m0; mxo[1] = m1; mxo[2] = m2;
                        mno[0] = n0; mno[1] = n1; mno[2] = n2;
                    }
                }
                if (upd) {
                    dmap[p] = dm;
                    label[p] = nlab;
                    H[p * 3] = mxo[0]; H[p * 3 + 1] = mxo[1]; H[p * 3 + 2] = mxo[2];
                    L[p * 3] = mno[0]; L[p * 3 + 1] = mno[1]; L[p * 3 + 2] = mno[2];
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void gdt_upd(float* dist, int* label, const float* cost, int ci, int pi, float coef) {
    const float dd = dist[pi] + coef * (cost[ci] + cost[pi]);
    if (dist[ci] > dd) { dist[ci] = dd; label[ci] = label[pi]; }
}

__global__ __launch_bounds__(1024) void gdt_kernel(const float* __restrict__ cost_all, const int* __restrict__ desc,
                                                   const int* __restrict__ seeds_x, const int* __restrict__ seeds_y,
                                                   float* dist_all, int* label_all) {
    const int* d = desc + blockIdx.x * 5;
    const int px_off = d[0], h = d[1], w = d[2], nseeds = d[4];
    const int* sx = seeds_x + d[3];
    const int* sy = seeds_y + d[3];
    const float* cost = cost_all + px_off;
    float* dist = dist_all + px_off;
    int* label = label_all + px_off;
    const int tid = threadIdx.x, n = h * w;
    const float c1 = 1.0f / 2.0f, c2 = sqrtf(2.0f) / 2.0f;
    for (int p = tid; p < n; p += 1024) { label[p] = -1; dist[p] = 255.f; }
    __syncthreads();
    if (tid == 0)
        for (int s = 0; s < nseeds; ++s) { const int p = sy[s] * w + sx[s]; label[p] = s; dist[p] = cost[p]; }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const int dir = pass == 0 ? 1 : -1;
        // scan coordinates (si, sj) run forward in both passes; the backward pass mirrors them
        for (int t = 0; t <= (w - 1) + 2 * (h - 1); ++t) {
            const int ilo = t - (w - 1) > 0 ? (t - (w - 1) + 1) / 2 : 0, ihi = t / 2 < h - 1 ? t / 2 : h - 1;
            for (int si = ilo + tid; si <= ihi; si += 1024) {
                const int sj = t - 2 * si;
                const int i = pass == 0 ? si : h - 1 - si, j = pass == 0 ? sj : w - 1 - sj;
                const int ci = i * w + j;
                const int left = ci - dir, up = ci - dir * w;
                if (si == 0) {
                    if (sj > 0) gdt_upd(dist, label, cost, ci, left, c1);
                } else if (sj == 0) {
                    gdt_upd(dist, label, cost, ci, up, c1);
                    gdt_upd(dist, label, cost, ci, up + dir, c2);
                } else {
                    gdt_upd(dist, label, cost, ci, left, c1);
                    gdt_upd(dist, label, cost, ci, up - dir, c2);
                    gdt_upd(dist, label, cost, ci, up, c1);
                    if (sj < w - 1) gdt_upd(dist, label, cost, ci, up + dir, c2);
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

extern "C" size_t radet_mbd_ws_bytes(size_t total_px) { return total_px * (4 + 3 + 3) + 64; }

extern "C" int radet_mbd(const uint8_t* images, const int* img_desc_dev, int nimg, const int* seeds_x, const int* seeds_y,
                         float alpha, int niter, int base_size, double* dmap, size_t total_px, void* ws, void* stream) {
    if (nimg <= 0) return RADET_OK;
    if (niter < 0 || base_size <= 0 || ws == nullptr) return RADET_ERR_ARG;
    char* p = (char*)ws;
    int* label = (int*)p; p += total_px * 4;
    uint8_t* H = (uint8_t*)p; p += total_px * 3;
    uint8_t* L = (uint8_t*)p;
    const double num = alpha * alpha;                        // float product, then promoted (as the reference)
    hipLaunchKernelGGL(mbd_kernel, dim3(nimg), dim3(1024), 0, (hipStream_t)stream, images, img_desc_dev, seeds_x, seeds_y,
                       num, niter, base_size, dmap, label, H, L);
    return radet_check_launch();
}

extern "C" int radet_gdt(const float* cost, const int* img_desc_dev, int nimg, const int* seeds_x, const int* seeds_y,
                         float* dist, void* ws_labels, void* stream) {
    if (nimg <= 0) return RADET_OK;
    if (ws_labels == nullptr) return RADET_ERR_ARG;
    hipLaunchKernelGGL(gdt_kernel, dim3(nimg), dim3(1024), 0, (hipStream_t)stream, cost, img_desc_dev, seeds_x, seeds_y, dist,
                       (int*)ws_labels);
    return radet_check_launch();
}
