// Torch-free reproduction of the "fifth stream" cliff (stream budget): a four-stream workload shaped like the train
// step's backward pass -- one dependent chain of short kernels on stream 0, three streams of longer kernels forked from it by
// events and joined at the end -- timed with 4, 5, 6 and 8 streams CREATED in the process (the extra ones idle, or carrying one
// small kernel per step), so that HIP's stream -> hardware-queue mapping is the only thing that changes.  Run it under different
// GPU_MAX_HW_QUEUES settings (read by the runtime at start-up):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_cliff.hip -o tools/_probe/stream_cliff
//   for q in "" 4 8 16; do GPU_MAX_HW_QUEUES=$q tools/_probe/stream_cliff; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void busy(float* out, int iters) {
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 ua = {threadIdx.x + 1u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3f803f80u, blockIdx.x + 1u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
    float s = 0;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double run(int created, int extra_busy, int prio, int steps) {
    std::vector<hipStream_t> st(created);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int i = 0; i < created; ++i) {
        if (prio && i == 0) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
        else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    std::vector<hipEvent_t> ev(64);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    auto step = [&]() {
        int e = 0;
        for (int k = 0; k < 48; ++k) {                       // the chain: 48 dependent ~25 us launches of 300 workgroups
            busy<<<300, 256, 0, st[0]>>>(out, 260);
            if (k % 4 == 0) {                                // every fourth: fork a ~70 us launch of 500 workgroups to a side stream
                hipEventRecord(ev[e], st[0]);
                hipStream_t s = st[1 + (k / 4) % 3];
                hipStreamWaitEvent(s, ev[e], 0);
                busy<<<500, 256, 0, s>>>(out + 512 * 256, 700);
                ++e;
            }
        }
        for (int i = 1; i < 4; ++i) {                        // join
            hipEventRecord(ev[e], st[i]);
            hipStreamWaitEvent(st[0], ev[e], 0);
            ++e;
        }
        for (int i = 4; i < created && extra_busy; ++i) busy<<<8, 256, 0, st[i]>>>(out + 2048 * 256, 50);   // extra streams in use
    };
    for (int w = 0; w < 5; ++w) step();
    hipDeviceSynchronize();
    hipEventRecord(t0, st[0]);
    for (int s = 0; s < steps; ++s) step();
    hipEventRecord(t1, st[0]);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, t0, t1);
    for (auto& s : st) hipStreamDestroy(s);
    for (auto& e : ev) hipEventDestroy(e);
    hipFree(out);
    return ms / steps;
}

// Part 2: WHICH streams collide.  Eight streams created once (creation order = the order HIP hands out hardware queues); the same
// four-stream step on a chosen subset of them, and two independent kernel trains on a chosen pair.
static double run_subset(std::vector<hipStream_t>& st, const int* role, int steps, float* out, std::vector<hipEvent_t>& ev) {
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    auto step = [&]() {
        int e = 0;
        for (int k = 0; k < 48; ++k) {
            busy<<<300, 256, 0, st[role[0]]>>>(out, 260);
            if (k % 4 == 0) {
                hipEventRecord(ev[e], st[role[0]]);
                hipStream_t s = st[role[1 + (k / 4) % 3]];
                hipStreamWaitEvent(s, ev[e], 0);
                busy<<<500, 256, 0, s>>>(out + 512 * 256, 700);
                ++e;
            }
        }
        for (int i = 1; i < 4; ++i) {
            hipEventRecord(ev[e], st[role[i]]);
            hipStreamWaitEvent(st[role[0]], ev[e], 0);
            ++e;
        }
    };
    for (int w = 0; w < 5; ++w) step();
    hipDeviceSynchronize();
    hipEventRecord(t0, st[role[0]]);
    for (int s = 0; s < steps; ++s) step();
    hipEventRecord(t1, st[role[0]]);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, t0, t1);
    return ms / steps;
}
static double run_pair(std::vector<hipStream_t>& st, int a, int b, float* out) {      // two trains of half-device kernels
    hipEvent_t t0, t1, j;
    hipEventCreate(&t0); hipEventCreate(&t1); hipEventCreate(&j);
    hipDeviceSynchronize();
    hipEventRecord(t0, st[a]);
    hipStreamWaitEvent(st[b], t0, 0);
    for (int k = 0; k < 200; ++k) {
        busy<<<128, 256, 0, st[a]>>>(out, 300);
        busy<<<128, 256, 0, st[b]>>>(out + 512 * 256, 300);
    }
    hipEventRecord(j, st[b]);
    hipStreamWaitEvent(st[a], j, 0);
    hipEventRecord(t1, st[a]);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, t0, t1);
    return ms;
}
static void part2() {
    std::vector<hipStream_t> st(8);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    std::vector<hipEvent_t> ev(64);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int sets[][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 2, 4}, {0, 4, 1, 2}, {0, 4, 1, 5}, {0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 6, 7}, {3, 4, 5, 6}};
    printf("  8 streams created once; the four-stream step on streams (chain, side, side, side):\n");
    for (auto& r : sets) printf("    (%d, %d, %d, %d): %.3f ms\n", r[0], r[1], r[2], r[3], run_subset(st, r, 30, out, ev));
    printf("  two independent trains of 200 half-device kernels each (128 workgroups x ~25 us) on a pair of them:\n   ");
    for (int b = 1; b < 8; ++b) printf(" (0, %d): %.2f ms ", b, run_pair(st, 0, b, out));
    printf("\n   ");
    for (int b = 2; b < 8; ++b) printf(" (1, %d): %.2f ms ", b, run_pair(st, 1, b, out));
    printf("\n");
}

int main() {
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", q && *q ? q : "(unset: default 4)");
    if (getenv("CLIFF_PART2")) { part2(); return 0; }
    for (int created : {4, 5, 6, 8}) {
        const double idle = run(created, 0, 0, 30), used = run(created, 1, 0, 30), pr = run(created, 1, 1, 30);
        printf("  %d streams created: 4-stream step %.3f ms (extra streams idle)  %.3f ms (one small kernel per step on each extra stream)"
               "  %.3f ms (... and the chain stream at high priority)\n", created, idle, used, pr);
    }
    return 0;
}
