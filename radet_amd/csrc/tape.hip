// Launch tape: replays the host side of a recorded train step (include/radet_hip.h "launch tape", radet_amd/tape.py).
// Host code only -- the ops are calls of this library's own C ABI through the generated thunks (tape_thunks.c) and
// event record / wait operations between the engine's streams.
#include <hip/hip_runtime.h>
#include <string.h>
#include "radet_hip.h"

typedef int (*radet_thunk_fn)(const uint64_t*);
extern "C" const struct RadetTapeThunk { const char* name; radet_thunk_fn fn; int nargs; } radet_tape_thunks[];
extern "C" const int radet_tape_nthunks;

extern "C" int radet_tape_fn_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < radet_tape_nthunks; ++i)
        if (strcmp(radet_tape_thunks[i].name, name) == 0) return i;
    return -1;
}

extern "C" int radet_tape_replay(const RadetTapeOp* ops, int first, int last, int* failed) {
    if (!ops || first < 0 || last < first) return -1;
    for (int i = first; i < last; ++i) {
        const RadetTapeOp& op = ops[i];
        int rc;
        switch (op.kind) {
            case 0:
                rc = (op.fn >= 0 && op.fn < radet_tape_nthunks) ? radet_tape_thunks[op.fn].fn(op.args) : -1;
                break;
            case 1:
                rc = hipEventRecord((hipEvent_t)op.event, (hipStream_t)op.stream) == hipSuccess ? 0 : -2;
                break;
            case 2:
                rc = hipStreamWaitEvent((hipStream_t)op.stream, (hipEvent_t)op.event, 0) == hipSuccess ? 0 : -2;
                break;
            default:
                rc = -1;
        }
        if (rc != 0) {
            if (failed) *failed = i;
            return rc;
        }
    }
    return 0;
}

extern "C" int radet_fill_zero(void* dst, size_t nbytes, void* stream) {
    if (!dst && nbytes) return -1;
    if (!nbytes) return 0;
    return hipMemsetAsync(dst, 0, nbytes, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}

extern "C" int radet_copy_d2d(void* dst, const void* src, size_t nbytes, void* stream) {
    if ((!dst || !src) && nbytes) return -1;
    if (!nbytes) return 0;
    return hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -2;
}

extern "C" int radet_stream_create_cumask(const uint32_t* cu_mask, int nwords, void** stream_out) {
    if (!cu_mask || nwords < 1 || !stream_out) return -1;
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, cu_mask) != hipSuccess) return -2;
    *stream_out = (void*)s;
    return 0;
}
