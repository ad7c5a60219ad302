// Library plumbing of libtag_hip.so: error reporting, ABI version, device query.
#include <stdarg.h>
#include "tag_common.h"

static thread_local char g_err[512] = "";

void tag_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifndef TAG_CSRC_SHA256
#define TAG_CSRC_SHA256 "unknown"
#endif
extern "C" int tag_abi_version(void) { return TAG_ABI_VERSION; }
// sha256 of the kernel sources this binary was compiled from (csrc/Makefile); equals lib.csrc_sha256() for a current build
extern "C" const char* tag_build_id(void) { return TAG_CSRC_SHA256; }
extern "C" const char* tag_last_error(void) { return g_err; }
extern "C" int tag_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

// A stream whose kernels may only run on the compute units set in mask (bit i = CU i of the device; `words` 32-bit words).
// ops.py: the weight-gradient side stream can be kept off a share of the CUs so that the short kernels of the main stream
// always find a free one (TAG_WGRAD_CU_SKIP).
extern "C" int tag_stream_create_cu_mask(const unsigned* mask, int words, void** stream_out) {
    TAG_CHECK_ARG(mask && words > 0 && stream_out);
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    if (e != hipSuccess) {
        tag_set_error("hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return (int)e;
    }
    *stream_out = reinterpret_cast<void*>(st);
    return 0;
}
