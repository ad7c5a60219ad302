// Library plumbing of libtag_hip.so: error reporting, ABI version, device query.
#include <stdarg.h>
#include <string.h>
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
// ---- developer switches (A/B timing by tools/ and tests/; never needed to run the path).  The launchers read NO environment: a
// switch is set through tag_set_option before the first launch that reads it (launchers cache the value); lib.py forwards the
// TAG_* environment variables of the tool scripts here when it loads the library.
namespace {
struct TagOpt { const char* name; int value; };
TagOpt g_opts[] = {
    {"conv_impl", 0},      // 1 = tap-by-tap conv kernel instead of the halo-tile kernel
    {"halo_lds_pad", 0},   // extra LDS bytes per halo workgroup (holds the kernel to fewer workgroups per CU: tools/hybrid_probe.py)
    {"halo_bn256", 1},     // 0 = 128-cout halo tiles everywhere, 2 = 256-cout tiles wherever they apply
    {"wgrad_wgs", 512},    // workgroups of the bf16 weight-gradient kernels (tools/overlap_probe.py)
    {"conv_rows", 1},      // 0 = no row-streaming bf16 conv kernel
    {"wgrad_dma", 1},      // 0 = no LDS-DMA bf16 weight-gradient kernel
    {"x3_products", 6},    // default product count of the split arithmetic
    {"gemm_big_min", 2048},// tile count from which the dense GEMM takes 128 x 128 tiles
    {"gru_tile4", 1},      // 0 = the 16-row persistent GRU kernels of round 2
    {"gru_xcd", 1},        // 0 = never the L2-resident (same-XCD) publishing
    {"gru_coop", 1},       // 0 = no cooperative launch of the persistent GRU kernels
    {"mha_mfma", 1},       // 0 = VALU attention core
};
}  // namespace
int tag_option(const char* name) {
    for (const TagOpt& o : g_opts)
        if (strcmp(o.name, name) == 0) return o.value;
    return 0;
}
extern "C" int tag_set_option(const char* name, int value) {
    TAG_CHECK_ARG(name != nullptr);
    for (TagOpt& o : g_opts)
        if (strcmp(o.name, name) == 0) { o.value = value; return 0; }
    tag_set_error("tag_set_option: unknown option '%s'", name);
    return TAG_EINVAL;
}

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
