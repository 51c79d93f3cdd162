// conv_gn3.hip -- k_conv_gn3 (conv_gn3.h) in a translation unit of its own: its slot-table loop is ~1,800 template instantiations per kernel, and
// ncsnpp.hip already takes the longest of the build.  The shared kernel headers define non-template kernels, so they are included into an anonymous
// namespace here (internal linkage: no second definition of k_gemm_bf16 & co. at link time); the interface to ncsnpp.hip is three plain functions that
// take the launch arguments as bytes (the same GemmArgs layout: the same header).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include <utility>

namespace {
#include "conv_gn3.h"
}
using namespace ncsn;

namespace {
template <int RES, int WM, int WN, int EPI>
bool cfg1() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gn3<RES, WM, WN, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               ConvGn3Cfg<RES, WM, WN>::LDS_BYTES) == hipSuccess;
}
#ifdef NATINF_CG3_TIMELINE
// development build: ONE epilogue (2: GroupNorm partials) per shape -- a sixth of the compile time; every launch runs it (timings only)
unsigned long long* g_cg3_ts = nullptr;
template <int RES, int WM, int WN>
bool cfg_shape() { return cfg1<RES, WM, WN, 2>(); }
#else
template <int RES, int WM, int WN>
bool cfg_shape() { return cfg1<RES, WM, WN, 1>() && cfg1<RES, WM, WN, 2>() && cfg1<RES, WM, WN, 5>() && cfg1<RES, WM, WN, 6>(); }
#endif
template <int RES, int WM, int WN>
void launch_shape(const GemmArgs& g, int epi, hipStream_t s) {
    using Cfg = ConvGn3Cfg<RES, WM, WN>;
    const dim3 grid((unsigned)((g.M / Cfg::BM_) * (g.N / Cfg::BN_)));
#ifdef NATINF_CG3_TIMELINE
    { GemmArgs gt = g; gt.dbg_ts = g_cg3_ts; hipLaunchKernelGGL((k_conv_gn3<RES, WM, WN, 2>), grid, dim3(256), Cfg::LDS_BYTES, s, gt); return; }
#endif
    switch (epi) {
        case 1: hipLaunchKernelGGL((k_conv_gn3<RES, WM, WN, 1>), grid, dim3(256), Cfg::LDS_BYTES, s, g); break;
        case 2: hipLaunchKernelGGL((k_conv_gn3<RES, WM, WN, 2>), grid, dim3(256), Cfg::LDS_BYTES, s, g); break;
        case 5: hipLaunchKernelGGL((k_conv_gn3<RES, WM, WN, 5>), grid, dim3(256), Cfg::LDS_BYTES, s, g); break;
        default: hipLaunchKernelGGL((k_conv_gn3<RES, WM, WN, 6>), grid, dim3(256), Cfg::LDS_BYTES, s, g); break;
    }
}
}  // namespace

// shape 0: 32x32 images, 512 pixels x 128 channels; 1: 32x32, 256 x 256; 2: 16x16, 256 x 256 (one image per tile)
namespace ncsn_cg3 {
__attribute__((visibility("hidden"))) bool configure() { return cfg_shape<32, 4, 1>() && cfg_shape<32, 2, 2>() && cfg_shape<16, 2, 2>(); }
__attribute__((visibility("hidden"))) int tile_rows(int shape) { return shape == 0 ? 512 : 256; }
__attribute__((visibility("hidden"))) int tile_cols(int shape) { return shape == 0 ? 128 : 256; }
__attribute__((visibility("hidden"))) void launch(const void* gemm_args, int shape, int epi, void* stream) {
    GemmArgs g;
    memcpy(&g, gemm_args, sizeof(g));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (shape == 0) launch_shape<32, 4, 1>(g, epi, s);
    else if (shape == 1) launch_shape<32, 2, 2>(g, epi, s);
    else launch_shape<16, 2, 2>(g, epi, s);
}
}  // namespace ncsn_cg3
#ifdef NATINF_CG3_TIMELINE
extern "C" int natinf_debug_cg3_timeline(void* dev_buf128) { g_cg3_ts = reinterpret_cast<unsigned long long*>(dev_buf128); return 0; }
#endif
