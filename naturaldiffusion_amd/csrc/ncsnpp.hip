// ncsnpp.hip -- host side of the NCSN++ denoiser engine: static execution plan, weight packing,
// forward; C ABI of include/natinf_ncsnpp.h.  Kernels: ncsnpp_kernels.h.
//
// Design (MI355X-first, not a translation of the reference's nn.Module tree):
//   * the network is compiled once into a flat op list (~330 launches) over ONE caller-supplied
//     workspace; every tensor offset is "bytes per image", so a plan built once serves any batch;
//   * activations are NHWC bf16; U-Net skip tensors are written by their producer straight into the
//     channel slice of the concat buffer their consumer will read (no torch.cat copies), and an up-path
//     block writes its output into the first channel slice of the next block's concat buffer;
//   * a res-block is 6 launches: GN-stats, GN-apply(+SiLU, +up/down of both branches), conv3x3 GEMM
//     (+bias +time-embedding row), GN-stats, GN-apply, conv3x3 GEMM whose K range is extended by the
//     1x1 shortcut conv (one GEMM computes Conv_1(h) + Conv_2(x)) with the residual add and the
//     1/sqrt(2) rescale in its epilogue;
//   * the 44 per-block time-embedding projections (Dense_0) are one GEMM at the top of the forward.
//
// Reference: deps/score_sde_pytorch/models/ncsnpp.py:232-381, layerspp.py:75-91,242-274,
// layers.py:515-555, up_or_down_sampling.py:59-69.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "natinf_ncsnpp.h"
#include "ncsnpp_kernels.h"
#include "gemm_dma.h"
#include "gemm_w128.h"
#include "conv_ring.h"
#include "conv_gn.h"
#include "conv_gn2.h"
#include "head_conv.h"
#ifdef NATINF_DEV                  // superseded kernels kept for A/B runs: development builds only (make EXTRA=-DNATINF_DEV)
#include "conv_patch.h"
#include "gemm_8phase.h"
#endif
#include "gemm_fp8.h"
#include "attn_fused.h"
#include "attn256.h"
#include "attn_qkv.h"
#include "attn_blk256.h"
#include "flash_attn.h"

using namespace ncsn;

namespace {

constexpr int NF = 128, IMG = 32, TEMB = 512, NLEVEL = 4;      // res-blocks per level: 4 (cifar10_ddpmpp_continuous) or 2 (the `ddpm` network), Builder::NUM_RES
constexpr int CH_MULT[NLEVEL] = {1, 2, 2, 2};
constexpr float GN_EPS = 1e-6f;
constexpr float INV_SQRT2 = 0.70710678118654752440f;
inline bool attn_at(int res) { return res == 16; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

enum Kind { K_LIN, K_CONV, K_RES, K_ATTN, K_GN, K_DOWN, K_UP };      // K_DOWN / K_UP: the plain resampling convolutions of the `ddpm` network
const char* kind_name(int k) { static const char* n[] = {"lin", "conv", "res", "attn", "gn", "down", "up"}; return n[k]; }

struct Mod { int idx, kind, cin, cout, up, down, res; int64_t poff; };

// tensor reference inside the workspace: `off` is BYTES PER IMAGE (actual = off * B)
// coff: channel offset inside a wider buffer; pad = 1: stored with a one-pixel zero border (3x3 GEMM inputs)
struct TRef { int64_t off = -1; int C = 0, ld = 0, res = 0, coff = 0, pad = 0; };

struct Ctx {                     // per-forward launch context
    int B; unsigned char* ws; const unsigned char* wp; hipStream_t stream;
    const float* x; const float* labels; float* out;
    int* part_bm;                // [n_parts] block-tile rows (BM) of the GEMM variant that wrote each partial table
    hipStream_t stream2 = nullptr; hipEvent_t* ev = nullptr;      // MMDiT engine: the text stream's own HIP stream and the fork / join events (null: everything on `stream`)
    unsigned char* fin_done = nullptr;   // [n_parts] per FORWARD, like part_bm: the launch that wrote partial table i also wrote its consumer's GroupNorm table (see Fin).  Not plan
                                         // state: a description pass (scratch array) or a second thread's forward on a shared plan cannot flip it under a running forward
    bf16* act(const TRef& t) const { return reinterpret_cast<bf16*>(ws + t.off * B) + t.coff; }
    template <class T> T* at(int64_t off) const { return reinterpret_cast<T*>(ws + off * B); }
    template <class T> const T* w(int64_t off) const { return reinterpret_cast<const T*>(wp + off); }
};
using OpFn = std::function<void(const Ctx&)>;
enum { CLS_GEMM = 0, CLS_OTHER = 1, CLS_CONV_GN = 2, CLS_CONV_GN8 = 3, N_CLS = 4 };      // CLS_CONV_GN: launches of the fused GroupNorm + SiLU + 3x3 conv kernel at 32x32 / 16x16; CLS_CONV_GN8: its 8x8 instantiation

struct PackCtx { const float* params; unsigned char* packed; hipStream_t stream; };
using PackFn = std::function<void(const PackCtx&)>;

// first-fit arena over "bytes per image"
struct Arena {
    bool keep = false; int64_t top = 0, peak = 0;
    std::map<int64_t, int64_t> free_;      // off -> size
    std::map<int64_t, int64_t> live_;
    int64_t alloc(int64_t bytes) {
        bytes = align_up(bytes, 256);
        if (!keep)
            for (auto it = free_.begin(); it != free_.end(); ++it)
                if (it->second >= bytes) {
                    const int64_t off = it->first, rest = it->second - bytes;
                    free_.erase(it);
                    if (rest) free_[off + bytes] = rest;
                    live_[off] = bytes;
                    return off;
                }
        const int64_t off = top; top += bytes; if (top > peak) peak = top;
        live_[off] = bytes;
        return off;
    }
    void release(int64_t off) {
        if (keep || off < 0) return;
        auto it = live_.find(off);
        if (it == live_.end()) return;
        int64_t o = off, s = it->second;
        live_.erase(it);
        auto nx = free_.lower_bound(o);
        if (nx != free_.end() && o + s == nx->first) { s += nx->second; nx = free_.erase(nx); }
        if (nx != free_.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == o) { o = pv->first; s += pv->second; free_.erase(pv); } }
        if (o + s == top) top = o; else free_[o] = s;
    }
};

}  // namespace

struct natinf_ncsnpp {
    int flags = 0;
    std::vector<Mod> mods;
    int64_t n_params = 0;
    std::vector<OpFn> ops;
    std::vector<int> op_cls;             // CLS_* per op (profiling)
    std::vector<PackFn> packs;
    // profiling: one HIP event pair per op while enabled
    bool prof = false;
    struct Rec { hipEvent_t a, b; int cls; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::map<int, TRef> taps;            // module idx -> output tensor
    int64_t ws_per_image = 0, packed_bytes = 0;
    uint64_t plan_sig = 0;               // the natinf_set_* switches a plan reads when it is BUILT (they decide the pack offsets): natinf_ncsnpp_share compares them
    const unsigned char* packed = nullptr;
    bool attr_set = false;
    int last_B = 0; unsigned char* last_ws = nullptr;
    std::vector<int> part_bm;            // see Ctx::part_bm
    std::vector<unsigned char> fin_done; // see Ctx::fin_done
};

// k_conv_gn3 (conv_gn3.h / conv_gn3.hip: one wave per SIMD, 128 x 128 wave tiles, slot-table K loop) -- a translation unit of its own
bool configure_conv_ring();          // inception_engine.inc: the k_conv_ring instantiations' LDS sizes
namespace { bool configure_dit_attention(); }      // dit_engine.inc: the row-major-v forms of k_attn_fused
namespace ncsn_cg3 { bool configure(); int tile_rows(int shape); int tile_cols(int shape); void launch(const void* gemm_args, int shape, int epi, void* stream); }

namespace {

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
int g_epi_fp32_slab = 0;           // natinf_set_gemm_epilogue(1): every launch takes the fp32-slab epilogue (A/B runs)
GemmArgs gemm_defaults() {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.epi_fp32_slab = g_epi_fp32_slab;
    g.taps = 1; g.batch = 1; g.scale = 1.0f; g.act = ACT_NONE; g.c_mode = OUT_BF16;
    return g;
}
// ------------------------------------------------------------------------------------------------
// GEMM kernel variants and the choice among them
// ------------------------------------------------------------------------------------------------
constexpr int NUM_CU = 256;
enum GemmVariant {
    V_AUTO = 0, V_GENERIC = 1,
    V_DMA_256x256 = 2, V_DMA_256x128 = 3, V_DMA_128x128 = 4,          // 2-stage, BK = 64
    V_RING_256x256 = 5, V_RING_256x128 = 6, V_RING_128x128 = 7, V_RING_64x128 = 8,   // NS-slot ring, BK = 32
    V_RING_256x128_W4 = 9, V_DMA_256x128_W4 = 10,                     // 4 waves, wave tile 128x64 (less LDS read traffic per MFMA)
    V_DMA_256x256_S = 11, V_DMA_128x128_S = 12,                       // two-stage with the DMA issue spread between MFMA groups
    V_DMA_512x128 = 13,                                                // 8 waves x (128x64), all 160 KiB of LDS
    V_PATCH_256x256 = 14, V_PATCH_256x128 = 15,                        // 3x3 conv with an LDS-resident input patch
    V_DMA_256x256_P = 16, V_DMA_128x128_P = 17, V_DMA_256x128W4_P = 18,  // two-stage + hand-counted LDS fragment pipeline
    V_8PH_256x256 = 19, V_8PH_NOPRIO = 20, V_8PH_READFIRST = 21, V_8PH_BOTH = 22,   // phase-interleaved schedule, counted vmcnt (gemm_8phase.h)
    V_FP8_256x256 = 23,                                                 // fp8 e4m3 operands (gemm_fp8.h); selected by GemmArgs::deq_m/deq_n callers only
    V_ABL_NODMA = 24, V_ABL_NOMFMA = 25,
    V_DMA_256x256_H = 26, V_DMA_512x128_H = 27,                         // hand pipeline, DMA issued by one wave per SIMD only
    V_CONV_GN = 28,                                                     // 3x3 conv with fused GroupNorm-apply + SiLU of its input (conv_gn.h); GemmArgs::gn_scale callers only
    V_W128 = 29,                                                        // 256x256x64, four waves with 128x128 wave tiles (one per SIMD, AGPR accumulators; gemm_w128.h)
    V_W128_A = 30, V_W128_D = 31, V_W128_X = 32,                               // -DNATINF_DEV: other K-loop schedules of k_gemm_w128 (EPI 1 launches only)
    V_COUNT
};
const char* variant_name(int v) {
    static const char* n[] = {"auto", "generic128", "dma256x256", "dma256x128", "dma128x128", "ring256x256", "ring256x128",
                              "ring128x128", "ring64x128", "ring256x128w4", "dma256x128w4", "dma256x256s", "dma128x128s", "dma512x128", "patch256x256", "patch256x128", "dma256x256p", "dma128x128p", "dma256x128w4p", "gemm8ph", "gemm8ph_np", "gemm8ph_rf", "gemm8ph_nprf", "fp8_256x256", "abl_nodma", "abl_nomfma", "dma256x256h", "dma512x128h", "conv_gn", "w128_256x256", "w128_a", "w128_d", "w128_x"};
    return v >= 0 && v < V_COUNT ? n[v] : "?";
}
// The shipped library instantiates only the tile variants the dispatcher selects (choose_variant, splitk) plus the generic kernel; every other
// variant of the enum -- superseded pipelines kept for A/B runs -- exists in -DNATINF_DEV builds only, and natinf_set_gemm_variant /
// natinf_debug_gemm refuse it (NATINF_ESTATE) elsewhere.
inline bool variant_shipped(int v) {
#ifdef NATINF_DEV
    return v >= 0 && v < V_COUNT;
#else
    switch (v) {
        case V_AUTO: case V_GENERIC: case V_RING_64x128: case V_RING_256x128_W4: case V_DMA_128x128_P: case V_FP8_256x256:
        case V_DMA_256x256_H: case V_DMA_512x128_H: case V_CONV_GN: case V_W128: return true;
        default: return false;
    }
#endif
}
unsigned long long* g_dbg_ts = nullptr;
int g_force_variant = V_AUTO;
int g_half_issue = 1;              // natinf_set_gemm_half_issue(0): every wave issues its own LDS-DMA pieces (A/B runs)
int g_round_model = 1;             // natinf_set_gemm_round_model(0): small-M plain GEMMs by the pre-round-4 rules (A/B runs)
int g_round_model_w128 = 13;       // cost of a round of k_gemm_w128 tiles in tenths of a round of 128 x 128 tiles (two blocks per CU)
int g_pref_512 = 1;                // N <= 128 layers with >= 2 tiles per CU: the 512x128 hand-pipelined tile (natinf_set_gemm_pref512: A/B runs)      // tuning / tests: force one variant for every DMA-eligible launch
std::string* g_record = nullptr;   // when set, launch_gemm describes the launch instead of issuing it

int g_raster_g = 8;                 // natinf_set_gemm_raster: row-tiles per raster group of wide-N launches (0 / 1 = plain row-major)
template <class Cfg, class K>
inline void launch_tiles(K kernel, const GemmArgs& g0, hipStream_t s) {
    const int nM = (g0.M + Cfg::BM_ - 1) / Cfg::BM_, nN = (g0.N + Cfg::BN_ - 1) / Cfg::BN_;
    GemmArgs g = g0;
    g.raster_g = (g_raster_g > 1 && nN >= 8 && nM >= g_raster_g) ? g_raster_g : 0;
    int lds = Cfg::LDS_BYTES;
#ifdef NATINF_DEV
    static const int one_per_cu = getenv("NATINF_ONE_BLOCK_PER_CU") ? 1 : 0;      // occupancy experiment: ask for > 80 KB of LDS
    if (one_per_cu && lds < 84000) {
        lds = 84000;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
#endif
    hipLaunchKernelGGL(kernel, dim3(nM * nN, 1, g.batch), dim3(Cfg::THREADS), lds, s, g);
}
template <class Cfg, class K>
inline bool set_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES) == hipSuccess;
}
using CfgD256x256 = DmaCfg<2, 4, 8, 4>;  using CfgD256x128 = DmaCfg<4, 2, 4, 4>;  using CfgD128x128 = DmaCfg<2, 2, 4, 4>;
using CfgR256x256 = RingCfg<2, 4, 8, 4, 4>; using CfgR256x128 = RingCfg<4, 2, 4, 4, 6>;
using CfgR128x128 = RingCfg<2, 2, 4, 4, 4>; using CfgR64x128 = RingCfg<2, 2, 2, 4, 4>;
using CfgR256x128W4 = RingCfg<2, 2, 8, 4, 3>; using CfgD256x128W4 = DmaCfg<2, 2, 8, 4>;
using CfgD512x128 = DmaCfg<4, 2, 8, 4>;
using CfgH32 = ConvGn2Cfg<32>; using CfgH16 = ConvGn2Cfg<16>; using CfgH16W = ConvGn2Cfg<16, true>; using CfgH32W = ConvGn2Cfg<32, true>; using CfgH8T = ConvGn2Cfg<8, true, 4>; using CfgH4T = ConvGn2Cfg<4, true, 4, 2, 2>;
#ifdef NATINF_DEV
using CfgH8W = ConvGn2Cfg<8, true>;          // 8x8: two images per 128-pixel tile (superseded by the one-image tile: 0.6 % slower per forward)
using CfgP256x256 = PatchCfg<2, 4, 8, 4, 344>; using CfgP256x128 = PatchCfg<4, 2, 4, 4, 400>;
using CfgG32 = ConvGnCfg<32>; using CfgG16 = ConvGnCfg<16>; using CfgG16W = ConvGnCfg<16, true>;
#endif
int g_fuse_head = 1;               // natinf_set_fuse_head (read when a plan is BUILT): GroupNorm + SiLU + the 128 -> 3 output convolution as ONE launch (head_conv.h)
int g_cg8_tm4 = 1;                 // natinf_set_conv_gn8_tile: 1 = 8x8 level on 64-pixel x 256-channel tiles (one image per tile, two blocks per CU), 0 = 128 x 256 (two images per tile; -DNATINF_DEV builds)
constexpr int ATTN_BLK_DEFAULT = 2;
int g_attn_blk = ATTN_BLK_DEFAULT;  // natinf_set_attn_block (read when a plan is BUILT): the whole 16x16 attention block as ONE launch -- 1: k_qkv256 + k_attn256<true, 8> in one kernel (attn_blk256.h: q stays in
                                   // registers, k / V^T through L2); 2 (default since round 6): k_attn_blk256_v2 -- q k^T and P V against h itself, h resident in LDS (forward -3.3 % at B = 512); 0: two launches
int g_attn_qkv = 1;                // natinf_set_attn_qkv (read when a plan is BUILT): GroupNorm-apply + the q | k | v projections of the 16x16 attention as ONE launch (attn_qkv.h)
int g_attn_w8 = 1;                 // natinf_set_attn_waves8: k_attn256<true> as one 8-wave block per sample (1) or two 4-wave blocks (0)
int g_attn_proj = 1;               // natinf_set_attn_proj (read when a plan is BUILT): the 16x16 attention's output projection + skip + GroupNorm partials inside k_attn256
int g_attn256 = 1;                 // natinf_set_attn256: 1 = k_attn256 (K / V^T streamed through a two-stage LDS ring, two blocks per CU), 0 = k_attn_fused<8,16,true>
int g_fuse_gn8 = 1;                // natinf_set_fuse_gn8 (read when a plan is BUILT): the 8x8 level on the fused kernel too (two images per 128-pixel tile)
int g_cg_warm = 15;                // natinf_set_conv_gn_warm: bit mask by resolution (1: 4x4, 2: 8x8, 4: 16x16, 8: 32x32) of the fused-convolution launches that warm L2 with their weights
int g_fuse_fin = 3;                // natinf_set_fuse_fin (read when a plan is BUILT): at 8x8 / 4x4 the fused convolution's epilogue writes the GroupNorm table of its
                                   // output's consumer itself (whole samples x all channels per tile) instead of a k_gn_finalize launch behind it
int g_fuse_gn4 = 1;                // natinf_set_fuse_gn4 (read when a plan is BUILT): the 4x4 level on the fused kernel too (four images per 64-pixel tile) instead of
                                   // k_gn_apply + split-K GEMM + k_splitk_reduce + k_gn_stats
int g_fuse_gn = 1;                 // natinf_set_fuse_gn (read when a plan is BUILT): GroupNorm-apply + SiLU inside the consuming 3x3 conv

// Packed-epilogue specializations (EPI, gemm_dma.h) that exist per tile family, as bit masks: a launch whose epilogue is not
// instantiated for its tile takes the general fp32-slab epilogue (EPI 0: the same terms, the same single rounding).  Who needs what:
// 3 (SiLU) only the small time-embedding GEMMs; 4 (tanh-GELU), 7 (fp32 residual stream), 8 (row bias) the transformer engines, none
// of which ever reaches the N <= 128 tile.
constexpr unsigned EPI_ALL = 0x1FF;
constexpr unsigned EPI_R64 = EPI_ALL, EPI_D128 = EPI_ALL, EPI_RW4 = EPI_ALL & ~(1u << 3), EPI_D256H = EPI_ALL & ~(1u << 3),
                   EPI_D512H = EPI_ALL & ~((1u << 3) | (1u << 4) | (1u << 7) | (1u << 8)),
                   EPI_W128 = EPI_ALL & ~((1u << 2) | (1u << 3) | (1u << 6));          // plain long-K GEMMs: the transformer engines (GroupNorm partials take the general epilogue there)
template <unsigned MASK, int E, class F> inline void epi_case(F&& f) { if constexpr ((MASK >> E) & 1u) f(std::integral_constant<int, E>{}); }
template <unsigned MASK, class F> inline bool for_each_epi(F&& f) {             // f(integral_constant<int, E>) -> bool, over the instantiated ones
    bool ok = true;
    auto one = [&](auto tag) { ok = ok && f(tag); };
    epi_case<MASK, 0>(one); epi_case<MASK, 1>(one); epi_case<MASK, 2>(one); epi_case<MASK, 3>(one); epi_case<MASK, 4>(one);
    epi_case<MASK, 5>(one); epi_case<MASK, 6>(one); epi_case<MASK, 7>(one); epi_case<MASK, 8>(one);
    return ok;
}
#define NATINF_EPI_OF(tag) decltype(tag)::value
bool set_lds_epi_all() {
    return for_each_epi<EPI_D128>([](auto t) { return set_lds<CfgD128x128>(&k_gemm_dma<2, 2, 4, 4, 2, NATINF_EPI_OF(t)>); }) &&
           for_each_epi<EPI_RW4>([](auto t) { return set_lds<CfgR256x128W4>(&k_gemm_ring<2, 2, 8, 4, 3, NATINF_EPI_OF(t)>); }) &&
           for_each_epi<EPI_R64>([](auto t) { return set_lds<CfgR64x128>(&k_gemm_ring<2, 2, 2, 4, 4, NATINF_EPI_OF(t)>); }) &&
           for_each_epi<EPI_D256H>([](auto t) { return set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 6, NATINF_EPI_OF(t)>); }) &&
           for_each_epi<EPI_D512H>([](auto t) { return set_lds<CfgD512x128>(&k_gemm_dma<4, 2, 8, 4, 6, NATINF_EPI_OF(t)>); }) &&
           for_each_epi<EPI_W128>([](auto t) { return set_lds<W128Cfg>(&k_gemm_w128<NATINF_EPI_OF(t)>); }) && set_lds<W128Cfg>(&k_gemm_w128<9>)
#ifdef NATINF_DEV
           && set_lds<W128Cfg>(&k_gemm_w128<1, W128SchA>) && set_lds<W128Cfg>(&k_gemm_w128<1, W128SchP>) && set_lds<W128Cfg>(&k_gemm_w128<1, W128SchX>)
           && for_each_epi<EPI_ALL>([](auto t) { return set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 2, NATINF_EPI_OF(t)>); })
           && for_each_epi<EPI_ALL>([](auto t) { return set_lds<CfgD512x128>(&k_gemm_dma<4, 2, 8, 4, 2, NATINF_EPI_OF(t)>); })
#endif
        ;
}
template <int EPI>
bool set_lds_conv_gn() {
    return set_lds<CfgH32>(&k_conv_gn2<32, false, EPI>) && set_lds<CfgH32W>(&k_conv_gn2<32, true, EPI>) && set_lds<CfgH16>(&k_conv_gn2<16, false, EPI>) && set_lds<CfgH16W>(&k_conv_gn2<16, true, EPI>) &&
           set_lds<CfgH8T>(&k_conv_gn2<8, true, EPI, 4>) && set_lds<CfgH4T>(&k_conv_gn2<4, true, EPI, 4, 2, 2>)
#ifdef NATINF_DEV
           && set_lds<CfgH8W>(&k_conv_gn2<8, true, EPI>)
           && set_lds<CfgG32>(&k_conv_gn<32, false, EPI>) && set_lds<CfgG16>(&k_conv_gn<16, false, EPI>) && set_lds<CfgG16W>(&k_conv_gn<16, true, EPI>)
#endif
        ;
}

bool configure_gemm_kernels() {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  GEMM_LDS_BYTES) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_conv), hipFuncAttributeMaxDynamicSharedMemorySize, HeadConvCfg::LDS_BYTES) == hipSuccess &&
         set_lds<CfgR128x128>(&k_gemm_ring<2, 2, 4, 4, 4, 9>) &&
         set_lds_epi_all() && ncsn_cg3::configure() && ::configure_conv_ring() && configure_dit_attention() && set_lds_conv_gn<1>() && set_lds_conv_gn<2>() && set_lds_conv_gn<5>() && set_lds_conv_gn<6>() &&
#ifdef NATINF_DEV
         set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4>) && set_lds<CfgD256x128>(&k_gemm_dma<4, 2, 4, 4>) &&
         set_lds<CfgD128x128>(&k_gemm_dma<2, 2, 4, 4>) && set_lds<CfgR256x256>(&k_gemm_ring<2, 4, 8, 4, 4>) &&
         set_lds<CfgR256x128>(&k_gemm_ring<4, 2, 4, 4, 6>) && set_lds<CfgR128x128>(&k_gemm_ring<2, 2, 4, 4, 4>) &&
         set_lds<CfgD256x128W4>(&k_gemm_dma<2, 2, 8, 4>) && set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 1>) &&
         set_lds<CfgD128x128>(&k_gemm_dma<2, 2, 4, 4, 1>) && set_lds<CfgD512x128>(&k_gemm_dma<4, 2, 8, 4>) &&
         set_lds<CfgP256x256>(&k_conv_patch<2, 4, 8, 4, 344>) && set_lds<CfgP256x128>(&k_conv_patch<4, 2, 4, 4, 400>) &&
         set_lds<CfgD256x128W4>(&k_gemm_dma<2, 2, 8, 4, 2>) &&
         set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 3, 1>) && set_lds<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 4, 1>) &&
         set_lds<Cfg8ph>(&k_gemm_8ph<0>) && set_lds<Cfg8ph>(&k_gemm_8ph<1>) && set_lds<Cfg8ph>(&k_gemm_8ph<2>) && set_lds<Cfg8ph>(&k_gemm_8ph<3>) &&
#endif
         set_lds<CfgD256x256>(&k_gemm_fp8<false, 0>) && set_lds<CfgD256x256>(&k_gemm_fp8<true, 0>) &&
         set_lds<CfgD256x256>(&k_gemm_fp8<false, 1>) && set_lds<CfgD256x256>(&k_gemm_fp8<true, 1>) &&
         set_lds<CfgD256x256>(&k_gemm_fp8<false, 2>) && set_lds<CfgD256x256>(&k_gemm_fp8<true, 2>) &&
         set_lds<CfgD256x256>(&k_gemm_fp8<false, 3>) && set_lds<CfgD256x256>(&k_gemm_fp8<true, 3>) &&
         set_lds<W128F8Cfg>(&k_gemm_w128_fp8<false, 0>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<true, 0>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<false, 1>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<true, 1>) &&
         set_lds<W128F8Cfg>(&k_gemm_w128_fp8<false, 2>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<true, 2>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<false, 3>) && set_lds<W128F8Cfg>(&k_gemm_w128_fp8<true, 3>) &&
#ifdef NATINF_DEV
         set_lds<AttnCfg<8, 16, true>>(&k_attn_fused<8, 16, true>) &&      // the LDS-resident K / V^T form of the 16x16 attention: superseded by k_attn256
#endif
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_qkv256), hipFuncAttributeMaxDynamicSharedMemorySize, QKV_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn256<false>), hipFuncAttributeMaxDynamicSharedMemorySize, A256_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn256<true>), hipFuncAttributeMaxDynamicSharedMemorySize, A256_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn256<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, A256_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_blk256), hipFuncAttributeMaxDynamicSharedMemorySize, ABLK_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_blk256_v2), hipFuncAttributeMaxDynamicSharedMemorySize, ABLK2_LDS_BYTES) == hipSuccess &&
         // (k_attn_fused<2,4> / <3,5> / <3,6> with v as V^T: superseded by the row-major-v forms of the DiT engine, configure_dit_attention)
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn64), hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_BYTES) == hipSuccess &&
#ifdef NATINF_DEV
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn64_v2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_BYTES) == hipSuccess &&
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn64_v2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_BYTES) == hipSuccess &&
#endif
         hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn64_v2<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_LDS_BYTES / 2) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}

// Automatic choice: the largest block tile that still gives every CU a tile (DMA kernels need K a multiple of
// 64 per segment and zero-bordered 3x3 operands); the register-staged, fully masked kernel otherwise (4x4
// attention: K = 16).
// k_gemm_8ph has no edge clamping and addresses the 1x1 segment row-linearly
#ifdef NATINF_DEV
inline bool eligible_8ph(const GemmArgs& g) { return g.M % 256 == 0 && g.N % 256 == 0; }
#endif

int variant_bm(int v);
// k_conv_gn instantiations: 32x32 and 16x16 images, 256 x 128 tiles -- 128 x 256 tiles for 16x16 layers whose N is a multiple of 256
// (natinf_set_conv_gn_wide: A/B runs); packed epilogues 1 / 2 / 5 / 6 only
int packed_epi(const GemmArgs& g, int bm);
int g_cg3 = 7;                     // natinf_set_conv_gn_w128: bit 0 = 32x32 layers with N % 256 != 0 on 512 x 128 tiles, bit 1 = 32x32 layers with N % 256 == 0 on 256 x 256
                                   // tiles, bit 2 = 16x16 layers with N % 256 == 0 on 256 x 256 tiles (one image per tile)
int g_cg_wide = 3;                 // natinf_set_conv_gn_wide: bit 0 = 128 x 256 tiles at 16x16, bit 1 = at 32x32 (N % 256 == 0 layers: the 16 -> 32 up-sampling block)
int g_fuse_up = 1;                 // natinf_set_fuse_up (read when a plan is BUILT): up blocks at 16x16 / 32x32 fetch their input up-sampled inside k_conv_gn2
int g_cg_regw = 1;                 // natinf_set_conv_gn_regw: 1 = k_conv_gn2 (weights streamed through registers) where GemmArgs::b_frag is given
// tile rows of the fused-convolution instantiation a launch takes: 128 x 256 tiles at 16x16 (N % 256 == 0) and at 8x8 (two images per tile), 256 x 128 elsewhere
// the k_conv_gn3 shape a fused-convolution launch takes (-1: k_conv_gn2).  One block per CU exposes a tile's prologue and epilogue (~28k clocks), which the two
// co-resident blocks of k_conv_gn2 partly hide: k_conv_gn3 is ahead where the K loop is long (same-process A/B at B = 512, profiles/r05/cg3_v2_ab.log: 1.04-1.08 at
// K >= 2,304 on 512 x 128 tiles, 1.035-1.04 on 256 x 256 tiles at 32x32, 1.01-1.06 at K >= 2,816 at 16x16; inside the network, rocprofv3 trace of both in one process, profiles/r05/cg3_in_network_ab_by_shape.txt: 1.08-1.15 at K >= 2,304 at 32x32, 1.02-1.09 at K >= 2,304 at 16x16) and level or behind at short K (0.97-0.99 at K = 1,152 .. 1,536)
int g_cg3_min_k[3] = {2304, 0, 2304};      // natinf_set_conv_gn_w128_min_k: smallest K (9 cin + shortcut channels) per shape that takes k_conv_gn3
inline int conv_gn3_shape(const GemmArgs& g) {
    const int res = 1 << g.logW;
    if (!g_cg3 || !g_cg_regw || !g.b_frag || g.N % 128) return -1;
    int sh = -1;
    if (res == 32) sh = g.N % 256 ? ((g_cg3 & 1) ? 0 : -1) : ((g_cg3 & 2) ? 1 : -1);
    else if (res == 16) sh = (g.N % 256 == 0 && (g_cg3 & 4)) ? 2 : -1;
    if (sh >= 0 && 9 * g.a0_C + (g.a1 ? g.a1_C : 0) < g_cg3_min_k[sh]) sh = -1;
    return sh;
}
inline int conv_gn_bm(const GemmArgs& g) {
    const int res = 1 << g.logW;
    if (const int sh3 = conv_gn3_shape(g); sh3 >= 0) return ncsn_cg3::tile_rows(sh3);
    if (res == 8) return g_cg8_tm4 ? 64 : 128;
    if (res == 4) return 64;
    if (res == 32) return ((g_cg_wide & 2) && g.N % 256 == 0 && g_cg_regw && g.b_frag) ? 128 : 256;      // (k_conv_gn2 only)
    return ((g_cg_wide & 1) && res == 16 && g.N % 256 == 0) ? 128 : 256;
}
// rows of one GroupNorm-partial table row the launch writes (what the caller divides H*W by): a tile, or one SAMPLE of the two an 8x8 tile holds
inline int conv_gn_part_rows(const GemmArgs& g) { const int res = 1 << g.logW; return res <= 8 ? res * res : conv_gn_bm(g); }
inline bool conv_gn_regw(const GemmArgs& g) { return g_cg_regw && g.b_frag && (conv_gn3_shape(g) >= 0 || g.N % ((conv_gn_bm(g) <= 128 && (1 << g.logW) != 4) ? 256 : 128) == 0); }      // (4x4: 64 x 128 tiles)
#ifdef NATINF_DEV
constexpr bool HAVE_CONV_GN_V1 = true;              // k_conv_gn (weights through an LDS ring): superseded, development builds only
#else
constexpr bool HAVE_CONV_GN_V1 = false;
#endif
// k_conv_gn / k_conv_gn2 have packed epilogues only: the fp32-slab A/B knob (natinf_set_gemm_epilogue) does not apply to them.  Per-sample terms need
// one sample per tile -- or, at 8x8, per HALF tile (the kernel keeps both samples' row vectors and partials: NSAMP)
inline int conv_gn_epi(const GemmArgs& g) { GemmArgs t = g; t.epi_fp32_slab = 0; return packed_epi(t, conv_gn_part_rows(g)); }
inline bool conv_gn_ok(const GemmArgs& g) {
    if (!g.gn_scale || !g.gn_shift || !g.gn_folded || g.taps != 9 || g.batch != 1 || g.a0_C % BK || (g.a1 && g.a1_C % BK)) return false;
    const int res = 1 << g.logW;
    if (g.logHW != 2 * g.logW || (res != 32 && res != 16 && res != 8 && res != 4) || g.N % 8) return false;
    if (res == 4 && (g.a0_C % (64 * CfgH4T::NG) || (g.a1 && g.a1_C % (64 * CfgH4T::NG)))) return false;          // two K groups per block: an even number of half-chunks / shortcut tiles EACH
    if (res <= 8 ? (g.M % (res * res) || g.N % (res == 4 ? 128 : 256) || g.a0_up || g.a1_up || !conv_gn_regw(g) || (g.resid && g.rowvec)) : g.M % std::max(256, conv_gn_bm(g)) != 0) return false;      // 8x8: whole images, k_conv_gn2 only;
    // (its residual epilogues keep one set of column terms for both samples of a tile: no per-sample row vector there)
    if ((g.a0_up || g.a1_up || !HAVE_CONV_GN_V1) && !conv_gn_regw(g)) return false;          // up-sampled fetches: k_conv_gn2 only
    const int e = conv_gn_epi(g);
    return e == 1 || e == 2 || e == 5 || e == 6;
}
int g_w128 = 1;                     // natinf_set_gemm_w128(0): plain GEMMs on the two-waves-per-SIMD 256x256 tile as before round 4 (A/B runs)
// k_gemm_w128 (gemm_w128.h): plain GEMMs only, 32-bit lane offsets into the operands
bool w128_ok(const GemmArgs& g) {
    if (g.taps != 1 || g.a1 || g.gn_scale || g.deq_m || g.deq_n || g.splitk > 1) return false;
    if (g.a0_C % BK || g.a0_C < 2 * BK || g.N % 8 || g.M % 8) return false;
    return (int64_t)g.M * g.a0_ld * 2 < (int64_t)1 << 32 && (int64_t)g.N * g.b_ld * 2 < (int64_t)1 << 32;
}
int choose_variant(const GemmArgs& g) {
    if (g.gn_scale) return V_CONV_GN;               // the operand is raw: no other kernel can read it (launch_gemm checks conv_gn_ok)
    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    const bool dma = K0 % BK == 0 && K1 % BK == 0 && (g.taps == 1 || (g.a0_padded && g.a0_C % BK == 0));
    if (!dma) return V_GENERIC;
    if (!variant_shipped(g_force_variant)) { /* refused by natinf_set_gemm_variant; never reached */ }
#ifdef NATINF_DEV
    else if (g_force_variant == V_PATCH_256x256 || g_force_variant == V_PATCH_256x128) {
        const bool patch_ok = g.taps == 9 && g.batch == 1 && g.logW >= 3;
        if (patch_ok && (g_force_variant == V_PATCH_256x128 || g.logW >= 4)) return g_force_variant;
    } else if (g_force_variant >= V_8PH_256x256 && g_force_variant <= V_8PH_BOTH) {
        if (eligible_8ph(g)) return g_force_variant;
    }
#endif
    else if (g_force_variant == V_W128) {
        if (w128_ok(g) && (!g.gn_part || (1 << g.logHW) % 256 == 0)) return V_W128;
    }
#ifdef NATINF_DEV
    else if (g_force_variant >= V_W128_A && g_force_variant <= V_W128_X) {
        if (w128_ok(g) && packed_epi(g, 256) == 1) return g_force_variant;
    }
#endif
    else if (g_force_variant > V_GENERIC && g_force_variant != V_CONV_GN && g_force_variant != V_FP8_256x256) {
        // a forced tile must keep GroupNorm partial tiles inside one sample (e.g. 512-row tiles on the 16x16 level do not)
        if (!g.gn_part || (g.taps == 9 && (1 << g.logHW) % variant_bm(g_force_variant) == 0)) return g_force_variant;
    }
    // measured on the engine's layer shapes (tools/bench_gemm.py, profiles/r01): 256x256 two-stage for wide-N,
    // long-K layers; the 4-wave 256x128 ring (wave tile 128x64, 2 blocks/CU) for N = 128 and short-K layers;
    // 128x128 when 256-row tiles would leave CUs idle; 64x128 for the 4x4 level
    const int64_t mt256 = (g.M + 255) / 256, mt128 = (g.M + 127) / 128;
    const int64_t nt128 = (g.N + 127) / 128;
#ifdef NATINF_DEV
    const bool half = g_half_issue != 0;                                       // natinf_set_gemm_half_issue(0): the every-wave-issues pipelines (A/B runs)
#else
    constexpr bool half = true;
#endif
    const bool w128 = g_w128 && w128_ok(g) && (!g.gn_part || (1 << g.logHW) % 256 == 0);       // round 4: plain GEMMs on the one-wave-per-SIMD tile (gemm_w128.h)
    // Round 4: small-M plain GEMMs (the text stream of the MMDiT: M = 8 x 333 rows) by ROUNDS of blocks, not by "enough tiles for every CU": at
    // (2664, 6144, 1536) the rule below took 256 x 256 tiles -- 264 of them: a second round for eight tiles, 77 us -- where 1,008 tiles of 128 x 128 run as two rounds of
    // two blocks per CU in 55 us; at (2664, 4608, 1536) it took 128 x 128 (756 tiles, two rounds, 52 us) where 198 tiles of 256 x 256 are ONE round (43 us).  Measured
    // cost of a round at K = 1,536: 25-28 us (128 x 128, two blocks per CU) against 37-43 us (256 x 256): ratio 1.5 (tools/scan_small_m_gemm.py).
    // (round 5: the four-wave tile multiplies its 256 columns as two 128-column halves and skips a half that lies beyond N, so N % 128 == 0 is enough for it --
    // DiT-XL/2's q | k | v projection, N = 3,456 = 13.5 tiles, had fallen to 128 x 128 tiles: 50.7 us against 30 us)
    const int64_t nt256 = (g.N + 255) / 256;
    const bool n_ok256 = g.N % 256 == 0 || (w128 && g.N % 128 == 0);
    if (g_round_model && g.taps == 1 && !g.gn_part && g.batch == 1 && n_ok256 && K0 + K1 >= 1024 && mt256 * nt256 < 2 * NUM_CU) {
        const int64_t r256 = (mt256 * nt256 + NUM_CU - 1) / NUM_CU, r128 = (mt128 * nt128 + 2 * NUM_CU - 1) / (2 * NUM_CU);
        // (a round of the four-wave 256 x 256 tile costs ~1.3 rounds of 128 x 128 tiles, not 1.5: DiT-XL/2's fc1 at B = 16, (4096, 4608, 1152), is 288 tiles = two rounds of
        // ~28 us against three rounds of two 128 x 128 blocks per CU in 74.6 us; natinf_set_gemm_round_model(v >= 10) sets the ratio to v / 10 for A/B runs)
        const int64_t c256 = w128 ? g_round_model_w128 : 15;
        if (mt128 * nt128 >= NUM_CU / 2) return c256 * r256 < 10 * r128 ? (w128 ? V_W128 : half ? V_DMA_256x256_H : V_DMA_256x256_P) : V_DMA_128x128_P;
    }
    if (n_ok256 && K0 + K1 >= 1024 && mt256 * nt256 * g.batch >= NUM_CU) return w128 ? V_W128 : half ? V_DMA_256x256_H : V_DMA_256x256_P;
    if (g_pref_512 && g.N <= 128 && K0 + K1 >= 1024 && ((g.M + 511) / 512) * g.batch >= 2 * NUM_CU &&
        (!g.gn_part || (g.taps == 9 && (1 << g.logHW) % 512 == 0)))           // GroupNorm partials: a tile inside one sample
        return half ? V_DMA_512x128_H : V_DMA_512x128;
    if (mt256 * nt128 * g.batch >= 2 * NUM_CU) return V_RING_256x128_W4;        // it runs two blocks per CU
    if (mt128 * nt128 * g.batch >= NUM_CU) return V_DMA_128x128_P;
    return V_RING_64x128;
}

int variant_bm(int v) {
    switch (v) {
        case V_CONV_GN: case V_DMA_256x256: case V_DMA_256x128: case V_RING_256x256: case V_RING_256x128: case V_RING_256x128_W4: case V_DMA_256x128_W4:
        case V_ABL_NODMA: case V_ABL_NOMFMA: case V_DMA_256x256_H: case V_DMA_256x256_S: case V_PATCH_256x256: case V_PATCH_256x128: case V_DMA_256x256_P: case V_DMA_256x128W4_P: case V_8PH_256x256: case V_8PH_NOPRIO: case V_8PH_READFIRST: case V_8PH_BOTH: case V_FP8_256x256: case V_W128: case V_W128_A: case V_W128_D: case V_W128_X: return 256;
        case V_DMA_512x128: case V_DMA_512x128_H: return 512;
        case V_RING_64x128: return 64;
        default: return 128;
    }
}

// fp8 operands (a0 / b point at e4m3 bytes, a0_ld / b_ld / a_bs / b_bs in bytes, a0_C = K % 128 == 0, deq_m / deq_n set)
int fp8_epi(const GemmArgs& g) {
    if (g.epi_fp32_slab || g.resid || g.gn_part) return 0;
    if ((g.rowvec || g.gate) && g.log_rows_per_sample < 30 && ((1 << g.log_rows_per_sample) % 256 != 0)) return 0;
    if (g.c_mode == OUT_F32 && g.resid_f32 && !g.bias_m && g.act == ACT_NONE && g.resid_f32_ld % 4 == 0 && g.c_ld % 4 == 0) return 3;
    if (g.resid_f32 || g.gate) return 0;
    if (g.c_mode == OUT_BF16 && g.act == ACT_NONE) return 1;
    if (g.c_mode == OUT_FP8_MX && g.act == ACT_GELU_TANH && g.c_mx && g.N % 32 == 0 && !g.bias_m && g.scale == 1.0f) return 2;      // (its epilogue carries neither term)
    return 0;
}
extern int g_w128;
// k_gemm_w128_fp8 (gemm_w128.h): an even number of 128-byte K-tiles, 32-bit offsets into the operands
bool w128_fp8_ok(const GemmArgs& g) {
    // (E8M0 block scales of A arrive by DMA as whole 256-row groups per K-tile: the plane must hold them -- whole row tiles only)
    return g.taps == 1 && !g.a1 && g.a0_C % 256 == 0 && g.N % 8 == 0 && g.M % 8 == 0 && (!g.a_mx || g.M % 256 == 0) &&
           (int64_t)g.M * g.a0_ld < (int64_t)1 << 32 && (int64_t)g.N * g.b_ld < (int64_t)1 << 32;
}
bool fp8_on_w128(const GemmArgs& g) { return g_w128 && w128_fp8_ok(g) && (fp8_epi(g) != 2 || g_w128 != 2); }      // the four-wave tile takes this launch
template <bool MXA>
void launch_gemm_fp8_t(const GemmArgs& g, hipStream_t s) {
    // (round 5: the e4m3 + E8M0 epilogue with its tanh-GELU -- fc1 -- takes the four-wave tile too: with the GELU issued stage by stage for eight values at a time
    // (gelu_tanh_fast8) (32768, 6144, 1536) runs 1,756-1,759 TFLOP/s there against 1,641-1,697 on the eight-wave tile, same process (tools/ab_fc1_w128.py).  Round 4 kept it on
    // the eight-wave tile on a figure -- 1,100-1,130 against 1,300-1,520 -- that the debug entry had measured on the fp32-SLAB epilogue in e4m3 mode (no activation
    // passed: fp8_epi() = 0), not on this one.  natinf_set_gemm_w128(2) = the round-4 rule, for A/B runs.)
    if (fp8_on_w128(g)) {
        switch (fp8_epi(g)) {
            case 1: launch_tiles<W128F8Cfg>(&k_gemm_w128_fp8<MXA, 1>, g, s); break;
            case 2: launch_tiles<W128F8Cfg>(&k_gemm_w128_fp8<MXA, 2>, g, s); break;
            case 3: launch_tiles<W128F8Cfg>(&k_gemm_w128_fp8<MXA, 3>, g, s); break;
            default: launch_tiles<W128F8Cfg>(&k_gemm_w128_fp8<MXA, 0>, g, s); break;
        }
        return;
    }
    switch (fp8_epi(g)) {
        case 1: launch_tiles<CfgD256x256>(&k_gemm_fp8<MXA, 1>, g, s); break;
        case 2: launch_tiles<CfgD256x256>(&k_gemm_fp8<MXA, 2>, g, s); break;
        case 3: launch_tiles<CfgD256x256>(&k_gemm_fp8<MXA, 3>, g, s); break;
        default: launch_tiles<CfgD256x256>(&k_gemm_fp8<MXA, 0>, g, s); break;
    }
}
// natinf_gemm_profile(1): every matmul-shaped launch of every engine -- launch_gemm and launch_gemm_fp8, i.e. the kernel WITH the epilogue it runs in the network, on
// the stream it runs on, between its real neighbours -- is bracketed by a HIP event pair and tagged with the line natinf_ncsnpp_describe_gemms would print for it.
// natinf_gemm_profile_read sums them per tag.  (Round-5 review, item 2: the SD3 bench line quoted isolated loops of debug entries with the plain epilogue.)
// One host thread at a time, like natinf_attention_profile; the events serialise nothing, but two HIP streams still overlap: a launch's span then includes what it
// shared the chip with -- bench.py reads the image-stream shapes, whose launches are 10-100x the text stream's.
struct GemmProf {
    struct Rec { hipEvent_t a, b; std::string tag; };
    bool on = false; std::vector<Rec> ev; std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    bool begin(Rec& r, hipStream_t s) {
        r.a = r.b = nullptr;
        if (!pool.empty()) { r.a = pool.back().first; r.b = pool.back().second; pool.pop_back(); }
        else if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { r.a = r.b = nullptr; (void)hipGetLastError(); return false; }
        (void)hipEventRecord(r.a, s);
        return true;
    }
    void end(Rec& r, hipStream_t s) { (void)hipEventRecord(r.b, s); ev.push_back(std::move(r)); }
} g_gemm_prof;
void launch_gemm_fp8(const GemmArgs& g, hipStream_t s) {
    GemmProf::Rec r;
    const bool prof = g_gemm_prof.on && !g_record && g_gemm_prof.begin(r, s);
    if (prof) {
        char line[160];
        snprintf(line, sizeof(line), "%d %d %d %d %d %d %s%s/e%d", g.M, g.N, g.taps * g.a0_C, 0, g.taps, g.batch, fp8_on_w128(g) ? "w128_fp8" : "fp8_256x256", g.a_mx ? "_mxa" : "", fp8_epi(g));
        r.tag = line;
    }
    if (g.a_mx) launch_gemm_fp8_t<true>(g, s); else launch_gemm_fp8_t<false>(g, s);
    if (prof) g_gemm_prof.end(r, s);
}

// Which epilogue a launch can take (see tile_epilogue in gemm_dma.h): 0 = the general fp32-slab one; 1..6 = packed, when only
// column terms (bias, a per-sample row vector with every block tile inside one sample) and a bf16 residual are fused, the output is bf16, and
// GroupNorm partials (no activation, whole tiles) or an activation -- not both -- are asked for.
int packed_epi(const GemmArgs& g, int bm) {      // resid must be 8-byte aligned per 4-column group: ld % 4, base from the arena
    if (g.epi_fp32_slab || g.deq_m || g.deq_n || g.act == ACT_RELU) return 0;      // (ReLU: the general epilogue applies it)
    if (g.bias_m) return (g.c_mode == OUT_BF16 && !g.resid && !g.resid_f32 && !g.gate && !g.rowvec && !g.gn_part && g.act == ACT_NONE) ? 8 : 0;
    if ((g.rowvec || g.gate) && g.log_rows_per_sample < 30 && ((1 << g.log_rows_per_sample) % bm != 0)) return 0;    // per-sample terms: one sample per tile
    if (g.c_mode == OUT_F32 && g.resid_f32 && !g.resid && !g.gn_part && g.act == ACT_NONE && g.resid_f32_ld % 4 == 0 && g.c_ld % 4 == 0) return 7;
    if (g.c_mode != OUT_BF16 || g.resid_f32 || g.gate) return 0;
    if (g.resid && (g.act != ACT_NONE || g.resid_ld % 4 != 0)) return 0;
    if (g.gn_part) return (g.act == ACT_NONE && g.M % bm == 0) ? (g.resid ? 6 : 2) : 0;
    if (g.resid) return 5;
    return g.act == ACT_NONE ? 1 : (g.act == ACT_SILU ? 3 : 4);
}
inline unsigned epi_mask(int v) {
    switch (v) {
        case V_RING_64x128: return EPI_R64;  case V_DMA_128x128_P: return EPI_D128;  case V_RING_256x128_W4: return EPI_RW4;
        case V_DMA_256x256_H: return EPI_D256H;  case V_DMA_512x128_H: return EPI_D512H;
        case V_DMA_256x256_P: case V_DMA_512x128: return EPI_ALL;
        case V_W128: return EPI_W128;
        default: return 1u;
    }
}
// the epilogue specialization a launch on tile variant v runs (0 where its packed one is not instantiated for that tile)
inline int effective_epi(int v, const GemmArgs& g) { const int e = packed_epi(g, variant_bm(v)); return ((epi_mask(v) >> e) & 1u) ? e : 0; }
#define NATINF_LAUNCH_EPI(MASK, CFG, KERN, ...)                                                                      \
    {                                                                                                                \
        auto run_ = [&](auto t_) { launch_tiles<CFG>(&KERN<__VA_ARGS__, NATINF_EPI_OF(t_)>, g, s); };                \
        switch (effective_epi(v, g)) {                                                                               \
            case 1: epi_case<MASK, 1>(run_); break;  case 2: epi_case<MASK, 2>(run_); break;                         \
            case 3: epi_case<MASK, 3>(run_); break;  case 4: epi_case<MASK, 4>(run_); break;                         \
            case 5: epi_case<MASK, 5>(run_); break;  case 6: epi_case<MASK, 6>(run_); break;                         \
            case 7: epi_case<MASK, 7>(run_); break;  case 8: epi_case<MASK, 8>(run_); break;                         \
            default: run_(std::integral_constant<int, 0>{}); break;                                                  \
        }                                                                                                            \
    }

// set when a launch is asked for something no kernel provides (a plan-builder bug, or an A/B knob flipped after the plan was built);
// natinf_ncsnpp_forward clears it on entry and reports it on exit -- per calling thread, so two engines on two threads do not see
// each other's, and a description pass (g_record) never sets it
thread_local int g_launch_error = 0;
thread_local bool g_fin_written = false;      // set by launch_gemm: the launch that just ran wrote the consumer's GroupNorm table (GemmArgs::fin_*) -- k_conv_gn3 at 16x16
int g_splitk = 1;                  // natinf_set_gemm_splitk: 0 = never split K
float* g_dbg_splitk_ws = nullptr; int g_dbg_splitk_max = 0;        // natinf_debug_set_splitk_workspace
// Split-K for launches that cannot fill the chip otherwise (the 8x8 and 4x4 levels: 32,768 / 8,192 rows x 256 columns, K = 2,304 ..
// 4,608): 128 x 128 tiles (fill per flop of the large tile) x S slices of K >= 2 blocks per CU, then one reduce pass with the fused
// terms.  Returns the slice count (1 = do not split).
int splitk_slices(const GemmArgs& g) {
    if (!g_splitk || !g.splitk_ws || g.splitk_max < 2 || g.batch != 1 || g.gn_scale || g.c_mode != OUT_BF16 || g.N % 8 || 256 % (g.N / 8)) return 1;
    if (g.gn_part && (g.M % SPLITK_ROWS || g.act != ACT_NONE || (g.taps == 9 && (1 << g.logHW) % SPLITK_ROWS))) return 1;
    if (g.bias_m || g.gate || g.resid_f32 || g.deq_m || g.deq_n) return 1;
    const int K0 = g.taps * g.a0_C, K1 = g.a1 ? g.a1_C : 0;
    if (K0 % 64 || K1 % 64 || (g.taps == 9 && !g.a0_padded) || K0 + K1 < 2048) return 1;
    const int64_t tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (tiles >= 2 * NUM_CU) return 1;
    int S = (int)((2 * NUM_CU + tiles - 1) / tiles);
    if (S > g.splitk_max) S = g.splitk_max;
    while (S > 1 && (K0 + K1) / 32 / S < 16) --S;                       // at least 16 K-tiles per slice
    return S;
}
// Split-K on the four-wave tile for under-filled long-K GEMMs with the gated fp32 residual epilogue (gemm_w128.h: k_gemm_w128<9> + k_splitk_reduce_f32).  Returns the
// slice count (1 = do not split): the tiles of 256 x 256 fill less than half the chip, every slice keeps >= 16 K-tiles.
int w128_splitk_slices(const GemmArgs& g) {
    // (batch 1 only: the workspace contract is splitk_max * M * N floats; a batched launch would need batch times that)
    if (!g_splitk || !g_w128 || !g.splitk_ws || g.splitk_max < 2 || g.batch != 1 || !w128_ok(g) || g.a0_C < 3072) return 1;
    if (g.c_mode != OUT_F32 || !g.resid_f32 || g.resid || g.rowvec || g.bias_m || g.gn_part || g.act != ACT_NONE || g.epi_fp32_slab || g.N % 4 || g.c_ld % 4 || g.resid_f32_ld % 4) return 1;
    const int64_t tiles = (int64_t)((g.M + 255) / 256) * ((g.N + 255) / 256) * g.batch;
    if (tiles * 2 > NUM_CU) return 1;
    int S = (int)(NUM_CU / tiles);
    if (S > g.splitk_max) S = g.splitk_max;
    while (S > 1 && g.a0_C / BK / S < 16) --S;
    return S;
}
int launch_gemm_run(const GemmArgs& g0, hipStream_t s);
// returns the block-tile row count of the variant used
int launch_gemm(const GemmArgs& g0, hipStream_t s) {
    if (!g_gemm_prof.on || g_record) return launch_gemm_run(g0, s);
    GemmProf::Rec r;
    std::string tag;
    g_record = &tag; (void)launch_gemm_run(g0, s); g_record = nullptr;          // description pass: the tag, nothing launched
    while (!tag.empty() && tag.back() == '\n') tag.pop_back();
    if (!g_gemm_prof.begin(r, s)) return launch_gemm_run(g0, s);
    r.tag = std::move(tag);
    const int bm = launch_gemm_run(g0, s);
    g_gemm_prof.end(r, s);
    return bm;
}
int launch_gemm_run(const GemmArgs& g0, hipStream_t s) {
    if (g_force_variant == V_AUTO) {
        const int S8 = w128_splitk_slices(g0);
        if (S8 > 1) {
            if (g_record) {
                char line[160];
                snprintf(line, sizeof(line), "%d %d %d %d %d %d splitk%d_w128_256x256/e7\n", g0.M, g0.N, g0.taps * g0.a0_C, 0, g0.taps, g0.batch, S8);
                *g_record += line;
                return 256;
            }
            GemmArgs p = g0;
            p.splitk = S8; p.c = g0.splitk_ws; p.c_mode = OUT_F32;
            const int nM = (p.M + 255) / 256, nN = (p.N + 255) / 256;
            p.raster_g = 0;
            hipLaunchKernelGGL((k_gemm_w128<9>), dim3(nM * nN, S8, p.batch), dim3(256), W128Cfg::LDS_BYTES, s, p);
            const int64_t per = (int64_t)g0.M * (g0.N / 4);
            hipLaunchKernelGGL(k_splitk_reduce_f32, dim3((unsigned)((per + 255) / 256), (unsigned)g0.batch), dim3(256), 0, s, g0.splitk_ws, S8, (int64_t)g0.batch * g0.M * g0.N, g0.M, g0.N,
                               g0.bias_n, g0.gate, g0.gate_ld, g0.log_rows_per_sample, g0.z_samples, g0.resid_f32, g0.resid_f32_ld, g0.c_bs, g0.scale,
                               reinterpret_cast<float*>(g0.c), g0.c_ld, g0.stream_f16);
            return 256;
        }
    }
    const int S = g_force_variant == V_AUTO ? splitk_slices(g0) : 1;
    if (S > 1) {
        if (g_record) {
            char line[160];
            snprintf(line, sizeof(line), "%d %d %d %d %d %d splitk%d_ring128x128/e9\n", g0.M, g0.N, g0.taps * g0.a0_C, g0.a1 ? g0.a1_C : 0, g0.taps, g0.batch, S);
            *g_record += line;
            return SPLITK_ROWS;
        }
        GemmArgs p = g0;
        p.splitk = S; p.c = g0.splitk_ws; p.c_mode = OUT_F32; p.batch = S;
        p.bias_n = nullptr; p.rowvec = nullptr; p.resid = nullptr; p.scale = 1.0f; p.act = ACT_NONE; p.gn_part = nullptr;
        launch_tiles<CfgR128x128>(&k_gemm_ring<2, 2, 4, 4, 4, 9>, p, s);
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((g0.M + SPLITK_ROWS - 1) / SPLITK_ROWS)), dim3(256), 0, s, g0.splitk_ws, S, g0.M, g0.N,
                           g0.bias_n, g0.rowvec, g0.rowvec_ld, g0.log_rows_per_sample, reinterpret_cast<const bf16*>(g0.resid), g0.resid_ld, g0.scale,
                           g0.act, reinterpret_cast<bf16*>(g0.c), g0.c_ld, reinterpret_cast<float2*>(g0.gn_part), g0.gn_quads);
        return SPLITK_ROWS;
    }
    const GemmArgs& g = g0;
    const int v = choose_variant(g);
    if (v == V_CONV_GN && !conv_gn_ok(g)) {
        if (g_record) {                 // description pass: say so in the table instead of dropping the row
            char line[160];
            snprintf(line, sizeof(line), "%d %d %d %d %d %d invalid_conv_gn/e0\n", g.M, g.N, g.taps * g.a0_C, g.a1 ? g.a1_C : 0, g.taps, g.batch);
            *g_record += line;
        } else g_launch_error = 1;
        return 256;
    }
    if (g_record) {
        char line[160];
        snprintf(line, sizeof(line), "%d %d %d %d %d %d %s/e%d\n", g.M, g.N, g.taps * g.a0_C, g.a1 ? g.a1_C : 0, g.taps, g.batch,
                 v == V_CONV_GN && conv_gn3_shape(g) >= 0 ? "conv_gn3" : variant_name(v), v == V_CONV_GN ? conv_gn_epi(g) : effective_epi(v, g));
        *g_record += line;
        return v == V_CONV_GN ? conv_gn_part_rows(g) : variant_bm(v);
    }
    switch (v) {
        case V_GENERIC: {
            const int nM = (g.M + BM - 1) / BM, nN = (g.N + BN - 1) / BN;
            hipLaunchKernelGGL(k_gemm_bf16, dim3(nM * nN, 1, g.batch), dim3(256), GEMM_LDS_BYTES, s, g);
            break;
        }
        case V_RING_64x128: NATINF_LAUNCH_EPI(EPI_R64, CfgR64x128, k_gemm_ring, 2, 2, 2, 4, 4) break;
        case V_RING_256x128_W4: NATINF_LAUNCH_EPI(EPI_RW4, CfgR256x128W4, k_gemm_ring, 2, 2, 8, 4, 3) break;
        case V_DMA_128x128_P: NATINF_LAUNCH_EPI(EPI_D128, CfgD128x128, k_gemm_dma, 2, 2, 4, 4, 2) break;
        case V_FP8_256x256: launch_tiles<CfgD256x256>(&k_gemm_fp8<false, 0>, g, s); break;
        case V_DMA_256x256_H: NATINF_LAUNCH_EPI(EPI_D256H, CfgD256x256, k_gemm_dma, 2, 4, 8, 4, 6) break;
        case V_DMA_512x128_H: NATINF_LAUNCH_EPI(EPI_D512H, CfgD512x128, k_gemm_dma, 4, 2, 8, 4, 6) break;
#ifdef NATINF_DEV
        case V_W128_A: launch_tiles<W128Cfg>(&k_gemm_w128<1, W128SchA>, g, s); break;
        case V_W128_D: launch_tiles<W128Cfg>(&k_gemm_w128<1, W128SchP>, g, s); break;        // (variant 31: the shipped schedule + the L2 prefetch)
        case V_W128_X: launch_tiles<W128Cfg>(&k_gemm_w128<1, W128SchX>, g, s); break;
#endif
        case V_W128: {
            auto run_ = [&](auto t_) { launch_tiles<W128Cfg>(&k_gemm_w128<NATINF_EPI_OF(t_)>, g, s); };
            switch (effective_epi(v, g)) {
                case 1: epi_case<EPI_W128, 1>(run_); break;  case 4: epi_case<EPI_W128, 4>(run_); break;
                case 5: epi_case<EPI_W128, 5>(run_); break;  case 7: epi_case<EPI_W128, 7>(run_); break;
                case 8: epi_case<EPI_W128, 8>(run_); break;
                default: run_(std::integral_constant<int, 0>{}); break;
            }
            break;
        }
#ifdef NATINF_DEV
        case V_DMA_256x256: launch_tiles<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4>, g, s); break;
        case V_DMA_256x128: launch_tiles<CfgD256x128>(&k_gemm_dma<4, 2, 4, 4>, g, s); break;
        case V_DMA_128x128: launch_tiles<CfgD128x128>(&k_gemm_dma<2, 2, 4, 4>, g, s); break;
        case V_RING_256x256: launch_tiles<CfgR256x256>(&k_gemm_ring<2, 4, 8, 4, 4>, g, s); break;
        case V_RING_256x128: launch_tiles<CfgR256x128>(&k_gemm_ring<4, 2, 4, 4, 6>, g, s); break;
        case V_RING_128x128: launch_tiles<CfgR128x128>(&k_gemm_ring<2, 2, 4, 4, 4>, g, s); break;
        case V_DMA_256x128_W4: launch_tiles<CfgD256x128W4>(&k_gemm_dma<2, 2, 8, 4>, g, s); break;
        case V_DMA_256x256_S: launch_tiles<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 1>, g, s); break;
        case V_DMA_128x128_S: launch_tiles<CfgD128x128>(&k_gemm_dma<2, 2, 4, 4, 1>, g, s); break;
        case V_DMA_512x128: NATINF_LAUNCH_EPI(EPI_ALL, CfgD512x128, k_gemm_dma, 4, 2, 8, 4, 2) break;
        case V_PATCH_256x256: launch_tiles<CfgP256x256>(&k_conv_patch<2, 4, 8, 4, 344>, g, s); break;
        case V_PATCH_256x128: launch_tiles<CfgP256x128>(&k_conv_patch<4, 2, 4, 4, 400>, g, s); break;
        case V_DMA_256x256_P: NATINF_LAUNCH_EPI(EPI_ALL, CfgD256x256, k_gemm_dma, 2, 4, 8, 4, 2) break;
        case V_DMA_256x128W4_P: launch_tiles<CfgD256x128W4>(&k_gemm_dma<2, 2, 8, 4, 2>, g, s); break;
        case V_8PH_256x256: launch_tiles<Cfg8ph>(&k_gemm_8ph<0>, g, s); break;
        case V_8PH_NOPRIO: launch_tiles<Cfg8ph>(&k_gemm_8ph<1>, g, s); break;
        case V_8PH_READFIRST: launch_tiles<Cfg8ph>(&k_gemm_8ph<2>, g, s); break;
        case V_8PH_BOTH: launch_tiles<Cfg8ph>(&k_gemm_8ph<3>, g, s); break;
        case V_ABL_NODMA: launch_tiles<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 3, 1>, g, s); break;
        case V_ABL_NOMFMA: launch_tiles<CfgD256x256>(&k_gemm_dma<2, 4, 8, 4, 4, 1>, g, s); break;
#endif
        case V_CONV_GN: {
            GemmArgs gw = g0;
            gw.w_warm = (g_cg_warm >> (g0.logW - 2)) & 1;
            const GemmArgs& g = gw;
            const int e = conv_gn_epi(g);
            const int e4 = e == 1 ? 0 : (e == 2 ? 1 : (e == 5 ? 2 : 3));
#define NATINF_CG2_LAUNCH(CFG, RES, WIDE)                                                                   \
            switch (e4) {                                                                                       \
                case 0: launch_tiles<CFG>(&k_conv_gn2<RES, WIDE, 1>, g, s); break;                              \
                case 1: launch_tiles<CFG>(&k_conv_gn2<RES, WIDE, 2>, g, s); break;                              \
                case 2: launch_tiles<CFG>(&k_conv_gn2<RES, WIDE, 5>, g, s); break;                              \
                default: launch_tiles<CFG>(&k_conv_gn2<RES, WIDE, 6>, g, s); break;                             \
            }
            if (const int sh3 = conv_gn3_shape(g); sh3 >= 0) {
                ncsn_cg3::launch(&g, sh3, e, (void*)s);
                g_fin_written = sh3 == 2 && g.fin_scale && (e == 2 || e == 6) && g.N == 256;      // (k_conv_gn3<16, 2, 2, 2 | 6>: FIN16, conv_gn3.h)
                return conv_gn_part_rows(g);
            }
            if (conv_gn_regw(g)) {
                if ((1 << g.logW) == 8 && g_cg8_tm4) {
                    switch (e4) {
                        case 0: launch_tiles<CfgH8T>(&k_conv_gn2<8, true, 1, 4>, g, s); break;
                        case 1: launch_tiles<CfgH8T>(&k_conv_gn2<8, true, 2, 4>, g, s); break;
                        case 2: launch_tiles<CfgH8T>(&k_conv_gn2<8, true, 5, 4>, g, s); break;
                        default: launch_tiles<CfgH8T>(&k_conv_gn2<8, true, 6, 4>, g, s); break;
                    }
                }
#ifdef NATINF_DEV
                else if ((1 << g.logW) == 8) { NATINF_CG2_LAUNCH(CfgH8W, 8, true) }
#endif
                else if ((1 << g.logW) == 4) {
                    switch (e4) {
                        case 0: launch_tiles<CfgH4T>(&k_conv_gn2<4, true, 1, 4, 2, 2>, g, s); break;
                        case 1: launch_tiles<CfgH4T>(&k_conv_gn2<4, true, 2, 4, 2, 2>, g, s); break;
                        case 2: launch_tiles<CfgH4T>(&k_conv_gn2<4, true, 5, 4, 2, 2>, g, s); break;
                        default: launch_tiles<CfgH4T>(&k_conv_gn2<4, true, 6, 4, 2, 2>, g, s); break;
                    }
                }
                else if ((1 << g.logW) == 32 && conv_gn_bm(g) == 128) { NATINF_CG2_LAUNCH(CfgH32W, 32, true) }
                else if ((1 << g.logW) == 32) { NATINF_CG2_LAUNCH(CfgH32, 32, false) }
                else if (conv_gn_bm(g) == 128) { NATINF_CG2_LAUNCH(CfgH16W, 16, true) }
                else { NATINF_CG2_LAUNCH(CfgH16, 16, false) }
            }
#undef NATINF_CG2_LAUNCH
#ifdef NATINF_DEV
#define NATINF_CG_LAUNCH(CFG, RES, WIDE)                                                                    \
            switch (e4) {                                                                                       \
                case 0: launch_tiles<CFG>(&k_conv_gn<RES, WIDE, 1>, g, s); break;                               \
                case 1: launch_tiles<CFG>(&k_conv_gn<RES, WIDE, 2>, g, s); break;                               \
                case 2: launch_tiles<CFG>(&k_conv_gn<RES, WIDE, 5>, g, s); break;                               \
                default: launch_tiles<CFG>(&k_conv_gn<RES, WIDE, 6>, g, s); break;                              \
            }
            else if ((1 << g.logW) == 32) { NATINF_CG_LAUNCH(CfgG32, 32, false) }
            else if (conv_gn_bm(g) == 128) { NATINF_CG_LAUNCH(CfgG16W, 16, true) }
            else { NATINF_CG_LAUNCH(CfgG16, 16, false) }
#undef NATINF_CG_LAUNCH
#endif
            return conv_gn_part_rows(g);
        }
        default: break;
    }
    return variant_bm(v);
}
inline int grid1d(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = (n + block - 1) / block; return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------
// plan builder
// ------------------------------------------------------------------------------------------------
struct Builder {
    natinf_ncsnpp& E;
    Arena arena;
    int64_t wtop = 0;                    // packed-weight bump pointer (bytes)
    int64_t poff = 0;                    // running parameter offset (floats)
    // time-embedding projection bank
    int dense_total = 0; int64_t dense_w = 0, dense_b = 0, dense_out = 0;

    // NATINF_NCSNPP_DDPM: the `ddpm` network (ddpm.py:39-181; configs/vp/ddpm/cifar10_continuous.py) -- two ResnetBlockDDPM per level (no
    // 1/sqrt(2) rescale, NIN shortcut), AttnBlock without rescale, Downsample / Upsample as plain 3x3 convolutions -- on the same kernels
    const bool ddpm;
    const int NUM_RES;
    const float res_scale;               // 1/sqrt(2) (skip_rescale of the ++ blocks) or 1
    explicit Builder(natinf_ncsnpp& e) : E(e), ddpm((e.flags & NATINF_NCSNPP_DDPM) != 0), NUM_RES(ddpm ? 2 : 4), res_scale(ddpm ? 1.0f : INV_SQRT2) {
        arena.keep = (e.flags & NATINF_NCSNPP_KEEP_ACTIVATIONS) != 0;
    }

    void op(int cls, OpFn f) { E.ops.push_back(std::move(f)); E.op_cls.push_back(cls); }
    int64_t wres(int64_t bytes) { const int64_t o = wtop; wtop += align_up(bytes, 256); return o; }
    int64_t take(int64_t n) { const int64_t o = poff; poff += n; return o; }

    // ---- weight packing recipes -------------------------------------------------------------
    void pack_conv(int64_t src, int64_t dst, int N, int Cin, int taps, int dst_ld, int koff, int tapstride, float wmul = 1.0f) {
        const int chunked = taps == 9 && Cin % BK == 0;      // every 3x3 conv except the 3-channel stem
        E.packs.push_back([=](const PackCtx& p) {
            const int64_t n = (int64_t)N * Cin * taps;
            hipLaunchKernelGGL(k_pack_conv, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, p.stream, p.params + src,
                               reinterpret_cast<bf16*>(p.packed + dst), N, Cin, taps, dst_ld, koff, tapstride, chunked, wmul);
        });
    }
    void pack_frag(int64_t src_packed, int64_t dst, int N, int ld, int cin, int c1) {      // packed -> packed (conv_gn2.h)
        E.packs.push_back([=](const PackCtx& p) {
            const int64_t n = (int64_t)(N / 16) * (9 * (cin / 32) + c1 / 32) * 64;
            hipLaunchKernelGGL(k_pack_frag, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, p.stream, reinterpret_cast<const bf16*>(p.packed + src_packed),
                               reinterpret_cast<bf16*>(p.packed + dst), N, ld, cin, c1);
        });
    }
    void pack_transpose(int64_t src, int64_t dst, int K, int N, int dst_ld) {
        E.packs.push_back([=](const PackCtx& p) {
            hipLaunchKernelGGL(k_pack_transpose, dim3(grid1d((int64_t)K * N, 256, 1 << 30)), dim3(256), 0, p.stream,
                               p.params + src, reinterpret_cast<bf16*>(p.packed + dst), K, N, dst_ld);
        });
    }
    void pack_transpose_at(int64_t src, int64_t dst_bytes, int K, int N, int dst_ld) { pack_transpose(src, dst_bytes, K, N, dst_ld); }
    void pack_zero(int64_t dst, int64_t n) {
        E.packs.push_back([=](const PackCtx& p) {
            hipLaunchKernelGGL(k_fill_bf16_zero, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, p.stream,
                               reinterpret_cast<bf16*>(p.packed + dst), n);
        });
    }
    int64_t pack_f32(int64_t src, int n, int64_t src2 = -1) {          // returns packed offset of an fp32 vector
        const int64_t dst = wres((int64_t)n * 4);
        pack_f32_at(src, n, dst, src2);
        return dst;
    }
    void pack_f32_at(int64_t src, int n, int64_t dst, int64_t src2 = -1) {
        E.packs.push_back([=](const PackCtx& p) {
            hipLaunchKernelGGL(k_copy_add_f32, dim3(grid1d(n)), dim3(256), 0, p.stream, p.params + src,
                               src2 >= 0 ? p.params + src2 : nullptr, reinterpret_cast<float*>(p.packed + dst), n);
        });
    }

    // ---- op emitters ------------------------------------------------------------------------
    struct GN { int64_t gamma, beta; };
    GN take_gn(int C) { GN g; g.gamma = pack_f32(take(C), C); g.beta = pack_f32(take(C), C); return g; }

    // GroupNorm statistics of x -> (scale, shift) per (image, channel); returns their arena offsets
    // (sc, sh: in / out -- where the producer's epilogue writes the table itself (Part::fin) they are REPLACED by its table)
    void emit_gn_stats(const TRef& x, GN gn, int64_t& sc, int64_t& sh, float out_mul = 1.0f) {
        if (emit_gn_from_parts(x, gn, sc, sh, out_mul)) return;
        const int HW = x.res * x.res;
        op(CLS_OTHER, [=](const Ctx& c) {
            hipLaunchKernelGGL(k_gn_stats, dim3(c.B), dim3(256), 0, c.stream, c.act(x), x.ld, x.C, HW,
                               c.w<float>(gn.gamma), c.w<float>(gn.beta), c.at<float>(sc), c.at<float>(sh), GN_EPS, out_mul);
        });
    }
    void emit_gn_apply(const TRef& x, int64_t sc, int64_t sh, const TRef& y, const TRef* xr, int act, int mode) {
        const int logW = ilog2(x.res), logHW = 2 * logW;
        const int rd = mode == RS_UP ? 2 * x.res : (mode == RS_DOWN ? x.res / 2 : x.res);
        const int rows_per_img = rd + 2 * y.pad;
        const TRef xrr = xr ? *xr : TRef();
        op(CLS_OTHER, [=](const Ctx& c) {
            hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)((rows_per_img + GN_ROWS - 1) / GN_ROWS), (unsigned)c.B), dim3(256), 0, c.stream, c.act(x), x.ld, x.C,
                               logW, logHW, c.at<float>(sc), c.at<float>(sh), c.act(y),
                               xrr.off >= 0 ? c.act(xrr) : (bf16*)nullptr, act, mode, y.pad);
        });
    }
    TRef new_act(int res, int C, int pad = 0) {
        TRef t; t.off = arena.alloc((int64_t)(res + 2 * pad) * (res + 2 * pad) * C * 2); t.C = C; t.ld = C; t.res = res; t.pad = pad;
        return t;
    }

    void emit_res(const Mod& m, const TRef& x, const TRef& out) {
        const int cin = m.cin, cout = m.cout;
        const int ro = m.up ? m.res * 2 : (m.down ? m.res / 2 : m.res);
        const bool shortcut = cin != cout || m.up || m.down;
        // parameters in registration order (layerspp.py:204-228)
        const GN gn0 = take_gn(cin);
        const int64_t p_c0w = take((int64_t)cout * cin * 9), p_c0b = take(cout);
        const int64_t p_dw = take((int64_t)cout * TEMB), p_db = take(cout);
        const GN gn1 = take_gn(cout);
        const int64_t p_c1w = take((int64_t)cout * cout * 9), p_c1b = take(cout);
        const int64_t p_c2w = shortcut ? take((int64_t)cout * cin) : -1, p_c2b = shortcut ? take(cout) : -1;

        const int K0a = 9 * cin, K1tot = 9 * cout + (shortcut ? cin : 0);
        // GroupNorm-apply + SiLU inside the consuming convolution (conv_gn.h) where an instantiation exists: output resolution
        // 32x32 or 16x16.  Conv_0 of a resampling block reads a resampled tensor and keeps the k_gn_apply pass (which also
        // produces the resampled shortcut input), Conv_1 is fused there too; the 8x8 / 4x4 levels are unfused.  Folded form: the
        // GroupNorm scale / shift carry -log2(e), the 3x3 weights -ln 2 (GemmArgs::gn_folded).
        const bool fusable_res = g_fuse_gn && (ro == 32 || ro == 16 || (ro == 8 && g_fuse_gn8 && cout % 256 == 0) || (ro == 4 && g_fuse_gn4 && cout % 256 == 0));
        const bool fuse1 = fusable_res && cout % BK == 0;                               // Conv_1
        const bool fuse_up = fuse1 && ro > 8 && g_fuse_up && m.up && cin % BK == 0 && cout % 128 == 0;   // up block: the 2x up-sampling of both branches happens in the fetches
        const bool fuse = (fusable_res && !m.up && !m.down && cin % BK == 0) || fuse_up;            // Conv_0
        const float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
        const float gn_mul = fuse ? -LOG2E : 1.0f, w_mul = fuse ? -LN2 : 1.0f, gn_mul1 = fuse1 ? -LOG2E : 1.0f, w_mul1 = fuse1 ? -LN2 : 1.0f;
        const int64_t w0 = wres((int64_t)cout * K0a * 2), w1 = wres((int64_t)cout * K1tot * 2);
        pack_conv(p_c0w, w0, cout, cin, 9, K0a, 0, cin, w_mul);
        pack_conv(p_c1w, w1, cout, cout, 9, K1tot, 0, cout, w_mul1);
        if (shortcut && !ddpm) pack_conv(p_c2w, w1, cout, cin, 1, K1tot, 9 * cout, cin);
        if (shortcut && ddpm) pack_transpose_at(p_c2w, w1 + (int64_t)9 * cout * 2, cin, cout, K1tot);      // NIN_0.W is [in][out] (layers.py:546-555)
        // k_conv_gn2 reads the weights fragment-major (after the packs above: the list runs in order)
        const int64_t w0f = (fuse && cout % 16 == 0) ? wres((int64_t)cout * K0a * 2) : -1, w1f = (fuse1 && cout % 16 == 0) ? wres((int64_t)cout * K1tot * 2) : -1;
        if (w0f >= 0) pack_frag(w0, w0f, cout, K0a, cin, 0);
        if (w1f >= 0) pack_frag(w1, w1f, cout, K1tot, cout, shortcut ? cin : 0);
        const int64_t b0 = pack_f32(p_c0b, cout), b1 = pack_f32(p_c1b, cout, shortcut ? p_c2b : -1);
        // time-embedding projection rows of this block inside the shared bank
        const int drow = dense_rows_next;
        pack_conv(p_dw, dense_w + (int64_t)drow * TEMB * 2, cout, TEMB, 1, TEMB, 0, TEMB);
        pack_f32_at(p_db, cout, dense_b + (int64_t)drow * 4);
        dense_rows_next += cout;

        const int64_t own_sc = arena.alloc((int64_t)std::max(cin, cout) * 4), own_sh = arena.alloc((int64_t)std::max(cin, cout) * 4);
        int64_t sc = own_sc, sh = own_sh;                 // GroupNorm_0's table: this block's buffers, or the one x's producer wrote
        emit_gn_stats(x, gn0, sc, sh, gn_mul);
        TRef h, xr;
        if (!fuse) {
            h = new_act(ro, cin, 1);
            if (m.up || m.down) xr = new_act(ro, cin);
            emit_gn_apply(x, sc, sh, h, (m.up || m.down) ? &xr : nullptr, ACT_SILU, m.up ? RS_UP : (m.down ? RS_DOWN : RS_NONE));
        }

        // split-K workspace for the low-resolution levels (launch_gemm decides per launch): 4 slices of fp32 partial sums
        const int SK_MAX = 4;
        const int64_t skws = (ro <= 8) ? arena.alloc((int64_t)SK_MAX * ro * ro * cout * 4) : -1;
        TRef t = new_act(ro, cout);
        const Part pt = register_output(t, fuse);
        const int logW = ilog2(ro), logHW = 2 * logW, HWo = ro * ro;
        const int dtotal = dense_total; const int64_t dout = dense_out;
        op(fuse ? (ro <= 8 ? CLS_CONV_GN8 : CLS_CONV_GN) : CLS_GEMM, [=](const Ctx& c) {
            GemmArgs g = gemm_defaults();
            if (fuse) { g.a0 = c.act(x); g.a0_ld = x.ld; g.gn_scale = c.at<float>(sc); g.gn_shift = c.at<float>(sh); g.gn_ld = cin; g.gn_folded = 1; g.a0_up = fuse_up; }
            else { g.a0 = c.act(h); g.a0_ld = h.ld; g.a0_padded = 1; }
            g.a0_C = cin; g.taps = 9; g.logW = logW; g.logHW = logHW;
            g.M = c.B * HWo; g.N = cout; g.b = c.w<bf16>(w0); g.b_ld = K0a;
            if (w0f >= 0) g.b_frag = c.w<bf16>(w0f);
            g.bias_n = c.w<float>(b0);
            g.rowvec = c.at<float>(dout) + drow; g.rowvec_ld = dtotal; g.log_rows_per_sample = logHW;
            if (skws >= 0) { g.splitk_ws = c.at<float>(skws); g.splitk_max = SK_MAX; }
            g.c = c.act(t); g.c_ld = t.ld;
            if (pt.valid) { g.gn_part = c.at<float>(pt.off); g.gn_quads = pt.quads; set_fin(g, pt, c); }
            g_fin_written = false;
            const int bm = launch_gemm(g, c.stream);
            if (pt.valid) c.part_bm[pt.id] = bm;
            if (pt.fin && pt.fin->check) c.fin_done[pt.id] = g_fin_written;
        });
        if (!fuse) arena.release(h.off);
        int64_t sc1 = own_sc, sh1 = own_sh;               // GroupNorm_1's table: the same buffers again, or the one Conv_0's epilogue wrote
        emit_gn_stats(t, gn1, sc1, sh1, gn_mul1);
        TRef u;
        if (!fuse1) {
            u = new_act(ro, cout, 1);
            emit_gn_apply(t, sc1, sh1, u, nullptr, ACT_SILU, RS_NONE);
            arena.release(t.off);
        }
        if (pt.valid) arena.release(pt.off);
        const TRef xs = ((m.up || m.down) && !fuse_up) ? xr : x;           // shortcut source at the output resolution (fuse_up: x itself, fetched up-sampled)
        const float rs = res_scale;
        const Part po = register_output(out, fuse1);
        op(fuse1 ? (ro <= 8 ? CLS_CONV_GN8 : CLS_CONV_GN) : CLS_GEMM, [=](const Ctx& c) {
            GemmArgs g = gemm_defaults();
            if (fuse1) { g.a0 = c.act(t); g.a0_ld = t.ld; g.gn_scale = c.at<float>(sc1); g.gn_shift = c.at<float>(sh1); g.gn_ld = cout; g.gn_folded = 1; }
            else { g.a0 = c.act(u); g.a0_ld = u.ld; g.a0_padded = 1; }
            g.a0_C = cout; g.taps = 9; g.logW = logW; g.logHW = logHW;
            if (shortcut) { g.a1 = c.act(xs); g.a1_ld = xs.ld; g.a1_C = cin; g.a1_up = fuse_up; }
            else { g.resid = c.act(xs); g.resid_ld = xs.ld; }
            g.M = c.B * HWo; g.N = cout; g.b = c.w<bf16>(w1); g.b_ld = K1tot;
            if (w1f >= 0) g.b_frag = c.w<bf16>(w1f);
            g.bias_n = c.w<float>(b1); g.scale = rs;
            if (skws >= 0) { g.splitk_ws = c.at<float>(skws); g.splitk_max = SK_MAX; }
            g.c = c.act(out); g.c_ld = out.ld;
            if (po.valid) { g.gn_part = c.at<float>(po.off); g.gn_quads = po.quads; set_fin(g, po, c); }
            g_fin_written = false;
            const int bm = launch_gemm(g, c.stream);
            if (po.valid) c.part_bm[po.id] = bm;
            if (po.fin && po.fin->check) c.fin_done[po.id] = g_fin_written;
        });
        if (skws >= 0) arena.release(skws);
        if (fuse1) arena.release(t.off); else arena.release(u.off);
        if ((m.up || m.down) && !fuse_up) arena.release(xr.off);
        arena.release(own_sc); arena.release(own_sh);
        if (sc != own_sc) { arena.release(sc); arena.release(sh); }          // tables written by producers' epilogues: consumed
        if (sc1 != own_sc) { arena.release(sc1); arena.release(sh1); }
        E.taps[m.idx] = out;
    }

    void emit_attn(const Mod& m, const TRef& x, const TRef& out) {
        const int C = m.cin, T = m.res * m.res;
        const GN gn = take_gn(C);
        int64_t pw[4], pb[4];
        for (int i = 0; i < 4; ++i) { pw[i] = take((int64_t)C * C); pb[i] = take(C); }
        const int64_t wqk = wres((int64_t)2 * C * C * 2), wv = wres((int64_t)C * C * 2), w3 = wres((int64_t)C * C * 2);
        pack_transpose(pw[0], wqk, C, C, C);
        pack_transpose(pw[1], wqk + (int64_t)C * C * 2, C, C, C);
        pack_transpose(pw[2], wv, C, C, C);
        pack_transpose(pw[3], w3, C, C, C);
        const int64_t bqk = wres((int64_t)2 * C * 4);
        pack_f32_at(pb[0], C, bqk); pack_f32_at(pb[1], C, bqk + (int64_t)C * 4);
        const int64_t bv = pack_f32(pb[2], C), b3 = pack_f32(pb[3], C);

        const int64_t own_sc = arena.alloc((int64_t)C * 4), own_sh = arena.alloc((int64_t)C * 4);
        int64_t sc = own_sc, sh = own_sh;
        emit_gn_stats(x, gn, sc, sh);
        const bool fuse_qkv = T == 256 && C == 256 && g_attn_qkv;      // k_qkv256: x -> [q | k], V^T in one launch; h never exists
        TRef h = new_act(m.res, C);
        if (!fuse_qkv) emit_gn_apply(x, sc, sh, h, nullptr, ACT_NONE, RS_NONE);
        const int64_t qk = arena.alloc((int64_t)T * 2 * C * 2), vT = arena.alloc((int64_t)C * T * 2);
        // k_attn_blk256: projections, attention and output projection of a sample in ONE launch (needs the three fusions it is made of)
        const bool blk = fuse_qkv && g_attn256 && g_attn_proj && g_attn_w8 && g_attn_blk;
        int64_t wqkvf = -1;
        const int64_t sc_q = sc, sh_q = sh;
        if (fuse_qkv) {
            wqkvf = wres((int64_t)3 * C * C * 2);
            const int64_t s0 = pw[0], s1 = pw[1], s2 = pw[2];
            E.packs.push_back([=](const PackCtx& p) {
                hipLaunchKernelGGL(k_pack_qkv_w, dim3(3 * 256), dim3(256), 0, p.stream, p.params + s0, p.params + s1, p.params + s2, reinterpret_cast<bf16*>(p.packed + wqkvf));
            });
            const int64_t sc_ = sc, sh_ = sh;
            if (!blk)
            op(CLS_GEMM, [=](const Ctx& c) {
                if (g_record) return;
                hipLaunchKernelGGL(k_qkv256, dim3((unsigned)c.B), dim3(512), QKV_LDS_BYTES, c.stream, (const bf16*)c.act(x), x.ld, c.at<float>(sc_), c.at<float>(sh_),
                                   c.w<bf16>(wqkvf), c.w<float>(bqk), c.w<float>(bv), c.at<bf16>(qk), c.at<bf16>(vT));
            });
        }
        if (!fuse_qkv)
        op(CLS_GEMM, [=](const Ctx& c) {             // q | k  = h Wq | h Wk
            GemmArgs g = gemm_defaults();
            g.a0 = c.act(h); g.a0_ld = C; g.a0_C = C; g.M = c.B * T; g.N = 2 * C;
            g.b = c.w<bf16>(wqk); g.b_ld = C; g.bias_n = c.w<float>(bqk);
            g.c = c.at<bf16>(qk); g.c_ld = 2 * C;
            launch_gemm(g, c.stream);
        });
        if (!fuse_qkv)
        op(CLS_GEMM, [=](const Ctx& c) {             // V^T[b] = Wv^T h[b]^T  (so that P V is an "A B^T" product)
            GemmArgs g = gemm_defaults();
            g.a0 = c.w<bf16>(wv); g.a0_ld = C; g.a0_C = C; g.a_bs = 0; g.M = C; g.N = T;
            g.b = c.act(h); g.b_ld = C; g.b_bs = (int64_t)T * C; g.bias_m = c.w<float>(bv);
            g.c = c.at<bf16>(vT); g.c_ld = T; g.c_bs = (int64_t)C * T; g.batch = c.B;
            launch_gemm(g, c.stream);
        });
        arena.release(h.off);
        // k_attn256<true>: the output projection, skip connection and GroupNorm partials in the attention launch (attn256.h); O never exists
        const bool proj = T == 256 && C == 256 && g_attn256 && g_attn_proj;
        int64_t w3f = -1;
        if (proj) {
            w3f = wres((int64_t)C * C * 2);
            const int64_t src = pw[3];
            E.packs.push_back([=](const PackCtx& p) {
                hipLaunchKernelGGL(k_pack_attn_w3, dim3(256), dim3(256), 0, p.stream, p.params + src, reinterpret_cast<bf16*>(p.packed + w3f));
            });
        }
        // k_attn_blk256_v2 (natinf_set_attn_block(2)): q k^T and P V against h itself -- the folded matrices Wqk = Wq Wk^T, Wvo = Wv W3 and their bias vectors, computed in
        // fp32 at load time (k_attn_fold_w) and packed like the q tiles / W3 of the four-projection plans
        const bool blk2 = blk && g_attn_blk == 2;
        int64_t wqf2 = -1, wvof2 = -1, cq2 = -1, bo2 = -1;
        if (blk2) {
            const int64_t fq = wres((int64_t)C * C * 4), fv = wres((int64_t)C * C * 4);
            cq2 = wres((int64_t)C * 4); bo2 = wres((int64_t)C * 4);
            wqf2 = wres((int64_t)C * C * 2); wvof2 = wres((int64_t)C * C * 2);
            const int64_t w0_ = pw[0], w1_ = pw[1], w2_ = pw[2], w3_ = pw[3], b0_ = pb[0], b2_ = pb[2], b3_ = pb[3], cq_ = cq2, bo_ = bo2, wq_ = wqf2, wv_ = wvof2;
            E.packs.push_back([=](const PackCtx& p) {
                float* fqp = reinterpret_cast<float*>(p.packed + fq); float* fvp = reinterpret_cast<float*>(p.packed + fv);
                hipLaunchKernelGGL(k_attn_fold_w, dim3(257), dim3(256), 0, p.stream, p.params + w0_, p.params + w1_, p.params + w2_, p.params + w3_, p.params + b0_, p.params + b2_,
                                   p.params + b3_, fqp, reinterpret_cast<float*>(p.packed + cq_), fvp, reinterpret_cast<float*>(p.packed + bo_));
                hipLaunchKernelGGL(k_pack_qkv_w, dim3(256), dim3(256), 0, p.stream, (const float*)fqp, (const float*)fqp, (const float*)fqp, reinterpret_cast<bf16*>(p.packed + wq_), 1);
                hipLaunchKernelGGL(k_pack_attn_w3, dim3(256), dim3(256), 0, p.stream, (const float*)fvp, reinterpret_cast<bf16*>(p.packed + wv_));
            });
        }
        const Part po_attn = proj ? register_output(out) : Part();
        const float rs_attn = res_scale;
        TRef O = new_act(m.res, C);
        if (T == 256 && C == 256) {
            // 16x16 attention: scores, softmax and P V of a sample in one block (attn_fused.h, two-phase: V^T follows K through LDS)
            op(CLS_GEMM, [=](const Ctx& c) {
                if (g_record) return;                    // natinf_ncsnpp_describe_gemms: GEMM launches only, nothing touches memory
                if (blk2) {
                    hipLaunchKernelGGL(k_attn_blk256_v2, dim3((unsigned)c.B), dim3(512), ABLK2_LDS_BYTES, c.stream, (const bf16*)c.act(x), x.ld, c.at<float>(sc_q), c.at<float>(sh_q),
                                       c.w<bf16>(wqf2), c.w<float>(cq2), 1.0f / sqrtf((float)C), c.w<bf16>(wvof2), c.w<float>(bo2),
                                       c.act(out), out.ld, rs_attn, po_attn.valid ? c.at<float2>(po_attn.off) : (float2*)nullptr, po_attn.quads);
                    if (po_attn.valid) c.part_bm[po_attn.id] = 256;
                } else if (blk) {
                    hipLaunchKernelGGL(k_attn_blk256, dim3((unsigned)c.B), dim3(512), ABLK_LDS_BYTES, c.stream, (const bf16*)c.act(x), x.ld, c.at<float>(sc_q), c.at<float>(sh_q),
                                       c.w<bf16>(wqkvf), c.w<float>(bqk), c.w<float>(bv), c.at<bf16>(qk), c.at<bf16>(vT), 1.0f / sqrtf((float)C), c.w<bf16>(w3f), c.w<float>(b3),
                                       c.act(out), out.ld, rs_attn, po_attn.valid ? c.at<float2>(po_attn.off) : (float2*)nullptr, po_attn.quads);
                    if (po_attn.valid) c.part_bm[po_attn.id] = 256;
                } else if (proj) {
                    if (g_attn_w8) {
                        hipLaunchKernelGGL((k_attn256<true, 8>), dim3((unsigned)c.B), dim3(512), A256_LDS_BYTES, c.stream, c.at<bf16>(qk), 2 * C, C, c.at<bf16>(vT), c.act(out), out.ld,
                                           1.0f / sqrtf((float)C), c.w<bf16>(w3f), c.w<float>(b3), (const bf16*)c.act(x), x.ld, rs_attn,
                                           po_attn.valid ? c.at<float2>(po_attn.off) : (float2*)nullptr, po_attn.quads);
                        if (po_attn.valid) c.part_bm[po_attn.id] = 256;
                    } else {
                    hipLaunchKernelGGL(k_attn256<true>, dim3((unsigned)(2 * c.B)), dim3(256), A256_LDS_BYTES, c.stream, c.at<bf16>(qk), 2 * C, C, c.at<bf16>(vT), c.act(out), out.ld,
                                       1.0f / sqrtf((float)C), c.w<bf16>(w3f), c.w<float>(b3), (const bf16*)c.act(x), x.ld, rs_attn,
                                       po_attn.valid ? c.at<float2>(po_attn.off) : (float2*)nullptr, po_attn.quads);
                    if (po_attn.valid) c.part_bm[po_attn.id] = 128;
                    }
                }
                else if (g_attn256)
                    hipLaunchKernelGGL(k_attn256<false>, dim3((unsigned)(2 * c.B)), dim3(256), A256_LDS_BYTES, c.stream, c.at<bf16>(qk), 2 * C, C, c.at<bf16>(vT), c.act(O), C,
                                       1.0f / sqrtf((float)C), (const bf16*)nullptr, (const float*)nullptr, (const bf16*)nullptr, 0, 1.0f, (float2*)nullptr, 0);
#ifdef NATINF_DEV
                else {
                    using Cfg = AttnCfg<8, 16, true>;
                    hipLaunchKernelGGL((&k_attn_fused<8, 16, true>), dim3((unsigned)c.B), dim3(Cfg::THREADS), Cfg::LDS_BYTES, c.stream, c.at<bf16>(qk), 2 * C, C,
                                       c.at<bf16>(vT), c.act(O), C, 1, C, 1.0f / sqrtf((float)C));
                }
#endif
            });
            arena.release(vT); arena.release(qk);
        } else {
        const int64_t S = arena.alloc((int64_t)T * T * 4);
            op(CLS_GEMM, [=](const Ctx& c) {             // S[b] = q[b] k[b]^T / sqrt(C)
                GemmArgs g = gemm_defaults();
                g.a0 = c.at<bf16>(qk); g.a0_ld = 2 * C; g.a0_C = C; g.a_bs = (int64_t)T * 2 * C; g.M = T; g.N = T;
                g.b = c.at<bf16>(qk) + C; g.b_ld = 2 * C; g.b_bs = (int64_t)T * 2 * C;
                g.scale = 1.0f / sqrtf((float)C);
                g.c = c.at<float>(S); g.c_ld = T; g.c_bs = (int64_t)T * T; g.c_mode = OUT_F32; g.batch = c.B;
                launch_gemm(g, c.stream);
            });
            const int64_t P = arena.alloc((int64_t)T * T * 2);
            op(CLS_OTHER, [=](const Ctx& c) {
                const int64_t rows = (int64_t)c.B * T;
                hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c.stream,
                                   c.at<float>(S), c.at<bf16>(P), T, rows);
            });
            arena.release(S);
            op(CLS_GEMM, [=](const Ctx& c) {             // O[b] = P[b] V[b]
                GemmArgs g = gemm_defaults();
                g.a0 = c.at<bf16>(P); g.a0_ld = T; g.a0_C = T; g.a_bs = (int64_t)T * T; g.M = T; g.N = C;
                g.b = c.at<bf16>(vT); g.b_ld = T; g.b_bs = (int64_t)C * T;
                g.c = c.act(O); g.c_ld = C; g.c_bs = (int64_t)T * C; g.batch = c.B;
                launch_gemm(g, c.stream);
            });
            arena.release(P); arena.release(vT); arena.release(qk);
        }
        if (proj) {
            arena.release(O.off); arena.release(own_sc); arena.release(own_sh);
            if (sc != own_sc) { arena.release(sc); arena.release(sh); }
            E.taps[m.idx] = out;
            return;
        }
        const Part po = register_output(out);
        const float rs = res_scale;
        op(CLS_GEMM, [=](const Ctx& c) {             // out = (x + O W3 + b3) / sqrt(2)   (`ddpm`: no rescale)
            GemmArgs g = gemm_defaults();
            g.a0 = c.act(O); g.a0_ld = C; g.a0_C = C; g.M = c.B * T; g.N = C;
            g.b = c.w<bf16>(w3); g.b_ld = C; g.bias_n = c.w<float>(b3);
            g.resid = c.act(x); g.resid_ld = x.ld; g.scale = rs;
            g.c = c.act(out); g.c_ld = out.ld;
            if (po.valid) { g.gn_part = c.at<float>(po.off); g.gn_quads = po.quads; }
            const int bm = launch_gemm(g, c.stream);
            if (po.valid) c.part_bm[po.id] = bm;
        });
        arena.release(O.off); arena.release(own_sc); arena.release(own_sh);
        if (sc != own_sc) { arena.release(sc); arena.release(sh); }
        E.taps[m.idx] = out;
    }

    // ---- `ddpm` resampling modules (layers.py:586-612) ------------------------------------------------
    // Downsample: 3x3 convolution, stride 2, 'SAME' padding emulated by one zero row / column at the bottom / right.  Three launches per
    // forward: an im2col pass (k_inc_im2col, K order tap-major) + one GEMM of the shared tile family.
    void emit_downconv(const Mod& m, const TRef& x, const TRef& out) {
        const int C = m.cin, ro = m.res / 2, Kd = 9 * C;
        const int64_t pw = take((int64_t)C * C * 9), pb = take(C);
        const int64_t w = wres((int64_t)C * Kd * 2);
        E.packs.push_back([=](const PackCtx& p) {              // k = tap * C + c (the im2col order)
            const int64_t n = (int64_t)C * C * 9;
            hipLaunchKernelGGL(k_pack_conv, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, p.stream, p.params + pw, reinterpret_cast<bf16*>(p.packed + w), C, C, 9, Kd, 0, C, 0, 1.0f);
        });
        const int64_t b = pack_f32(pb, C);
        const int64_t col = arena.alloc((int64_t)ro * ro * Kd * 2);
        const Part po = register_output(out);
        const int rin = m.res;
        op(CLS_GEMM, [=](const Ctx& c) {
            const int64_t M = (int64_t)c.B * ro * ro;
            if (!g_record) {
                const int64_t total = M * (Kd / 8);
                hipLaunchKernelGGL(k_inc_im2col, dim3(grid1d(total, 256, 1 << 30)), dim3(256), 0, c.stream, c.act(x), rin, rin, x.ld, C, 3, 3, 2, 0, 0, ro, ro, Kd,
                                   c.at<bf16>(col), total);
            }
            GemmArgs g = gemm_defaults();
            g.a0 = c.at<bf16>(col); g.a0_ld = Kd; g.a0_C = Kd; g.M = (int)M; g.N = C; g.b = c.w<bf16>(w); g.b_ld = Kd; g.bias_n = c.w<float>(b);
            g.log_rows_per_sample = 2 * ilog2(ro); g.logHW = 2 * ilog2(ro); g.logW = ilog2(ro);
            g.c = c.act(out); g.c_ld = out.ld;
            if (po.valid) { g.gn_part = c.at<float>(po.off); g.gn_quads = po.quads; }
            const int bm = launch_gemm(g, c.stream);
            if (po.valid) c.part_bm[po.id] = bm;
        });
        arena.release(col);
        E.taps[m.idx] = out;
    }
    // Upsample: nearest 2x, then a 3x3 convolution: the resampling pass (identity scale / shift, no activation) writes the zero-bordered
    // up-sampled copy the implicit GEMM reads
    int64_t ident_sc = -1, ident_sh = -1;
    void emit_upconv(const Mod& m, const TRef& x, const TRef& out) {
        const int C = m.cin, ro = m.res * 2, Ku = 9 * C;
        const int64_t pw = take((int64_t)C * C * 9), pb = take(C);
        const int64_t w = wres((int64_t)C * Ku * 2);
        pack_conv(pw, w, C, C, 9, Ku, 0, C);
        const int64_t b = pack_f32(pb, C);
        if (ident_sc < 0) {                                   // per (sample, channel) scale 1 / shift 0, filled once per forward
            ident_sc = arena.alloc((int64_t)2 * NF * 4); ident_sh = arena.alloc((int64_t)2 * NF * 4);
            const int64_t isc = ident_sc, ish = ident_sh;
            op(CLS_OTHER, [=](const Ctx& c) {
                const int64_t n = (int64_t)c.B * 2 * NF;
                hipLaunchKernelGGL(k_fill_f32, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, c.stream, c.at<float>(isc), 1.0f, n);
                hipLaunchKernelGGL(k_fill_f32, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, c.stream, c.at<float>(ish), 0.0f, n);
            });
        }
        TRef u = new_act(ro, C, 1);
        emit_gn_apply(x, ident_sc, ident_sh, u, nullptr, ACT_NONE, RS_UP);
        const Part po = register_output(out);
        const int logW = ilog2(ro);
        op(CLS_GEMM, [=](const Ctx& c) {
            GemmArgs g = gemm_defaults();
            g.a0 = c.act(u); g.a0_ld = u.ld; g.a0_C = C; g.taps = 9; g.logW = logW; g.logHW = 2 * logW; g.a0_padded = 1;
            g.M = c.B * ro * ro; g.N = C; g.b = c.w<bf16>(w); g.b_ld = Ku; g.bias_n = c.w<float>(b); g.log_rows_per_sample = 2 * logW;
            g.c = c.act(out); g.c_ld = out.ld;
            if (po.valid) { g.gn_part = c.at<float>(po.off); g.gn_quads = po.quads; }
            const int bm = launch_gemm(g, c.stream);
            if (po.valid) c.part_bm[po.id] = bm;
        });
        arena.release(u.off);
        E.taps[m.idx] = out;
    }

    int dense_rows_next = 0;

    // ---- fused GroupNorm statistics: partial tables written by GEMM epilogues ------------------------
    // Fin: the producer's epilogue writes its consumer's GroupNorm table (GemmArgs::fin_*).  The table buffers are allocated with the Part -- in front
    // of the producing launch, so they cannot alias anything that launch still reads -- and the consumer fills in its gamma / beta when it is emitted
    // (the producer's op reads the struct when it RUNS).  Only the first single-source consumer claims it; anyone else takes k_gn_finalize.
    // check (round 5, the 16x16 level): whether the producer's launch wrote the table is only known when it RUNS (k_conv_gn3's 256 x 256 tile does, the k_conv_gn2 tiles a
    // tuning knob may select instead do not): the producer's op records it in Ctx::fin_done (per forward), and the consumer's op launches k_gn_finalize into the same buffers when it is false
    struct Fin { int64_t sc = -1, sh = -1, gamma = -1, beta = -1; int C = 0; float out_mul = 1.0f; bool claimed = false, check = false; };
    struct Part { int64_t off = -1; int quads = 0, id = -1, res = 0; bool valid = false; std::shared_ptr<Fin> fin; };
    static void set_fin(GemmArgs& g, const Part& p, const Ctx& c) {
        if (!p.fin || !p.fin->claimed) return;
        const Fin& f = *p.fin;
        g.fin_scale = c.at<float>(f.sc); g.fin_shift = c.at<float>(f.sh); g.fin_gamma = c.w<float>(f.gamma); g.fin_beta = c.w<float>(f.beta);
        g.fin_ld = f.C; g.fin_cg = f.C / 32; g.fin_mul = f.out_mul; g.fin_eps = GN_EPS;
    }
    std::map<std::pair<int64_t, int>, Part> parts;      // (tensor offset, channel offset) -> latest producer's table
    int n_parts = 0;
    static bool fusable(int res) { return res >= 16; }   // every block tile (<= 256 rows) lies inside one sample
    Part new_part(int res, int C) {
        Part p; p.quads = C / 4; p.res = res; p.id = n_parts++; p.valid = true;
        p.off = arena.alloc((int64_t)std::max(res * res / 64, 1) * p.quads * 8);       // worst case: 64-row block tiles (4x4: one row per sample)
        return p;
    }
    // called for EVERY module output so that a stale table can never be matched to a later tensor at the same place
    // fused8: the producer is the fused convolution at 8x8, whose epilogue writes one partial row per SAMPLE (two per tile); any other producer's
    // 128-row tiles span two 8x8 samples, and its consumers take the streaming statistics kernel
    Part register_output(const TRef& out, bool fused8 = false) {
        Part p;
        if (fusable(out.res) || (out.res <= 8 && fused8)) p = new_part(out.res, out.C);
        if (p.valid && out.res <= 8 && fused8 && (g_fuse_fin & 1) && out.C == 256) {      // (the fused kernel's 64-pixel x 256-channel tile: whole samples, every channel)
            p.fin = std::make_shared<Fin>();
            p.fin->sc = arena.alloc((int64_t)out.C * 4); p.fin->sh = arena.alloc((int64_t)out.C * 4); p.fin->C = out.C;
        }
        // 16x16 (round 5): k_conv_gn3's 256 x 256 tile = ONE sample x every channel (fused8 here: "the producer is the fused convolution")
        if (p.valid && out.res == 16 && fused8 && (g_fuse_fin & 2) && out.C == 256 && !ddpm) {
            p.fin = std::make_shared<Fin>();
            p.fin->sc = arena.alloc((int64_t)out.C * 4); p.fin->sh = arena.alloc((int64_t)out.C * 4); p.fin->C = out.C; p.fin->check = true;
        }
        parts[{out.off, out.coff}] = p;
        return p;
    }
    // statistics of x from partial tables if every channel slice of x has a valid one; otherwise the streaming kernel.
    // (Folding the tables inside k_gn_apply instead of this one-block-per-sample launch was built and measured: bit-identical,
    // 1.6 % SLOWER per forward in a same-box A/B -- every 4-row apply block then starts with two dependent load round trips.)
    bool emit_gn_from_parts(const TRef& x, GN gn, int64_t& sc, int64_t& sh, float out_mul) {
        std::vector<Part> src;
        int ch = 0;
        while (ch < x.C) {
            auto it = parts.find({x.off, x.coff + ch});
            if (it == parts.end() || !it->second.valid || it->second.res != x.res) return false;
            src.push_back(it->second);
            ch += it->second.quads * 4;
        }
        if (ch != x.C || src.empty() || src.size() > 2) return false;
        const Part p0 = src[0], p1 = src.size() > 1 ? src[1] : Part();
        const int HW = x.res * x.res, C = x.C;
        if (src.size() == 1 && p0.fin && !p0.fin->claimed && p0.fin->C == C) {          // the producer's epilogue writes this table: no launch
            Fin& f = *p0.fin;
            f.gamma = gn.gamma; f.beta = gn.beta; f.out_mul = out_mul; f.claimed = true;
            sc = f.sc; sh = f.sh;
            if (f.check) {                                                              // ... unless the launch that ran was not the one that can (see Fin)
                const std::shared_ptr<Fin> fp = p0.fin;
                const int64_t fsc = f.sc, fsh = f.sh;
                op(CLS_OTHER, [=](const Ctx& c) {
                    if (c.fin_done[p0.id]) return;
                    hipLaunchKernelGGL(k_gn_finalize, dim3(c.B), dim3(256), 0, c.stream, c.at<float2>(p0.off), HW / c.part_bm[p0.id], p0.quads, (const float2*)nullptr, 0, 0, C, HW,
                                       c.w<float>(gn.gamma), c.w<float>(gn.beta), c.at<float>(fsc), c.at<float>(fsh), GN_EPS, out_mul);
                });
            }
            return true;
        }
        const int64_t sc_ = sc, sh_ = sh;
        op(CLS_OTHER, [=](const Ctx& c) {
            const int tps0 = HW / c.part_bm[p0.id], tps1 = p1.valid ? HW / c.part_bm[p1.id] : 0;
            hipLaunchKernelGGL(k_gn_finalize, dim3(c.B), dim3(256), 0, c.stream, c.at<float2>(p0.off), tps0, p0.quads,
                               p1.valid ? c.at<float2>(p1.off) : (const float2*)nullptr, tps1, p1.valid ? p1.quads : 0, C, HW,
                               c.w<float>(gn.gamma), c.w<float>(gn.beta), c.at<float>(sc_), c.at<float>(sh_), GN_EPS, out_mul);
        });
        return true;
    }

    // ---- module list (ncsnpp.py:66-230) ----------------------------------------------------
    void list_modules() {
        auto add = [&](int kind, int cin, int cout, int up, int down, int res) {
            Mod m; m.idx = (int)E.mods.size(); m.kind = kind; m.cin = cin; m.cout = cout; m.up = up; m.down = down; m.res = res; m.poff = 0;
            E.mods.push_back(m);
        };
        add(K_LIN, NF, TEMB, 0, 0, 0); add(K_LIN, TEMB, TEMB, 0, 0, 0); add(K_CONV, 3, NF, 0, 0, IMG);
        std::vector<int> skip = {NF};
        int ch = NF, res = IMG;
        for (int l = 0; l < NLEVEL; ++l) {
            for (int b = 0; b < NUM_RES; ++b) {
                add(K_RES, ch, NF * CH_MULT[l], 0, 0, res); ch = NF * CH_MULT[l];
                if (attn_at(res)) add(K_ATTN, ch, ch, 0, 0, res);
                skip.push_back(ch);
            }
            if (l != NLEVEL - 1) { if (ddpm) add(K_DOWN, ch, ch, 0, 0, res); else add(K_RES, ch, ch, 0, 1, res); res /= 2; skip.push_back(ch); }
        }
        add(K_RES, ch, ch, 0, 0, res); add(K_ATTN, ch, ch, 0, 0, res); add(K_RES, ch, ch, 0, 0, res);
        for (int l = NLEVEL - 1; l >= 0; --l) {
            for (int b = 0; b < NUM_RES + 1; ++b) { add(K_RES, ch + skip.back(), NF * CH_MULT[l], 0, 0, res); skip.pop_back(); ch = NF * CH_MULT[l]; }
            if (attn_at(res)) add(K_ATTN, ch, ch, 0, 0, res);
            if (l != 0) { if (ddpm) add(K_UP, ch, ch, 0, 0, res); else add(K_RES, ch, ch, 1, 0, res); res *= 2; }
        }
        add(K_GN, ch, ch, 0, 0, res); add(K_CONV, ch, 3, 0, 0, res);
    }

    void build() {
        list_modules();
        auto& M = E.mods;
        // ---- the U-Net's concat buffers: up-path res-block j reads cat([h, skip]) --------------
        struct Cat { TRef buf; int ch_h, ch_s; };
        std::vector<Cat> cats;
        std::vector<int> skip_ch, skip_res;
        {   // dry walk to size them
            int ch = NF, res = IMG; skip_ch.push_back(NF); skip_res.push_back(IMG);
            for (int l = 0; l < NLEVEL; ++l) {
                for (int b = 0; b < NUM_RES; ++b) { ch = NF * CH_MULT[l]; skip_ch.push_back(ch); skip_res.push_back(res); }
                if (l != NLEVEL - 1) { res /= 2; skip_ch.push_back(ch); skip_res.push_back(res); }
            }
            int sp = (int)skip_ch.size();
            for (int l = NLEVEL - 1; l >= 0; --l) {
                for (int b = 0; b < NUM_RES + 1; ++b) {
                    --sp;
                    Cat c; c.ch_h = ch; c.ch_s = skip_ch[sp];
                    c.buf = new_act(res, c.ch_h + c.ch_s);
                    cats.push_back(c);
                    ch = NF * CH_MULT[l];
                }
                if (l != 0) res *= 2;
            }
        }
        const int P = (int)skip_ch.size();
        auto skip_slot = [&](int i) { const Cat& c = cats[P - 1 - i]; TRef t = c.buf; t.C = c.ch_s; t.coff = c.ch_h; return t; };
        auto h_slot = [&](int j) { const Cat& c = cats[j]; TRef t = c.buf; t.C = c.ch_h; t.coff = 0; return t; };

        // ---- shared time-embedding bank -------------------------------------------------------
        for (auto& m : M) if (m.kind == K_RES) dense_total += m.cout;
        dense_w = wres((int64_t)dense_total * TEMB * 2); dense_b = wres((int64_t)dense_total * 4);
        dense_out = arena.alloc((int64_t)dense_total * 4);
        const int64_t emb = arena.alloc(128 * 2), t1 = arena.alloc(TEMB * 2), t2 = arena.alloc(TEMB * 2);

        size_t mi = 0;
        // modules 0, 1: Linear(128,512), Linear(512,512)
        {
            const Mod& m0 = M[mi++]; const Mod& m1 = M[mi++]; (void)m0; (void)m1;
            const int64_t pw0 = take((int64_t)TEMB * NF), pb0 = take(TEMB), pw1 = take((int64_t)TEMB * TEMB), pb1 = take(TEMB);
            const int64_t w0 = wres((int64_t)TEMB * NF * 2), w1 = wres((int64_t)TEMB * TEMB * 2);
            pack_conv(pw0, w0, TEMB, NF, 1, NF, 0, NF); pack_conv(pw1, w1, TEMB, TEMB, 1, TEMB, 0, TEMB);
            const int64_t b0 = pack_f32(pb0, TEMB), b1 = pack_f32(pb1, TEMB);
            const int64_t dw = dense_w, db = dense_b, dout = dense_out; const int dtot = dense_total;
            op(CLS_OTHER, [=](const Ctx& c) {
                hipLaunchKernelGGL(k_time_embed, dim3(grid1d((int64_t)c.B * 128)), dim3(256), 0, c.stream, c.labels, c.at<bf16>(emb), c.B);
            });
            op(CLS_GEMM, [=](const Ctx& c) {
                GemmArgs g = gemm_defaults();                        // act(Linear_0(emb))
                g.a0 = c.at<bf16>(emb); g.a0_ld = NF; g.a0_C = NF; g.M = c.B; g.N = TEMB;
                g.b = c.w<bf16>(w0); g.b_ld = NF; g.bias_n = c.w<float>(b0); g.act = ACT_SILU;
                g.c = c.at<bf16>(t1); g.c_ld = TEMB;
                launch_gemm(g, c.stream);
                g = gemm_defaults();                                 // act(temb) = act(Linear_1(.))
                g.a0 = c.at<bf16>(t1); g.a0_ld = TEMB; g.a0_C = TEMB; g.M = c.B; g.N = TEMB;
                g.b = c.w<bf16>(w1); g.b_ld = TEMB; g.bias_n = c.w<float>(b1); g.act = ACT_SILU;
                g.c = c.at<bf16>(t2); g.c_ld = TEMB;
                launch_gemm(g, c.stream);
                g = gemm_defaults();                                 // every block's Dense_0(act(temb)) at once
                g.a0 = c.at<bf16>(t2); g.a0_ld = TEMB; g.a0_C = TEMB; g.M = c.B; g.N = dtot;
                g.b = c.w<bf16>(dw); g.b_ld = TEMB; g.bias_n = c.w<float>(db);
                g.c = c.at<float>(dout); g.c_ld = dtot; g.c_mode = OUT_F32;
                launch_gemm(g, c.stream);
            });
        }
        // module 2: stem conv 3 -> 128 as im2col (K padded 27 -> 64) + GEMM
        int si = 0;
        TRef cur = skip_slot(si++);
        {
            const Mod& m = M[mi++];
            const int64_t pw = take((int64_t)NF * 27), pb = take(NF);
            const int64_t w = wres((int64_t)NF * 64 * 2);
            pack_zero(w, (int64_t)NF * 64);
            pack_conv(pw, w, NF, 3, 9, 64, 0, 3);
            const int64_t b = pack_f32(pb, NF);
            const int64_t a0 = arena.alloc((int64_t)IMG * IMG * 64 * 2);
            const TRef dst = cur;
            const Part po = register_output(dst);
            op(CLS_OTHER, [=](const Ctx& c) {
                const int64_t rows = (int64_t)c.B * IMG * IMG;
                hipLaunchKernelGGL(k_stem_im2col, dim3((unsigned)((rows * 8 + 255) / 256)), dim3(256), 0, c.stream, c.x, c.at<bf16>(a0), rows);
            });
            op(CLS_GEMM, [=](const Ctx& c) {
                const int64_t rows = (int64_t)c.B * IMG * IMG;
                GemmArgs g = gemm_defaults();
                g.a0 = c.at<bf16>(a0); g.a0_ld = 64; g.a0_C = 64; g.M = (int)rows; g.N = NF;
                g.b = c.w<bf16>(w); g.b_ld = 64; g.bias_n = c.w<float>(b);
                g.c = c.act(dst); g.c_ld = dst.ld;
                if (po.valid) { g.gn_part = c.at<float>(po.off); g.gn_quads = po.quads; }
                const int bm = launch_gemm(g, c.stream);
                if (po.valid) c.part_bm[po.id] = bm;
            });
            arena.release(a0);
            E.taps[m.idx] = dst;
        }
        // ---- down path ------------------------------------------------------------------------
        int res = IMG;
        for (int l = 0; l < NLEVEL; ++l) {
            for (int b = 0; b < NUM_RES; ++b) {
                const Mod& m = M[mi++];
                const TRef dst = skip_slot(si++);
                if (attn_at(res)) {
                    TRef tmp = new_act(res, m.cout);
                    emit_res(m, cur, tmp);
                    emit_attn(M[mi++], tmp, dst);
                    arena.release(tmp.off);
                } else {
                    emit_res(m, cur, dst);
                }
                cur = dst;
            }
            if (l != NLEVEL - 1) {
                const Mod& m = M[mi++];
                const TRef dst = skip_slot(si++);
                if (ddpm) emit_downconv(m, cur, dst); else emit_res(m, cur, dst);
                cur = dst; res /= 2;
            }
        }
        // ---- middle ---------------------------------------------------------------------------
        {
            TRef a = new_act(res, cur.C), b = new_act(res, cur.C);
            emit_res(M[mi++], cur, a);
            emit_attn(M[mi++], a, b);
            arena.release(a.off);
            emit_res(M[mi++], b, h_slot(0));
            arena.release(b.off);
        }
        // ---- up path --------------------------------------------------------------------------
        int j = 0;
        TRef last;
        for (int l = NLEVEL - 1; l >= 0; --l) {
            for (int b = 0; b < NUM_RES + 1; ++b) {
                const Mod& m = M[mi++];
                const bool level_end = b == NUM_RES;
                TRef in = cats[j].buf;                           // cat([h, skip]) along channels
                TRef dst = level_end ? new_act(res, m.cout) : h_slot(j + 1);
                emit_res(m, in, dst);
                arena.release(cats[j].buf.off);
                last = dst; ++j;
            }
            if (attn_at(res)) {
                TRef dst = new_act(res, last.C);
                emit_attn(M[mi++], last, dst);
                arena.release(last.off);
                last = dst;
            }
            if (l != 0) {
                const Mod& m = M[mi++];
                if (ddpm) emit_upconv(m, last, h_slot(j)); else emit_res(m, last, h_slot(j));
                arena.release(last.off);
                res *= 2;
            }
        }
        // ---- head: GroupNorm -> SiLU -> conv3x3 128 -> 3, written as fp32 NCHW -----------------
        {
            const Mod& mg = M[mi++]; const Mod& mc = M[mi++];
            const GN gn = take_gn(mg.cin);
            const int64_t pw = take((int64_t)3 * mc.cin * 9), pb = take(3);
            const int Kf = 9 * mc.cin;
            const int64_t w = wres((int64_t)3 * Kf * 2);
            pack_conv(pw, w, 3, mc.cin, 9, Kf, 0, mc.cin);
            const int64_t b = pack_f32(pb, 3);
            int64_t sc = arena.alloc((int64_t)mg.cin * 4), sh = arena.alloc((int64_t)mg.cin * 4);          // (32x32: never a producer-written table)
            const int logW = ilog2(res), cinf = mc.cin;
            if (g_fuse_head && res == HeadConvCfg::RES && mc.cin == HeadConvCfg::C) {
                // one launch (head_conv.h): raw tensor in, fp32 NCHW out; folded GroupNorm form (scale / shift x -log2 e, weights x -ln 2)
                const int64_t w16 = wres((int64_t)16 * Kf * 2);
                pack_zero(w16, (int64_t)16 * Kf);
                pack_conv(pw, w16, 3, mc.cin, 9, Kf, 0, mc.cin, -0.6931471805599453f);
                emit_gn_stats(last, gn, sc, sh, -1.4426950408889634f);
                const TRef src = last;
                op(CLS_GEMM, [=](const Ctx& c) {
                    if (g_record) {
                        char line[160];
                        snprintf(line, sizeof(line), "%d %d %d %d %d %d head_conv/e0\n", c.B * res * res, 3, Kf, 0, 9, 1);
                        *g_record += line;
                        return;
                    }
                    hipLaunchKernelGGL(k_head_conv, dim3((unsigned)(c.B * (HeadConvCfg::RES / HeadConvCfg::ROWS))), dim3(256), HeadConvCfg::LDS_BYTES, c.stream,
                                       c.act(src), src.ld, c.at<float>(sc), c.at<float>(sh), c.w<bf16>(w16), c.w<float>(b), c.out);
                });
            } else {
            emit_gn_stats(last, gn, sc, sh);
            TRef u = new_act(res, mg.cin, 1);
            emit_gn_apply(last, sc, sh, u, nullptr, ACT_SILU, RS_NONE);
            op(CLS_GEMM, [=](const Ctx& c) {
                GemmArgs g = gemm_defaults();
                g.a0 = c.act(u); g.a0_ld = u.ld; g.a0_C = cinf; g.taps = 9; g.logW = logW; g.logHW = 2 * logW; g.a0_padded = 1;
                g.M = c.B * res * res; g.N = 3; g.b = c.w<bf16>(w); g.b_ld = Kf; g.bias_n = c.w<float>(b);
                g.c = c.out; g.c_mode = OUT_F32_NCHW;
                launch_gemm(g, c.stream);
            });
            }
            E.taps[mg.idx] = last;          // (pre-norm tensor; the GN module's own output is internal)
        }
        E.n_params = poff;
        E.part_bm.assign(n_parts > 0 ? n_parts : 1, 128);
        E.fin_done.assign(n_parts > 0 ? n_parts : 1, 0);
        E.ws_per_image = arena.peak;
        E.packed_bytes = wtop;
        // parameter offsets for describe(): recompute by module in order
    }
};

int64_t module_param_count(const Mod& m) {
    switch (m.kind) {
        case K_LIN: return (int64_t)m.cout * m.cin + m.cout;
        case K_CONV: return (int64_t)m.cout * m.cin * 9 + m.cout;
        case K_GN: return 2 * (int64_t)m.cin;
        case K_DOWN: case K_UP: return (int64_t)m.cout * m.cin * 9 + m.cout;
        case K_ATTN: return 2 * (int64_t)m.cin + 4 * ((int64_t)m.cin * m.cin + m.cin);
        case K_RES: {
            int64_t n = 2 * (int64_t)m.cin + (int64_t)m.cout * m.cin * 9 + m.cout + (int64_t)m.cout * TEMB + m.cout +
                        2 * (int64_t)m.cout + (int64_t)m.cout * m.cout * 9 + m.cout;
            if (m.cin != m.cout || m.up || m.down) n += (int64_t)m.cout * m.cin + m.cout;
            return n;
        }
    }
    return 0;
}

natinf_ncsnpp* make_engine(int flags) {
    natinf_ncsnpp* e = new natinf_ncsnpp();
    e->flags = flags;
    Builder b(*e);
    b.build();
    int64_t off = 0;
    for (auto& m : e->mods) { m.poff = off; off += module_param_count(m); }
    if (off != e->n_params) { delete e; return nullptr; }      // plan and parameter walk disagree
    return e;
}

// The size queries without a handle answer for the LARGEST plan (every build-time fusion on: each adds repacked weight copies), whatever the
// A/B knobs are set to when they are first asked -- a packed buffer of that size fits every plan.
const natinf_ncsnpp& reference_engine() {
    static natinf_ncsnpp* e = [] {
        // every knob that is read when a plan is BUILT and adds a repacked weight copy or a table: saved, forced on, restored
        int* knobs[] = {&g_fuse_gn, &g_fuse_up, &g_fuse_head, &g_fuse_gn8, &g_fuse_gn4, &g_fuse_fin, &g_attn_qkv, &g_attn_proj, &g_attn256};
        int saved[sizeof(knobs) / sizeof(knobs[0])];
        for (size_t i = 0; i < sizeof(knobs) / sizeof(knobs[0]); ++i) { saved[i] = *knobs[i]; *knobs[i] = 1; }
        g_fuse_fin = 3;                                      // (a bit mask: both levels' producer-written tables)
        natinf_ncsnpp* r = make_engine(0);
        for (size_t i = 0; i < sizeof(knobs) / sizeof(knobs[0]); ++i) *knobs[i] = saved[i];
        return r;
    }();
    return *e;
}

}  // namespace

extern "C" {

int64_t natinf_ncsnpp_param_count(void) { return reference_engine().n_params; }
int64_t natinf_ncsnpp_packed_bytes(void) { return reference_engine().packed_bytes; }

int64_t natinf_ncsnpp_handle_param_count(natinf_ncsnpp_t h) { return h ? h->n_params : NATINF_EINVAL; }
int64_t natinf_ncsnpp_handle_packed_bytes(natinf_ncsnpp_t h) { return h ? h->packed_bytes : NATINF_EINVAL; }

int64_t natinf_ncsnpp_workspace_bytes(natinf_ncsnpp_t h, int max_batch) {
    if (!h || max_batch <= 0) return NATINF_EINVAL;
    return h->ws_per_image * (int64_t)max_batch;
}

// every switch a plan builder reads (layout of the packed weights included), one byte each
static uint64_t plan_signature() {
    const int k[] = {g_fuse_head, g_cg8_tm4, g_attn_qkv, g_attn_w8, g_attn_proj, g_attn256, g_fuse_gn8, g_fuse_fin, g_fuse_gn4, g_fuse_gn, g_cg_wide, g_fuse_up, g_cg_regw,
                     g_attn_blk /* (2: the folded attention weights are packed too) */, g_cg3 /* read by the plan builders (which launches exist; which tables a producer may write) */};
    uint64_t h = 1469598103934665603ull;
    for (int v : k) h = (h ^ (uint64_t)(v & 0xff)) * 1099511628211ull;
    return h;
}

int natinf_ncsnpp_create(natinf_ncsnpp_t* out, int flags) {
    if (!out || (flags & ~(NATINF_NCSNPP_KEEP_ACTIVATIONS | NATINF_NCSNPP_DDPM))) return NATINF_EINVAL;
    natinf_ncsnpp* e = make_engine(flags);
    if (!e) return NATINF_ESTATE;
    e->plan_sig = plan_signature();
    *out = e;
    return NATINF_OK;
}

int natinf_ncsnpp_destroy(natinf_ncsnpp_t h) {
    if (!h) return NATINF_EINVAL;
    for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    delete h;
    return NATINF_OK;
}

int natinf_ncsnpp_describe(natinf_ncsnpp_t h, char* buf, int cap) {
    if (!h || !buf || cap <= 0) return NATINF_EINVAL;
    std::string s;
    char line[160];
    for (const auto& m : h->mods) {
        snprintf(line, sizeof(line), "%d %s %d %d %d %d %d %lld\n", m.idx, kind_name(m.kind), m.cin, m.cout, m.up, m.down, m.res,
                 (long long)m.poff);
        s += line;
    }
    if ((int)s.size() + 1 > cap) return NATINF_EINVAL;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int natinf_ncsnpp_load(natinf_ncsnpp_t h, const float* params_f32, int64_t n_params, void* packed, int64_t packed_bytes,
                       natinf_stream_t stream) {
    if (!h || !params_f32 || !packed || n_params != h->n_params || packed_bytes < h->packed_bytes) return NATINF_EINVAL;
    PackCtx p{params_f32, reinterpret_cast<unsigned char*>(packed), (hipStream_t)stream};
    for (const auto& f : h->packs) f(p);
    h->packed = reinterpret_cast<const unsigned char*>(packed);
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

int natinf_ncsnpp_share(natinf_ncsnpp_t h, natinf_ncsnpp_t loaded) {
    if (!h || !loaded || h == loaded) return NATINF_EINVAL;
    if (!loaded->packed) return NATINF_ESTATE;
    // same network, same plan-build-time switches: the two plans address the packed buffer identically
    if ((h->flags & NATINF_NCSNPP_DDPM) != (loaded->flags & NATINF_NCSNPP_DDPM) || h->n_params != loaded->n_params || h->packed_bytes != loaded->packed_bytes ||
        h->packs.size() != loaded->packs.size() || h->plan_sig != loaded->plan_sig)
        return NATINF_EINVAL;
    h->packed = loaded->packed;
    return NATINF_OK;
}

int natinf_ncsnpp_forward(natinf_ncsnpp_t h, const float* x, const float* labels, float* out, int B, void* workspace,
                          int64_t workspace_bytes, natinf_stream_t stream) {
    if (!h || !x || !labels || !out || !workspace || B <= 0) return NATINF_EINVAL;
    if (!h->packed) return NATINF_ESTATE;
    if (workspace_bytes < h->ws_per_image * (int64_t)B || (int64_t)B * IMG * IMG >= (1LL << 31)) return NATINF_EINVAL;
    if (!h->attr_set) {
        if (!configure_gemm_kernels()) return NATINF_ENODEV;
        h->attr_set = true;
    }
    Ctx c{B, reinterpret_cast<unsigned char*>(workspace), h->packed, (hipStream_t)stream, x, labels, out, h->part_bm.data()};
    c.fin_done = h->fin_done.data();
    g_launch_error = 0;
    if (!h->prof) {
        for (const auto& f : h->ops) f(c);
    } else {
        for (size_t i = 0; i < h->ops.size(); ++i) {
            natinf_ncsnpp::Rec r; r.cls = h->op_cls[i];
            for (hipEvent_t* e : {&r.a, &r.b}) {
                if (!h->pool.empty()) { *e = h->pool.back(); h->pool.pop_back(); }
                else if (hipEventCreate(e) != hipSuccess) return NATINF_ELAUNCH;
            }
            (void)hipEventRecord(r.a, c.stream);
            h->ops[i](c);
            (void)hipEventRecord(r.b, c.stream);
            h->recs.push_back(r);
        }
    }
    h->last_B = B; h->last_ws = c.ws;
    if (g_launch_error) { g_launch_error = 0; return NATINF_ESTATE; }
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

int natinf_ncsnpp_describe_gemms(natinf_ncsnpp_t h, int B, char* buf, int cap) {
    if (!h || !buf || cap <= 0 || B <= 0) return NATINF_EINVAL;
    std::string out;
    g_record = &out;
    std::vector<int> scratch_bm(h->part_bm.size(), 128);
    // (a non-null fake workspace base: launches are only described, but "is this pointer set" decides the kernel variant)
    std::vector<unsigned char> scratch_done(h->part_bm.size(), 0);
    Ctx c{B, reinterpret_cast<unsigned char*>(4096), reinterpret_cast<const unsigned char*>(4096), nullptr, nullptr, nullptr, nullptr, scratch_bm.data()};
    c.fin_done = scratch_done.data();
    for (size_t i = 0; i < h->ops.size(); ++i)
        if (h->op_cls[i] == CLS_GEMM || h->op_cls[i] == CLS_CONV_GN || h->op_cls[i] == CLS_CONV_GN8) h->ops[i](c);          // GEMM ops only compute pointers and call launch_gemm
    g_record = nullptr;
    if ((int)out.size() + 1 > cap) return NATINF_EINVAL;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

int natinf_debug_gemm(int variant, int M, int N, int K0, int K1, int taps, int logW, int batch,
                      const void* a0, const void* a1, const void* b, const float* bias_n, void* c, int c_f32, float scale,
                      int iters, natinf_stream_t stream) {
    if (variant < 0 || variant >= V_COUNT || !a0 || !b || !c || M <= 0 || N <= 0 || iters <= 0 || (taps != 1 && taps != 9)) return NATINF_EINVAL;
    if (!variant_shipped(variant) || variant == V_CONV_GN || variant == V_FP8_256x256) return NATINF_ESTATE;      // superseded / ablation variants: -DNATINF_DEV builds
#ifndef NATINF_DEV
    if (c_f32 >= 2) return NATINF_ESTATE;      // timing experiments: -DNATINF_DEV builds
#endif
    static bool configured = false;
    if (!configured) { if (!configure_gemm_kernels()) return NATINF_ENODEV; configured = true; }
    GemmArgs g = gemm_defaults();
    g.a0 = (const bf16*)a0; g.a0_C = K0 / taps; g.a0_ld = g.a0_C; g.taps = taps; g.logW = logW; g.logHW = 2 * logW; g.a0_padded = taps == 9;
    if (a1) { g.a1 = (const bf16*)a1; g.a1_C = K1; g.a1_ld = K1; }
    g.M = M; g.N = N; g.b = (const bf16*)b; g.b_ld = K0 + (a1 ? K1 : 0); g.batch = batch;
    if (batch > 1) { g.a_bs = (int64_t)M * g.a0_ld; g.b_bs = (int64_t)N * g.b_ld; g.c_bs = (int64_t)M * N; }
    g.bias_n = bias_n; g.scale = scale; g.c = c; g.c_ld = N; g.c_mode = c_f32 == 1 ? OUT_F32 : (c_f32 >= 2 ? 100 + c_f32 : OUT_BF16);      // >= 2: timing experiments (tools/bench_gemm.py)
    g.dbg_ts = g_dbg_ts;
    g.splitk_ws = g_dbg_splitk_ws; g.splitk_max = g_dbg_splitk_max;
    const int saved = g_force_variant;
    g_force_variant = variant;
    for (int i = 0; i < iters; ++i) launch_gemm(g, (hipStream_t)stream);
    g_force_variant = saved;
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

// One plain GEMM C = A B^T with the fused epilogue terms of GemmArgs (tests/test_gpu_gemm_epilogue.py): every pointer but a / b / c
// may be NULL.  rowvec / gate are [samples][N] tables indexed by row >> log_rows_per_sample; gn_part receives (sum, sum of
// squares) per block tile and 4-column quad ([ceil(M / *bm_out)][N / 4] float2).  fp32_slab forces the general epilogue.
int natinf_debug_gemm_fused(int variant, int M, int N, int K, const void* a, const void* b, const float* bias_n, const float* bias_m,
                            const float* rowvec, const float* gate, int log_rows_per_sample, const void* resid_bf16, const float* resid_f32,
                            float scale, int act, void* c, int c_f32, float* gn_part, int* bm_out, int fp32_slab, natinf_stream_t stream) {
    if (variant < 0 || variant >= V_COUNT || !a || !b || !c || M <= 0 || N <= 0 || K <= 0 || N % 8) return NATINF_EINVAL;
    if (!variant_shipped(variant) || variant == V_CONV_GN || variant == V_FP8_256x256) return NATINF_ESTATE;
    if (!configure_gemm_kernels()) return NATINF_ENODEV;
    GemmArgs g = gemm_defaults();
    g.a0 = (const bf16*)a; g.a0_C = K; g.a0_ld = K; g.M = M; g.N = N; g.b = (const bf16*)b; g.b_ld = K;
    g.bias_n = bias_n; g.bias_m = bias_m; g.rowvec = rowvec; g.rowvec_ld = N; g.gate = gate; g.gate_ld = N; g.log_rows_per_sample = log_rows_per_sample;
    g.resid = (const bf16*)resid_bf16; g.resid_ld = N; g.resid_f32 = resid_f32; g.resid_f32_ld = N;
    g.scale = scale; g.act = act; g.c = c; g.c_ld = N; g.c_mode = c_f32 ? OUT_F32 : OUT_BF16;
    g.gn_part = gn_part; g.gn_quads = N / 4; g.epi_fp32_slab = fp32_slab != 0;
    g.splitk_ws = g_dbg_splitk_ws; g.splitk_max = g_dbg_splitk_max;
    g.dbg_ts = g_dbg_ts;
    const int saved = g_force_variant;
    g_force_variant = variant;
    const int bm = launch_gemm(g, (hipStream_t)stream);
    g_force_variant = saved;
    if (bm_out) *bm_out = bm;
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

// The fused GroupNorm-apply + SiLU + 3x3 convolution kernel on its own (tests/test_gpu_conv_gn.py, tools/bench_conv_gn.py):
//   out[m, n] = (sum_{tap, c} silu(x[pixel(m) + tap, c] * scale[b, c] + shift[b, c]) * w[n, c, tap]  +  sum_c a1[m, c] * w1[n, c]
//                + bias[n] + resid[m, n]) * out_scale,   zero padding outside the image (applied AFTER the activation).
// x: raw bf16 [B][res][res][cin]; w_packed: bf16 [N][9*cin + c1] in the engine's K order ((c / 64) * 9 + tap) * 64 + c % 64, then
// the c1 shortcut columns; a1: bf16 [B*res*res][c1] or NULL (c1 = 0); resid: bf16 [M][N] or NULL; gn_part: NULL or
// [M / 256][N / 4] float2 partial (sum, sum of squares) of the outputs.  The kernel takes its operands in FOLDED form
// (GemmArgs::gn_folded): the caller passes scale * -log2(e), shift * -log2(e) and the 3x3 columns of w_packed * -ln 2.
int g_dbg_cg_up = 0;
// bit 0: the next natinf_debug_conv_gn calls read x as [B][res/2][res/2][cin] through the kernel's nearest-2x up-sampling fetch (GemmArgs::a0_up);
// bit 1: the same for a1 ([B*(res/2)^2][c1], GemmArgs::a1_up) -- the fetch paths of the up-sampling res-blocks (natinf_set_fuse_up)
int natinf_debug_conv_gn_up(int flags) { if (flags & ~3) return NATINF_EINVAL; g_dbg_cg_up = flags; return NATINF_OK; }
int natinf_debug_conv_gn(int res, int B, int N, int cin, int c1, const void* x, const float* scale, const float* shift, const void* w_packed,
                         void* w_frag, const void* a1, const float* bias_n, const void* resid, float out_scale, void* out, float* gn_part, int iters,
                         natinf_stream_t stream) {
    if ((res != 32 && res != 16 && res != 8 && res != 4) || B <= 0 || N <= 0 || N % 8 || cin <= 0 || cin % 64 || c1 < 0 || c1 % 64 || (c1 > 0) != (a1 != nullptr) ||
        !x || !scale || !shift || !w_packed || !out || iters <= 0) return NATINF_EINVAL;
    static bool configured = false;
    if (!configured) { if (!configure_gemm_kernels()) return NATINF_ENODEV; configured = true; }
    GemmArgs g = gemm_defaults();
    g.a0 = (const bf16*)x; g.a0_ld = cin; g.a0_C = cin; g.taps = 9; g.logW = ilog2(res); g.logHW = 2 * g.logW;
    g.gn_scale = scale; g.gn_shift = shift; g.gn_ld = cin; g.gn_folded = 1;
    if (a1) { g.a1 = (const bf16*)a1; g.a1_ld = c1; g.a1_C = c1; }
    g.M = B * res * res; g.N = N; g.b = (const bf16*)w_packed; g.b_ld = 9 * cin + c1; g.bias_n = bias_n;
    g.resid = (const bf16*)resid; g.resid_ld = N; g.scale = out_scale; g.c = out; g.c_ld = N;
    g.gn_part = gn_part; g.gn_quads = N / 4;
    g.dbg_ts = g_dbg_ts;
    g.a0_up = g_dbg_cg_up & 1; g.a1_up = (a1 && (g_dbg_cg_up & 2)) ? 1 : 0;      // natinf_debug_conv_gn_up: x / a1 are given at HALF the resolution
    if (w_frag && N % 16 == 0) g.b_frag = (const bf16*)w_frag;      // k_conv_gn2 (used when N is a whole number of its column tiles): w_frag receives the fragment-major copy
    if (!conv_gn_ok(g)) return NATINF_EINVAL;                        // (shipped builds: k_conv_gn2 only -- w_frag is required and N % 128 == 0)
    if (g.b_frag) {
        const int64_t n = (int64_t)(N / 16) * (9 * (cin / 32) + c1 / 32) * 64;
        hipLaunchKernelGGL(k_pack_frag, dim3(grid1d(n, 256, 1 << 30)), dim3(256), 0, (hipStream_t)stream, (const bf16*)w_packed, (bf16*)w_frag, N, g.b_ld, cin, c1);
    }
    for (int i = 0; i < iters; ++i) launch_gemm(g, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

int natinf_debug_quant_fp8_rows(const float* w, void* q, float* row_scale, int rows, int cols, natinf_stream_t stream) {
    if (!w || !q || !row_scale || rows <= 0 || cols <= 0 || cols % 2) return NATINF_EINVAL;
    hipLaunchKernelGGL(k_pack_fp8_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, (uint8_t*)q, row_scale, rows, cols);
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

int natinf_debug_gemm_fp8(int M, int N, int K, const void* a8, const float* a_scale, const void* a_mx, const void* b8, const float* b_scale,
                          const float* bias_n, void* c, void* c_mx, int c_mode, int iters, natinf_stream_t stream) {
    if (!a8 || !b8 || !c || M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 8 || iters <= 0) return NATINF_EINVAL;
    static bool configured = false;
    if (!configured) { if (!configure_gemm_kernels()) return NATINF_ENODEV; configured = true; }
    GemmArgs g = gemm_defaults();
    g.a0 = (const bf16*)a8; g.a0_C = K; g.a0_ld = K; g.M = M; g.N = N; g.b = (const bf16*)b8; g.b_ld = K;
    const int act = c_mode >> 8; c_mode &= 0xff;                         // bits 8+: activation (2 = tanh-GELU: with c_mode 3 the fc1 epilogue of the MMDiT engine)
    if (act != ACT_NONE && act != ACT_GELU_TANH) return NATINF_EINVAL;
    g.deq_m = a_scale; g.deq_n = b_scale; g.bias_n = bias_n; g.c = c; g.c_ld = N; g.c_mode = c_mode; g.act = act;
    g.a_mx = (const uint8_t*)a_mx; g.a_mx_ld = M; g.c_mx = (uint8_t*)c_mx; g.c_mx_ld = M;       // K-tile-major planes of M rows
    if ((c_mode == OUT_FP8_MX && (!c_mx || N % 32)) || c_mode < 0 || c_mode > OUT_FP8_MX || c_mode == OUT_F32_NCHW) return NATINF_EINVAL;
    for (int i = 0; i < iters; ++i) launch_gemm_fp8(g, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

// timing experiments: device buffer of 16 uint64 s_memtime stamps written by block 0 / thread 0 of natinf_debug_gemm launches
int natinf_debug_timestamps(void* dev_buf16) {
#ifdef NATINF_DEV
    g_dbg_ts = reinterpret_cast<unsigned long long*>(dev_buf16); return NATINF_OK;
#else
    (void)dev_buf16; return NATINF_ESTATE;             // the shipped kernels carry no stamps (make EXTRA=-DNATINF_DEV)
#endif
}

int natinf_set_gemm_raster(int rows) { g_raster_g = rows; return NATINF_OK; }
int natinf_set_fuse_gn(int on) { g_fuse_gn = on != 0; return NATINF_OK; }
int natinf_set_fuse_gn8(int on) { g_fuse_gn8 = on != 0; return NATINF_OK; }
int natinf_set_attn256(int on) {
#ifndef NATINF_DEV
    if (!on) return NATINF_ESTATE;                 // k_attn_fused<8,16,true> is a development-build kernel
#endif
    g_attn256 = on != 0; return NATINF_OK;
}
int natinf_set_conv_gn_warm(int mask) { if (mask < 0 || mask > 15) return NATINF_EINVAL; g_cg_warm = mask; return NATINF_OK; }
int natinf_set_attn_block(int on) { g_attn_blk = on < 0 ? ATTN_BLK_DEFAULT : (on > 2 ? 2 : on); return NATINF_OK; }
int natinf_set_attn_qkv(int on) { g_attn_qkv = on != 0; return NATINF_OK; }
int natinf_set_attn_waves8(int on) { g_attn_w8 = on != 0; return NATINF_OK; }
int natinf_set_attn_proj(int on) { g_attn_proj = on != 0; return NATINF_OK; }
int natinf_set_fuse_fin(int on) {              // 1 = every level that can (the default), 0 = none, 2 = 8x8 / 4x4 only (the round-4 plan), 3 = 16x16 only
    if (on < 0 || on > 3) return NATINF_EINVAL;
    static const int mask[4] = {0, 3, 1, 2};    // g_fuse_fin: bit 0 = 8x8 / 4x4 (k_conv_gn2), bit 1 = 16x16 (k_conv_gn3)
    g_fuse_fin = mask[on];
    return NATINF_OK;
}
int natinf_set_gemm_round_model(int on) { g_round_model = on != 0; g_round_model_w128 = on >= 10 ? on : 13; return NATINF_OK; }
int natinf_set_gemm_w128(int on) { g_w128 = on < 0 ? 0 : (on > 2 ? 2 : on); return NATINF_OK; }
int natinf_set_fuse_gn4(int on) { g_fuse_gn4 = on != 0; return NATINF_OK; }
int natinf_set_conv_gn8_tile(int one_image) {
#ifndef NATINF_DEV
    if (!one_image) return NATINF_ESTATE;          // the two-image tile is a development-build kernel
#endif
    g_cg8_tm4 = one_image != 0; return NATINF_OK;
}
int natinf_set_fuse_head(int on) { g_fuse_head = on != 0; return NATINF_OK; }
int natinf_set_conv_gn_w128(int mask) { if (mask < 0 || mask > 7) return NATINF_EINVAL; g_cg3 = mask; return NATINF_OK; }
int natinf_set_conv_gn_w128_min_k(int shape, int k) { if (shape < 0 || shape > 2 || k < 0) return NATINF_EINVAL; g_cg3_min_k[shape] = k; return NATINF_OK; }
int natinf_set_conv_gn_wide(int mask) { if (mask < 0 || mask > 3) return NATINF_EINVAL; g_cg_wide = mask; return NATINF_OK; }
int natinf_set_conv_gn_regw(int on) {
    if (!on && !HAVE_CONV_GN_V1) return NATINF_ESTATE;      // k_conv_gn (the LDS-ring form) exists in -DNATINF_DEV builds only
    g_cg_regw = on != 0; return NATINF_OK;
}
int natinf_set_fuse_up(int on) { g_fuse_up = on != 0; return NATINF_OK; }
int natinf_set_gemm_splitk(int on) { g_splitk = on != 0; return NATINF_OK; }
int natinf_debug_set_splitk_workspace(float* ws, int max_slices) { g_dbg_splitk_ws = ws; g_dbg_splitk_max = ws ? max_slices : 0; return NATINF_OK; }
int natinf_set_gemm_half_issue(int on) {
#ifndef NATINF_DEV
    if (!on) return NATINF_ESTATE;                          // the every-wave-issues pipelines exist in -DNATINF_DEV builds only
#endif
    g_half_issue = on != 0; return NATINF_OK;
}
int natinf_set_gemm_pref512(int on) { g_pref_512 = on != 0; return NATINF_OK; }
int natinf_set_gemm_epilogue(int fp32_slab) { g_epi_fp32_slab = fp32_slab != 0; return NATINF_OK; }

int natinf_set_gemm_variant(int variant) {
    if (variant < 0 || variant >= V_COUNT) return NATINF_EINVAL;
    if (!variant_shipped(variant) || variant == V_CONV_GN || variant == V_FP8_256x256) return NATINF_ESTATE;      // development-build variants; operand-type-specific kernels
    g_force_variant = variant;
    return NATINF_OK;
}

int natinf_gemm_profile(int enable) { g_gemm_prof.on = enable != 0; return NATINF_OK; }
int natinf_gemm_profile_read(char* buf, int cap) {
    if (!buf || cap <= 0) return NATINF_EINVAL;
    std::map<std::string, std::pair<double, int64_t>> by;
    std::vector<std::string> order;
    int rc = NATINF_OK;
    for (auto& e : g_gemm_prof.ev) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) { (void)hipGetLastError(); rc = NATINF_ELAUNCH; }
        else {
            auto it = by.find(e.tag);
            if (it == by.end()) { order.push_back(e.tag); it = by.emplace(e.tag, std::make_pair(0.0, (int64_t)0)).first; }
            it->second.first += ms; it->second.second += 1;
        }
        g_gemm_prof.pool.emplace_back(e.a, e.b);
    }
    g_gemm_prof.ev.clear();
    if (rc != NATINF_OK) return rc;
    std::string out;
    for (auto& t : order) {
        char tail[64];
        snprintf(tail, sizeof(tail), " %lld %.6f\n", (long long)by[t].second, by[t].first);
        out += t + tail;
    }
    if ((int)out.size() + 1 > cap) return NATINF_EINVAL;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}
int natinf_ncsnpp_profile(natinf_ncsnpp_t h, int enable) {
    if (!h) return NATINF_EINVAL;
    h->prof = enable != 0;
    return NATINF_OK;
}

int natinf_ncsnpp_profile_read(natinf_ncsnpp_t h, double* ms_by_class, int64_t* launches_by_class) {
    if (!h || !ms_by_class || !launches_by_class) return NATINF_EINVAL;
    for (int i = 0; i < N_CLS; ++i) { ms_by_class[i] = 0.0; launches_by_class[i] = 0; }
    for (const auto& r : h->recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
            (void)hipGetLastError();
            return NATINF_ELAUNCH;
        }
        ms_by_class[r.cls] += ms; launches_by_class[r.cls] += 1;
        h->pool.push_back(r.a); h->pool.push_back(r.b);
    }
    h->recs.clear();
    return NATINF_OK;
}

int natinf_ncsnpp_debug_tap(natinf_ncsnpp_t h, int module_idx, float* out, int64_t capacity_elems, natinf_stream_t stream) {
    if (!h || !out) return NATINF_EINVAL;
    if (!(h->flags & NATINF_NCSNPP_KEEP_ACTIVATIONS) || !h->last_ws) return NATINF_ESTATE;
    auto it = h->taps.find(module_idx);
    if (it == h->taps.end()) return NATINF_EINVAL;
    const TRef& t = it->second;
    const int HW = t.res * t.res;
    const int64_t total = (int64_t)h->last_B * t.C * HW;
    if (capacity_elems < total) return NATINF_EINVAL;
    const bf16* src = reinterpret_cast<const bf16*>(h->last_ws + t.off * h->last_B) + t.coff;
    hipLaunchKernelGGL(k_nhwc_to_nchw_f32, dim3(grid1d(total, 256, 1 << 30)), dim3(256), 0, (hipStream_t)stream, src, t.ld, t.C, HW, out, total);
    return hipGetLastError() == hipSuccess ? NATINF_OK : NATINF_ELAUNCH;
}

}  // extern "C"
#include "dit_engine.inc"
#include "mmdit_engine.inc"
#include "vae_engine.inc"
#include "inception_engine.inc"
