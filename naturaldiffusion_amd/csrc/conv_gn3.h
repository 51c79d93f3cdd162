// conv_gn3.h -- k_conv_gn3: the fused GroupNorm-apply + SiLU + 3x3 convolution (+ 1x1 shortcut segment) of conv_gn2.h on the structure of gemm_w128.h:
// FOUR waves per block, ONE PER SIMD, wave tile 128 pixels x 128 channels, its 256 fp32 accumulators in AGPRs, and the K loop written out slot by slot
// (slot S = the instructions issued in front of MFMA S of a 64-MFMA K step; every instruction of the loop is a volatile asm statement, hipcc only
// allocates registers).  Reference arithmetic: layerspp.py:242-274 (h = Conv(act(GroupNorm(x))), Conv_1 + Conv_2 as one K range).
//
// What changes against k_conv_gn2 (two blocks of four 128 x 64 wave tiles per CU, hipcc-scheduled between sched_group_barriers):
//   * a K step (one tap of a 32-channel half-chunk) is 64 MFMAs per wave for 8 A-fragment reads and 8 weight-fragment loads: half the LDS fragment
//     reads per MFMA, and with the accumulators in AGPRs the 256 VGPRs hold TWO complete fragment sets -- the reads / loads of step k + 1 are
//     issued during step k and no MFMA ever waits for an LDS read issued just in front of it;
//   * block tiles: 512 pixels x 128 channels (WM = 4, WN = 1; N % 256 != 0 layers at 32x32: half an image per tile, halo 18 / 16 patch rows
//     instead of 10 / 8) and 256 pixels x 256 channels (WM = 2, WN = 2; N % 256 == 0 layers: every patch element is normalised ONCE per 256
//     output channels, a whole 16x16 image per tile);
//   * the in-loop normalisation (unpack, fma, exp2, add, rcp, mul, pack per element) is hand-placed: per channel PAIR nine MFMA gaps of at most
//     8 issue cycles each (two plain vector instructions or one transcendental: the guide's issue prices), no consumer of a transcendental in
//     the gap that produced it, and the table / patch reads of round r + 1 issued inside the tail of round r;
//   * ONE block-wide barrier per half-chunk (in tap 8, behind the last normalisation store): it publishes the normalised patch of half-chunk
//     h + 1 and frees the buffer of h for the raw patch of h + 2; the 1x1 shortcut tiles have two LDS buffers of their own (requested two steps ahead);
//   * no run-time branch inside the loop: every wave normalises NROUND pieces (the pad pieces read a clamped pixel and are never read back).
// Same K order, same normalisation arithmetic, one accumulation chain per output element: the convolution sums are the same bytes as k_conv_gn2's
// (tests/test_gpu_conv_gn.py compares them bit for bit); the GroupNorm partial sums of the OUTPUT are grouped by 512- / 256-pixel tiles.
#pragma once
#include "gemm_dma.h"

namespace ncsn {

template <int RES, int WM_, int WN_>
struct ConvGn3Cfg {
    static_assert(WM_ * WN_ == 4 && (RES == 32 || RES == 16), "four waves; 32x32 / 16x16 images");
    static constexpr int W = RES, WS = RES + 2, HW = RES * RES, NW = 4, THREADS = 256, KT = 32;
    static constexpr int WM = WM_, WN = WN_, TM = 8, TN = 8;
    static constexpr int BM_ = WM_ * 128, BN_ = WN_ * 128;
    static_assert(HW % BM_ == 0 && BM_ % W == 0, "a tile is whole image rows of one image");
    static constexpr int PR = BM_ / W + 2, PLAST = PR * WS;                     // image rows of a tile + the halo rows; patch rows that are ever read
    static constexpr int NPIECE = (PLAST + 15) / 16, NROUND = (NPIECE + NW - 1) / NW;
    static constexpr int PATCH_BYTES = NROUND * NW * 1024;                      // every wave owns NROUND 1-KiB pieces (the last ones may be padding)
    static constexpr int TAB_BYTES = 256;                                       // (scale 32 | shift 32) fp32 of a half-chunk
    static constexpr int PSW = BM_ / 16 / NW, SC_BYTES = BM_ * 64;              // shortcut tile [BM][32] bf16: pieces per wave, bytes
    static constexpr int OFF_TAB = 2 * PATCH_BYTES, OFF_SC = OFF_TAB + 2 * TAB_BYTES, LOOP_BYTES = OFF_SC + 2 * SC_BYTES;
    using Epi = EpiCfg<WM_, WN_, 8, 8, 163840>;
    static constexpr int LDS_BYTES = LOOP_BYTES > Epi::PACK_BYTES ? LOOP_BYTES : Epi::PACK_BYTES;
    static_assert(Epi::PACK_OK && LDS_BYTES <= 163840, "one block per CU");
    static constexpr int swz_key(int xx) { return (xx >> 1) & 2; }
    // A fragment of row-tile i (16 pixels) at dy = 0, relative to the lane's base: 32x32: row-tile i = image row i >> 1, columns (i & 1) * 16 ..; 16x16: image row i
    static constexpr int aoff(int i) { return (RES == 32 ? (i >> 1) * WS + (i & 1) * 16 : i * WS) * 64; }
    // ---- the schedule (slots of a 64-MFMA step) ----
    static constexpr int BAR = 40;                                              // tap 8 / shortcut steps: the block-wide barrier
    static constexpr int rd_a(int i) { return 10 + 2 * i; }                    // A fragment i of the next tap (taps 0..7)
    static constexpr int rd_a_late(int i) { return BAR + 2 + 2 * i; }          // ... of the step behind the barrier
    static constexpr int DMA_TAB = 26;
    static constexpr int dma_piece(int r) { return 28 + 3 * r; }               // tap 0: the raw patch of the next half-chunk, round r's piece
    static constexpr int dma_sc(int n) { return BAR + 1 + 2 * n; }             // shortcut steps: piece n of tile s + 2
    static_assert(dma_piece(NROUND - 1) < 64 && rd_a_late(7) < 64 && dma_sc(PSW - 1) < 64, "inside the step");
    static constexpr int NREQ = 2 + NROUND;                                     // requests a wave issues at tap 0 behind the weight loads of tap 1
    // normalisation rounds: global slot G = (T - 2) * 64 + S over taps 2..8; round r's four channel pairs occupy G in [gp(r), gp(r) + 36)
    static constexpr int G_END = 6 * 64 + BAR - 2;
    static constexpr int STRIDE = (G_END - 1 - 38 - 12) / (NROUND - 1) > 37 ? (G_END - 1 - 38 - 12) / (NROUND - 1) : 37;
    static constexpr int gp(int r) { return 12 + r * STRIDE; }
    static_assert(gp(NROUND - 1) + 38 < G_END, "the rounds end in front of the barrier of tap 8");
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- instruction wrappers (volatile: the order of the K loop is the order written) ----
__device__ __forceinline__ void cg3_mfma(f32x4& acc, const u32x4& b, const u32x4& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
__device__ __forceinline__ u32x4 cg3_gload16(unsigned voff, const void* sbase) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
__device__ __forceinline__ void cg3_glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// the (scale | shift) table of a half-chunk: scale by lanes 0-31, shift by lanes 32-63, 4 bytes per lane -> 256 consecutive bytes of LDS (two requests)
__device__ __forceinline__ void cg3_gtab(unsigned toff, const float* sc, const float* sh, unsigned lds_dst) {
    unsigned long long save;
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, 0xffffffff\n\t"
                 "global_load_lds_dword %1, %2\n\t"
                 "s_not_b64 exec, exec\n\t"
                 "global_load_lds_dword %1, %3\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(toff), "s"(sc), "s"(sh), "s"(lds_dst) : "memory", "m0", "scc");
}
template <int OFF> __device__ __forceinline__ void cg3_lds_write16(unsigned addr, const u32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int VM, int LGKM> __device__ __forceinline__ void cg3_wait() {
    if constexpr (LGKM >= 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)" :: "n"(VM), "n"(LGKM) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VM) : "memory");
}
__device__ __forceinline__ void cg3_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void cg3_barrier() { asm volatile("s_barrier" ::: "memory"); }

// everything a step needs, in registers (the struct is taken apart by SROA: every member is accessed with compile-time indices)
template <class Cfg>
struct CG3Ctx {
    u32x4 fa[2][8], fb[2][8];                 // A (pixels) / weight fragments of the step being multiplied and of the next one
    unsigned boff[8];                         // the lane's byte offset into weight fragment block j of a K step
    unsigned vo[Cfg::NROUND];                 // patch requests: the lane's source byte offset of its piece of round r
    unsigned scv[Cfg::PSW];                   // shortcut requests: the lane's source byte offset of piece n
    unsigned a_dx[3];                         // A fragment bases (dx = -1, 0, +1 at dy = -1) in patch buffer 0
    unsigned a_sc;                            // A fragment base in shortcut buffer 0
    unsigned n_addr[2];                       // the lane's 16 bytes of its piece of round 0 in patch buffer 0 / 1
    unsigned n_tab;                           // the lane's row (channel chunk lane & 3) of table 0
    unsigned nmask;                           // bit r: the pixel of round r lies inside the image; bit 16 + r: bit 2 of its patch column (the swizzle key)
    // normalisation rounds
    u32x4 nv, ns0, ns1, nh0, nh1, npk;
    unsigned ntb, nin[2], u0, u1, t0, t1;
    // wave-uniform
    const unsigned char* wnext;               // weight fragments of the NEXT K step
    const bf16* pnext;                        // raw patch source of the next half-chunk
    const float* tsc; const float* tsh;       // its table rows
    const bf16* scnext;                       // source of shortcut tile s + 2
    unsigned lds_patch, lds_tab, lds_sc, wave;
};

// ---- one channel pair of a normalisation round, gap g (0..8) ----
template <int P, int G, class Cfg>
__device__ __forceinline__ void cg3_pair_gap(CG3Ctx<Cfg>& c, int rpar) {
    // scale / shift of channels 2 P, 2 P + 1 of the lane's chunk
    if constexpr (G == 0) {
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(c.u0) : "v"(c.nv[P]));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(c.u1) : "v"(c.nv[P]));
    } else if constexpr (G == 1) {
        if constexpr (P < 2) {
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.t0) : "v"(c.u0), "v"(c.ns0[2 * P]), "v"(c.nh0[2 * P]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.t1) : "v"(c.u1), "v"(c.ns0[2 * P + 1]), "v"(c.nh0[2 * P + 1]));
        } else {
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.t0) : "v"(c.u0), "v"(c.ns1[2 * P - 4]), "v"(c.nh1[2 * P - 4]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.t1) : "v"(c.u1), "v"(c.ns1[2 * P - 3]), "v"(c.nh1[2 * P - 3]));
        }
    } else if constexpr (G == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(c.u0) : "v"(c.t0));
    else if constexpr (G == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(c.u1) : "v"(c.t1));
    else if constexpr (G == 4) {
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(c.u0));
        asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(c.u1));
    } else if constexpr (G == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(c.u0));
    else if constexpr (G == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(c.u1));
    else if constexpr (G == 7) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c.t0) : "v"(c.u0));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c.t1) : "v"(c.u1));
    } else {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c.npk[P]) : "v"(c.t0), "v"(c.t1));
        if (rpar) asm volatile("v_and_b32 %0, %0, %1" : "+v"(c.npk[P]) : "v"(c.nin[1]));
        else asm volatile("v_and_b32 %0, %0, %1" : "+v"(c.npk[P]) : "v"(c.nin[0]));
    }
}

// the normalisation work of global slot G (taps 2..8 of a half-chunk that has a successor): rounds R = 0 .. NROUND - 1 on patch buffer NB / table NB
template <class Cfg, int NB, int G, int R = 0>
__device__ __forceinline__ void cg3_norm_slot(CG3Ctx<Cfg>& c) {
    if constexpr (R < Cfg::NROUND) {
        constexpr int GP = Cfg::gp(R);
        // the loads of round R: round 0 at G = 1..4, round R > 0 inside the tail of round R - 1 (its last pair's unpack / fma are at GP' + 27 / + 28)
        constexpr int GL = R == 0 ? 1 : Cfg::gp(R - 1) + 29;
        if constexpr (G == GL) c.nv = lds_read16<R * Cfg::NW * 1024>(c.n_addr[NB]);
        if constexpr (G == GL + 1) {
            asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(c.ntb) : "v"(c.nmask), "n"(16 + R));
            asm volatile("v_lshlrev_b32 %0, 6, %0" : "+v"(c.ntb));
        }
        if constexpr (G == GL + 2) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c.ntb) : "v"(c.n_tab));
            asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(c.nin[R & 1]) : "v"(c.nmask), "n"(R));
            c.ns0 = lds_read16<NB * Cfg::TAB_BYTES>(c.ntb);
            c.ns1 = lds_read16<NB * Cfg::TAB_BYTES + 16>(c.ntb);
        }
        if constexpr (G == GL + 3) {
            c.nh0 = lds_read16<NB * Cfg::TAB_BYTES + 128>(c.ntb);
            c.nh1 = lds_read16<NB * Cfg::TAB_BYTES + 144>(c.ntb);
        }
        if constexpr (G == GP) cg3_wait_lgkm0();
        if constexpr (G >= GP && G < GP + 36) {
            constexpr int P = (G - GP) / 9, GG = (G - GP) % 9;
            cg3_pair_gap<P, GG>(c, R & 1);
        }
        if constexpr (G == GP + 36) cg3_lds_write16<R * Cfg::NW * 1024>(c.n_addr[NB], c.npk);
        cg3_norm_slot<Cfg, NB, G, R + 1>(c);
    }
}

// ---- the slots of one tap.  T: tap 0..8; HP: parity of the half-chunk (= its patch buffer; register set of the step = (T + HP) & 1);
// ---- NEXT: a further half-chunk follows (its raw patch is requested at tap 0 and normalised behind taps 2..8); S: the slot
template <class Cfg, int T, int HP, bool NEXT, int S>
struct CG3Tap {
    template <int I>
    static __device__ __forceinline__ void reads(CG3Ctx<Cfg>& c) {
        if constexpr (I < 8) {
            constexpr int P = (T + HP) & 1;
            if constexpr (T < 8) {
                if constexpr (S == Cfg::rd_a(I))
                    c.fa[P ^ 1][I] = lds_read16<HP * Cfg::PATCH_BYTES + Cfg::aoff(I) + ((T + 1) / 3) * Cfg::WS * 64>(c.a_dx[(T + 1) % 3]);
            } else if constexpr (S == Cfg::rd_a_late(I)) {
                if constexpr (NEXT) c.fa[P ^ 1][I] = lds_read16<(HP ^ 1) * Cfg::PATCH_BYTES + Cfg::aoff(I)>(c.a_dx[0]);
                else c.fa[P ^ 1][I] = lds_read16<I * 1024>(c.a_sc);               // the first shortcut tile (buffer 0); unused when there is none
            }
            reads<I + 1>(c);
        }
    }
    template <int R>
    static __device__ __forceinline__ void requests(CG3Ctx<Cfg>& c) {
        if constexpr (R < Cfg::NROUND) {
            if constexpr (S == Cfg::dma_piece(R))
                cg3_glds16(c.vo[R], c.pnext, c.lds_patch + (HP ^ 1) * Cfg::PATCH_BYTES + (R * Cfg::NW) * 1024 + c.wave * 1024);
            requests<R + 1>(c);
        }
    }
    static __device__ __forceinline__ void run(f32x4 (&acc)[8][8], CG3Ctx<Cfg>& c) {
        constexpr int P = (T + HP) & 1;
        if constexpr (S == 0) {
            // the weight fragments of this step have landed (tap 1: the requests of tap 0 stay in flight); so have its A fragments
            if constexpr (T == 1 && NEXT) cg3_wait<Cfg::NREQ, 0>(); else cg3_wait<0, 0>();
        }
        if constexpr (S >= 1 && S <= 8) c.fb[P ^ 1][S - 1] = cg3_gload16(c.boff[S - 1], c.wnext);
        reads<0>(c);
        if constexpr (T == 0 && NEXT) {
            if constexpr (S == Cfg::DMA_TAB) cg3_gtab((c.wave * 0u) + 0u + c.vo[0] * 0u + c.scv[0] * 0u + c.nin[0] * 0u + c.a_sc * 0u + c.n_tab * 0u + c.nmask * 0u + c.boff[0] * 0u + c.n_addr[0] * 0u + c.tabvo(), c.tsc, c.tsh, c.lds_tab + (HP ^ 1) * Cfg::TAB_BYTES);
            requests<0>(c);
        }
        if constexpr (T >= 2 && NEXT) cg3_norm_slot<Cfg, HP ^ 1, (T - 2) * 64 + S>(c);
        if constexpr (T == 8) {
            if constexpr (S == Cfg::BAR - 1) cg3_wait_lgkm0();                    // this wave's normalised pieces are written
            if constexpr (S == Cfg::BAR) cg3_barrier();                           // the other buffer is complete; nobody reads this one any more
        }
        constexpr int i = S >> 3, j = S & 7;
        cg3_mfma(acc[i][j], c.fb[P][j], c.fa[P][i]);
        if constexpr (S + 1 < 64) CG3Tap<Cfg, T, HP, NEXT, S + 1>::run(acc, c);
    }
};
