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
//     (the segment with its A fragments loaded straight from global memory into four register sets -- no LDS, no barrier, two steps of lead -- was built and
//     measured 4-7 % SLOWER per launch than the LDS tiles: sixteen half-line loads per step and wave through the vector-memory return path; not kept);
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
    static constexpr int dma_sc(int n) { return BAR + 1 + 2 * n; }             // piece n of the shortcut tile two steps ahead
    static_assert(rd_a_late(7) < 64 && dma_sc(PSW - 1) < 64, "inside the step");
    // normalisation rounds: global slot G = (T - 2) * 64 + S over taps 2..8; the table rows of the half-chunk are read at G = 0, 1; round r's eight elements
    // occupy the 34 gaps from gp(r), its store follows at gp(r) + 34; its patch bytes are read at gl(r) (round r > 0: behind the unpack stage of round r - 1)
    static constexpr int RLEN = 34;
    static constexpr int G_END = 6 * 64 + BAR - 2;
    static constexpr int STRIDE = (G_END - 1 - RLEN - 2 - 12) / (NROUND - 1) > RLEN + 1 ? (G_END - 1 - RLEN - 2 - 12) / (NROUND - 1) : RLEN + 1;
    static constexpr int gp(int r) { return 12 + r * STRIDE; }
    static constexpr int gl(int r) { return r == 0 ? 2 : gp(r - 1) + 5; }
    static_assert(gp(NROUND - 1) + RLEN + 2 < G_END, "the rounds end in front of the barrier of tap 8");
    // The raw patch of the NEXT half-chunk arrives piece by piece, just in time: vector-memory operations retire in order, so every wait for a tap's weight
    // fragments (slot 0 of every tap) also waits for every request issued before those loads -- a request can stay in flight for two taps at most,
    // whatever its data is needed for.  Twelve requests at tap 0 (the first form) were therefore due ~1.4 taps later from EVERY block of the chip at once, and
    // the half-chunks whose 64-byte rows are the first touch of their 128-byte lines stalled 2-5k clocks at tap 2 (tile timeline: odd half-chunks 13-17k clocks,
    // even ones 11.3k).  Piece r is requested in tap td(r) = gl(r) / 64 (behind that tap's weight loads: it must have landed at the start of tap td(r) + 2, which
    // is where its round reads it at the earliest): three requests in flight per wave at most, spread over six taps.
    static constexpr int td(int r) { return gl(r) / 64; }
    static constexpr int nd(int t) { int n = t == 0 ? 2 : 0; for (int r = 0; r < NROUND; ++r) n += td(r) == t; return n; }      // requests issued in tap t (the table: two)
    static constexpr int DMA_TAB = 9;                                           // tap 0
    static constexpr int dma_piece(int r) { int k = td(r) == 0 ? 1 : 0; for (int q = 0; q < r; ++q) k += td(q) == td(r); return 9 + 2 * k; }
    static_assert(td(NROUND - 1) < 8 && dma_piece(NROUND - 1) < 40, "requests behind the weight loads, in front of the barrier");
};

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
    NATINF_M0_ASM_BEGIN
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
    NATINF_M0_ASM_END
}
// the (scale | shift) table of a half-chunk: scale by lanes 0-31, shift by lanes 32-63, 4 bytes per lane -> 256 consecutive bytes of LDS (two requests)
__device__ __forceinline__ void cg3_gtab(unsigned toff, const float* sc, const float* sh, unsigned lds_dst) {
    unsigned long long save;
    NATINF_M0_ASM_BEGIN
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "s_mov_b64 %0, exec\n\t"
                 "s_mov_b32 exec_lo, -1\n\t"
                 "s_mov_b32 exec_hi, 0\n\t"
                 "global_load_lds_dword %1, %2\n\t"
                 "s_not_b64 exec, exec\n\t"
                 "global_load_lds_dword %1, %3\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(toff), "s"(sc), "s"(sh), "s"(lds_dst) : "memory", "m0", "scc");
    NATINF_M0_ASM_END
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

// Development timeline (make EXTRA=-DNATINF_CG3_TIMELINE; tools/cg3_timeline.py): shader-clock stamps of wave 0 of blocks 0 and 300 at the section
// boundaries of a tile -- [0] start, [1] requests issued, [2] patch landed, [3] K loop starts, [4 + h] half-chunk h starts, [40] shortcut segment starts,
// [41] K loop done, [42] epilogue starts, [43] end.  The shipped kernels carry no stamps.
#ifdef NATINF_CG3_TIMELINE
#define NATINF_CG3_STAMP(i) do { if (g.dbg_ts && (blockIdx.x == 0 || blockIdx.x == 300) && threadIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g.dbg_ts[(blockIdx.x ? 64 : 0) + (i)] = t_; } } while (0)
#else
#define NATINF_CG3_STAMP(i) do { } while (0)
#endif

// everything a step needs, in registers (the struct is taken apart by SROA: every member is accessed with compile-time indices)
template <class Cfg>
struct CG3Ctx {
    u32x4 fa[2][8], fb[2][8];                 // A (pixels) / weight fragments of the step being multiplied and of the next one
    unsigned boff[8];                         // the lane's byte offset into weight fragment block j of a K step
    unsigned vo[Cfg::NROUND];                 // patch requests: the lane's source byte offset of its piece of round r
    unsigned scv[Cfg::PSW];                   // shortcut requests: the lane's source byte offset of piece n
    unsigned a_dx[3];                         // A fragment bases (dx = -1, 0, +1 at dy = -1) in patch buffer 0
    unsigned a_sc;                            // A fragment base in shortcut buffer 0
    // normalisation: the lane owns channel chunk (lane & 3) of patch row piece * 16 + (lane >> 2) -- the SAME eight channels in every round, so its rows of the
    // (scale | shift) table are read once per half-chunk; the chunk sits in slot (lane & 3) ^ key of the row (key: the swizzle, bit 2 of the patch column)
    unsigned nadr[Cfg::NROUND];               // the lane's 16 bytes of round r in patch buffer 0
    unsigned ninf[Cfg::NROUND];               // 1.0f where the pixel of round r lies inside the image, +inf outside: y = t * rcp(2^t + ninf) is then 0 (the zero padding)
    unsigned n_tab;                           // the lane's row of table 0
    unsigned tvo;                             // table request: (lane & 31) * 4
    // normalisation rounds
    u32x4 nv, ns0, ns1, nh0, nh1, npk;
    unsigned nx[8], ne[8];
    // wave-uniform
    const unsigned char* wnext;               // weight fragments of the NEXT K step
    const bf16* pnext;                        // raw patch source of the next half-chunk
    const float* tsc; const float* tsh;       // its table rows
    const bf16* scnext;                       // source of shortcut tile s + 2
    unsigned lds_patch, lds_tab, lds_sc, wave;
};

// ---- the normalisation of one round (eight elements per lane = four channel pairs), STAGE BY STAGE over 34 MFMA gaps: unpack x 8 (gaps 0-3), fma x 8
// ---- (4-7), exp2 x 8 (8-15), + 1 x 8 (16-19), rcp x 8 (20-27), mul x 8 (28-31), pack x 4 (32-33): two plain vector instructions or one
// ---- transcendental per gap (8 issue cycles beside the MFMA's 8).  Stage by stage, not element by element, because hipcc's hazard recogniser assumes
// ---- a forwarding hazard between ANY inline-asm definition of a vector register and its first reader and does not count inline-asm statements as
// ---- wait states: it puts an `s_nop 0` in front of the first statement that reads a register another statement wrote (unless a nop or a real
// ---- instruction lies between them).  A dependent chain per element costs one nop per link (26 per round); eight elements a stage cost one per stage (6).
template <int GG, int R, class Cfg>
__device__ __forceinline__ void cg3_norm_gap(CG3Ctx<Cfg>& c) {
    if constexpr (GG < 4) {                                     // unpack pair GG: x[2 GG] = low half << 16, x[2 GG + 1] = high half
        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(c.nx[2 * GG]) : "v"(c.nv[GG]));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(c.nx[2 * GG + 1]) : "v"(c.nv[GG]));
    } else if constexpr (GG < 8) {                              // t = x * scale + shift (scale / shift carry -log2 e: gn_folded)
        constexpr int P = GG - 4;
        if constexpr (P < 2) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c.nx[2 * P]) : "v"(c.ns0[2 * P]), "v"(c.nh0[2 * P]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c.nx[2 * P + 1]) : "v"(c.ns0[2 * P + 1]), "v"(c.nh0[2 * P + 1]));
        } else {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c.nx[2 * P]) : "v"(c.ns1[2 * P - 4]), "v"(c.nh1[2 * P - 4]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c.nx[2 * P + 1]) : "v"(c.ns1[2 * P - 3]), "v"(c.nh1[2 * P - 3]));
        }
    } else if constexpr (GG < 16) asm volatile("v_exp_f32 %0, %1" : "=v"(c.ne[GG - 8]) : "v"(c.nx[GG - 8]));
    else if constexpr (GG < 20) {                               // 2^t + 1 (+ inf for a pixel outside the image: its reciprocal, hence the element, is 0)
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(c.ne[2 * (GG - 16)]) : "v"(c.ninf[R]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(c.ne[2 * (GG - 16) + 1]) : "v"(c.ninf[R]));
    } else if constexpr (GG < 28) asm volatile("v_rcp_f32 %0, %0" : "+v"(c.ne[GG - 20]));
    else if constexpr (GG < 32) {                               // t / (1 + 2^t) = -log2 e * silu(v)
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c.nx[2 * (GG - 28)]) : "v"(c.ne[2 * (GG - 28)]));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(c.nx[2 * (GG - 28) + 1]) : "v"(c.ne[2 * (GG - 28) + 1]));
    } else {                                                    // round to bf16 (RNE)
        constexpr int P = 2 * (GG - 32);
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c.npk[P]) : "v"(c.nx[2 * P]), "v"(c.nx[2 * P + 1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c.npk[P + 1]) : "v"(c.nx[2 * P + 2]), "v"(c.nx[2 * P + 3]));
    }
}

// the normalisation work of global slot G (taps 2..8 of a half-chunk that has a successor): rounds R = 0 .. NROUND - 1 on patch buffer NB / table NB
template <class Cfg, int NB, int G, int R = 0>
__device__ __forceinline__ void cg3_norm_slot(CG3Ctx<Cfg>& c) {
    if constexpr (R == 0) {                                     // the lane's rows of the half-chunk's table, once
        if constexpr (G == 0) { c.ns0 = lds_read16<NB * Cfg::TAB_BYTES>(c.n_tab); c.ns1 = lds_read16<NB * Cfg::TAB_BYTES + 16>(c.n_tab); }
        if constexpr (G == 1) { c.nh0 = lds_read16<NB * Cfg::TAB_BYTES + 128>(c.n_tab); c.nh1 = lds_read16<NB * Cfg::TAB_BYTES + 144>(c.n_tab); }
    }
    if constexpr (R < Cfg::NROUND) {
        constexpr int GP = Cfg::gp(R);
        if constexpr (G == Cfg::gl(R)) c.nv = lds_read16<NB * Cfg::PATCH_BYTES>(c.nadr[R]);
        if constexpr (G == GP) cg3_wait_lgkm0();
        if constexpr (G >= GP && G < GP + Cfg::RLEN) cg3_norm_gap<G - GP, R>(c);
        if constexpr (G == GP + Cfg::RLEN) cg3_lds_write16<NB * Cfg::PATCH_BYTES>(c.nadr[R], c.npk);
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
            if constexpr (T == Cfg::td(R) && S == Cfg::dma_piece(R))
                cg3_glds16(c.vo[R], c.pnext, c.lds_patch + (HP ^ 1) * Cfg::PATCH_BYTES + (R * Cfg::NW) * 1024 + c.wave * 1024);
            requests<R + 1>(c);
        }
    }
    static __device__ __forceinline__ void run(f32x4 (&acc)[8][8], CG3Ctx<Cfg>& c) {
        constexpr int P = (T + HP) & 1;
        if constexpr (S == 0) {
            // the weight fragments of this step have landed (the requests issued behind them, in the tap before, stay in flight); so have its A fragments
            if constexpr (T >= 1 && NEXT) cg3_wait<Cfg::nd(T - 1), 0>(); else cg3_wait<0, 0>();
        }
        if constexpr (S >= 1 && S <= 8) c.fb[P ^ 1][S - 1] = cg3_gload16(c.boff[S - 1], c.wnext);
        reads<0>(c);
        if constexpr (NEXT) {
            if constexpr (T == 0 && S == Cfg::DMA_TAB) cg3_gtab(c.tvo, c.tsc, c.tsh, c.lds_tab + (HP ^ 1) * Cfg::TAB_BYTES);
            requests<0>(c);
        } else if constexpr (T == 8 && S > Cfg::BAR && (S - Cfg::BAR - 1) % 2 == 0 && (S - Cfg::BAR - 1) / 2 < Cfg::PSW) {
            // the last half-chunk: the SECOND shortcut tile -> shortcut buffer 1 (the first one came with the prologue; no shortcut segment: a dummy source)
            constexpr int n = (S - Cfg::BAR - 1) / 2;
            cg3_glds16(c.scv[n], c.scnext, c.lds_sc + Cfg::SC_BYTES + n * 1024 + c.wave * (Cfg::PSW * 1024));
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

// ---- one step of the 1x1 shortcut segment: tile s (register set P = s & 1) is multiplied.  Barrier 1 (slot 1): every wave holds its A fragments of tile s, so
// ---- tile s + 2 may be requested into that buffer right behind the weight loads (1.5 steps before its fragments are read instead of 1: a step is ~1,100 clocks and
// ---- every second tile is the first touch of its 128-byte lines); barrier 2 (slot BAR): every wave's pieces of tile s + 1 have landed -- its fragments are read behind it.
template <class Cfg, int P, int S>
struct CG3Sc {
    static __device__ __forceinline__ void run(f32x4 (&acc)[8][8], CG3Ctx<Cfg>& c) {
        if constexpr (S == 0) cg3_wait<Cfg::PSW, 0>();                              // the weights of step s have landed (the requests for tile s + 1, issued behind them, stay in flight); so have its A fragments
        if constexpr (S == 1) cg3_barrier();
        if constexpr (S >= 1 && S <= 8) c.fb[P ^ 1][S - 1] = cg3_gload16(c.boff[S - 1], c.wnext);
        if constexpr (S >= 9 && (S - 9) % 2 == 0 && (S - 9) / 2 < Cfg::PSW) {
            constexpr int n = (S - 9) / 2;
            cg3_glds16(c.scv[n], c.scnext, c.lds_sc + P * Cfg::SC_BYTES + n * 1024 + c.wave * (Cfg::PSW * 1024));
        }
        if constexpr (S == Cfg::BAR - 1) cg3_wait<8 + Cfg::PSW, -1>();              // this wave's pieces of tile s + 1 have landed (the weight loads and requests of this step stay in flight)
        if constexpr (S == Cfg::BAR) cg3_barrier();
        if constexpr (S > Cfg::BAR + 1 && (S - Cfg::BAR - 2) % 2 == 0 && (S - Cfg::BAR - 2) / 2 < 8) {
            constexpr int i = (S - Cfg::BAR - 2) / 2;
            c.fa[P ^ 1][i] = lds_read16<(P ^ 1) * Cfg::SC_BYTES + i * 1024>(c.a_sc);
        }
        constexpr int i = S >> 3, j = S & 7;
        cg3_mfma(acc[i][j], c.fb[P][j], c.fa[P][i]);
        if constexpr (S + 1 < 64) CG3Sc<Cfg, P, S + 1>::run(acc, c);
    }
};

// EPI: the packed epilogues of gemm_dma.h (1 plain, 2 + GroupNorm partials, 5 + bf16 residual, 6 both).  g.b_frag = k_pack_frag's output (conv_gn2.h).
template <int RES, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv_gn3(const GemmArgs g)
{
    using Cfg = ConvGn3Cfg<RES, WM, WN>;
    constexpr int W = Cfg::W, WS = Cfg::WS, HW = Cfg::HW, BM_ = Cfg::BM_, BN_ = Cfg::BN_, NROUND = Cfg::NROUND, PSW = Cfg::PSW, KT = Cfg::KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lds_poison();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    NATINF_CG3_STAMP(0);
    const int wm = wave / WN, wn = wave % WN;
    const int nN = g.N / BN_, nM = g.M / BM_;
    const int tile = xcd_remap(blockIdx.x, nM * nN);
    const int mt = tile / nN, nt = tile - mt * nN;
    const int m0 = mt * BM_, n0 = nt * BN_;
    const int b = m0 / HW, y0 = (m0 % HW) / W;                            // image and first image row of this tile
    const int ush = g.a0_up;                                              // 1: the source image has half the resolution (nearest up-sampling in the fetch)
    const bf16* const img = g.a0 + (int64_t)b * (HW >> (2 * ush)) * g.a0_ld;
    const float* const gsc = g.gn_scale + (int64_t)b * g.gn_ld;
    const float* const gsh = g.gn_shift + (int64_t)b * g.gn_ld;
    const int n_half = g.a0_C / KT, n_sc = g.a1 ? g.a1_C / KT : 0;       // both even (a0_C, a1_C multiples of 64)
    const int nk = 9 * n_half, NT = nk + n_sc;
    const unsigned char* const wfrag = reinterpret_cast<const unsigned char*>(g.b_frag);
    const int sup = g.a1_up;
    const bf16* const a1base = g.a1 ? g.a1 + (int64_t)(sup ? b * (HW >> 2) : m0) * g.a1_ld : nullptr;

    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    CG3Ctx<Cfg> c;
    c.wave = (unsigned)wave;
    c.lds_patch = (unsigned)(uintptr_t)((lds_u8*)smem);
    c.lds_tab = c.lds_patch + Cfg::OFF_TAB;
    c.lds_sc = c.lds_patch + Cfg::OFF_SC;
    const int prow = lane >> 2, pslot = lane & 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.boff[j] = (unsigned)(((n0 >> 4) + wn * 8 + j) * NT) * 1024u + (unsigned)lane * 16u;
    // patch requests: piece r * 4 + wave, patch row pp = piece * 16 + (lane >> 2) = pixel (y0 - 1 + yy, xx - 1), clamped (halo / pad rows: any readable pixel);
    // slot (lane & 3) of the row receives channel chunk (lane & 3) ^ key.  Normalisation: the lane owns chunk (lane & 3), i.e. slot (lane & 3) ^ key.
#pragma unroll
    for (int r = 0; r < NROUND; ++r) {
        const int pp = (r * 4 + wave) * 16 + prow;
        const int yy = pp / WS, xx = pp - yy * WS;
        const int y = min(max(y0 - 1 + yy, 0), RES - 1), x = min(max(xx - 1, 0), RES - 1);
        c.vo[r] = (unsigned)(((y >> ush) * (W >> ush) + (x >> ush)) * g.a0_ld + ((pslot ^ Cfg::swz_key(xx)) << 3)) * 2u;
        const bool inside = (unsigned)(y0 - 1 + yy) < (unsigned)RES && (unsigned)(xx - 1) < (unsigned)RES;
        c.ninf[r] = inside ? 0x3f800000u : 0x7f800000u;
        c.nadr[r] = c.lds_patch + (unsigned)((r * 4 + wave) * 1024 + prow * 64 + ((pslot ^ Cfg::swz_key(xx)) << 4));
    }
#pragma unroll
    for (int n = 0; n < PSW; ++n) {
        const int pp = (wave * PSW + n) * 16 + prow;
        const int src = sup ? ((y0 + pp / W) >> 1) * (W >> 1) + ((pp % W) >> 1) : pp;
        c.scv[n] = (unsigned)(src * g.a1_ld + ((pslot ^ ((pp >> 1) & 2)) << 3)) * 2u;
    }
    c.tvo = (unsigned)(lane & 31) * 4u;
    c.n_tab = c.lds_tab + ((unsigned)pslot << 5);
    {
        const int frow = lane & 15, fq = lane >> 4;
        const int ml = wm * 128 + frow;                                   // first pixel of this wave, the lane's row
        const int pc = ((ml / W) + 1) * WS + (ml % W) + 1, xc = ml % W;    // its patch row at the centre tap, and its patch column - 1
#pragma unroll
        for (int d = 0; d < 3; ++d) c.a_dx[d] = c.lds_patch + (unsigned)((pc - WS + d - 1) * 64 + ((fq ^ Cfg::swz_key(xc + d)) << 4));
        c.a_sc = c.lds_sc + (unsigned)(ml * 64 + ((fq ^ ((ml >> 1) & 2)) << 4));
    }

    // ---- prologue: the first shortcut tile, then table + raw patch of half-chunk 0 -> buffer 0 and the first step's weights ----
    if (n_sc > 0) {
#pragma unroll
        for (int n = 0; n < PSW; ++n) cg3_glds16(c.scv[n], a1base, c.lds_sc + n * 1024 + c.wave * (PSW * 1024));
    }
    cg3_gtab(c.tvo, gsc, gsh, c.lds_tab);
#pragma unroll
    for (int r = 0; r < NROUND; ++r) cg3_glds16(c.vo[r], img, c.lds_patch + (r * 4) * 1024 + c.wave * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) c.fb[0][j] = cg3_gload16(c.boff[j], wfrag);
    c.scnext = n_sc > 0 ? a1base + KT : reinterpret_cast<const bf16*>(wfrag);      // the second shortcut tile (requested in the last half-chunk); none: any readable address
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    NATINF_CG3_STAMP(1);
    // this wave's pieces of half-chunk 0, normalised in place as they land (requests retire in order: round r waits for everything up to its piece; no MFMAs
    // to hide behind yet, the arithmetic is left to hipcc to interleave)
    auto norm0 = [&](auto r_tag, const f32x4& s0, const f32x4& s1, const f32x4& h0, const f32x4& h1) __attribute__((always_inline)) {
        constexpr int R = decltype(r_tag)::value;
        cg3_wait<NROUND - 1 - R + 8, -1>();
        unsigned char* const pa = smem + (c.nadr[R] - c.lds_patch);
        const u32x4 nv = *reinterpret_cast<const u32x4*>(pa);
        const float ni = __uint_as_float(c.ninf[R]);
        u32x4 ou;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned w_ = nv[p];
            const float x0 = __uint_as_float(w_ << 16), x1 = __uint_as_float(w_ & 0xffff0000u);
            const float sa = p < 2 ? s0[2 * p] : s1[2 * p - 4], sb = p < 2 ? s0[2 * p + 1] : s1[2 * p - 3];
            const float ha = p < 2 ? h0[2 * p] : h1[2 * p - 4], hb = p < 2 ? h0[2 * p + 1] : h1[2 * p - 3];
            const float t0 = __builtin_fmaf(x0, sa, ha), t1 = __builtin_fmaf(x1, sb, hb);
            const float y0_ = t0 * __builtin_amdgcn_rcpf(ni + __builtin_amdgcn_exp2f(t0)), y1_ = t1 * __builtin_amdgcn_rcpf(ni + __builtin_amdgcn_exp2f(t1));
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bf16x2_t pr_ = {(bf16)y0_, (bf16)y1_};
            ou[p] = __builtin_bit_cast(unsigned, pr_);
        }
        *reinterpret_cast<u32x4*>(pa) = ou;
    };
    {
        cg3_wait<NROUND + 8, -1>();                                        // the table
        NATINF_CG3_STAMP(2);
        const unsigned char* const tp = smem + (c.n_tab - c.lds_patch);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(tp), s1 = *reinterpret_cast<const f32x4*>(tp + 16);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(tp + 128), h1 = *reinterpret_cast<const f32x4*>(tp + 144);
        using std::integral_constant;
        norm0(integral_constant<int, 0>{}, s0, s1, h0, h1); norm0(integral_constant<int, 1>{}, s0, s1, h0, h1); norm0(integral_constant<int, 2>{}, s0, s1, h0, h1);
        norm0(integral_constant<int, 3>{}, s0, s1, h0, h1); norm0(integral_constant<int, 4>{}, s0, s1, h0, h1); norm0(integral_constant<int, 5>{}, s0, s1, h0, h1);
        if constexpr (NROUND > 6) { norm0(integral_constant<int, 6>{}, s0, s1, h0, h1); norm0(integral_constant<int, 7>{}, s0, s1, h0, h1); }
        if constexpr (NROUND > 8) { norm0(integral_constant<int, 8>{}, s0, s1, h0, h1); norm0(integral_constant<int, 9>{}, s0, s1, h0, h1); }
        static_assert(NROUND == 6 || NROUND == 8 || NROUND == 10, "rounds of the prologue");
    }
    cg3_wait<0, -1>();                                                     // the first step's weights (and the first shortcut tile)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) c.fa[0][i] = *reinterpret_cast<const u32x4*>(smem + (c.a_dx[0] - c.lds_patch) + Cfg::aoff(i));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    NATINF_CG3_STAMP(3);
    using std::integral_constant;
    auto half_chunk = [&](auto hp_tag, auto next_tag, int h) __attribute__((always_inline)) {
        constexpr int HP = decltype(hp_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        c.pnext = img + (h + 1) * KT; c.tsc = gsc + (h + 1) * KT; c.tsh = gsh + (h + 1) * KT;
        const int kt = h * 9;
        NATINF_CG3_STAMP(4 + (h < 36 ? h : 35));
#define NATINF_CG3_TAP(T)                                                                                             \
        c.wnext = wfrag + (int64_t)((T < 8 || NEXT) ? kt + T + 1 : min(nk, NT - 1)) * 1024;                              \
        CG3Tap<Cfg, T, HP, NEXT, 0>::run(acc, c);
        NATINF_CG3_TAP(0) NATINF_CG3_TAP(1) NATINF_CG3_TAP(2) NATINF_CG3_TAP(3) NATINF_CG3_TAP(4)
        NATINF_CG3_TAP(5) NATINF_CG3_TAP(6) NATINF_CG3_TAP(7) NATINF_CG3_TAP(8)
#undef NATINF_CG3_TAP
    };
    for (int h = 0; h + 2 < n_half; h += 2) {
        half_chunk(integral_constant<int, 0>{}, std::true_type{}, h);
        half_chunk(integral_constant<int, 1>{}, std::true_type{}, h + 1);
    }
    half_chunk(integral_constant<int, 0>{}, std::true_type{}, n_half - 2);
    half_chunk(integral_constant<int, 1>{}, std::false_type{}, n_half - 1);
    NATINF_CG3_STAMP(40);
    // ---- 1x1 shortcut segment (nk is even: shortcut step s lives in register set s & 1, its tile in shortcut buffer s & 1) ----
    for (int s = 0; s < n_sc; s += 2) {
        c.wnext = wfrag + (int64_t)min(nk + s + 1, NT - 1) * 1024; c.scnext = a1base + min(s + 2, n_sc - 1) * KT;
        CG3Sc<Cfg, 0, 0>::run(acc, c);
        c.wnext = wfrag + (int64_t)min(nk + s + 2, NT - 1) * 1024; c.scnext = a1base + min(s + 3, n_sc - 1) * KT;
        CG3Sc<Cfg, 1, 0>::run(acc, c);
    }
    // The last step's look-ahead loads (clamped: nobody multiplies them) may still be in flight and, to hipcc, are DEAD: it may put the epilogue's first values into
    // their destination registers at once -- and the load then lands on top of them.  (Seen with a form of the shortcut segment that read its A fragments from global
    // memory: the residual epilogues, whose address arithmetic is hoisted up here, gave wrong tiles at B = 512 and right ones at B = 1.)  So: wait for everything,
    // THEN read every register a load may have been writing.
    cg3_wait<0, 0>();
#define NATINF_CG3_PIN8(a) asm volatile("" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
    NATINF_CG3_PIN8(c.fb[0]) NATINF_CG3_PIN8(c.fb[1]) NATINF_CG3_PIN8(c.fa[0]) NATINF_CG3_PIN8(c.fa[1])
#undef NATINF_CG3_PIN8
    NATINF_CG3_STAMP(41);
    // the epilogue's arguments are fetched from the kernel-argument segment HERE (conv_gn.h: kept in scalar registers across the K loop they end up spilled
    // into vector-register lanes) -- and IN FRONT of the drain below, pinned by an empty statement that reads them: hipcc otherwise issues the scalar loads
    // behind its ~190 accumulator reads and waits a full round trip for them with nothing else to do (one block per CU)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const unsigned __attribute__((address_space(4))) *kernarg_u32_t;
    kernarg_u32_t gp = (kernarg_u32_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(gp));
    GemmArgs ge;
    {
        unsigned* d = reinterpret_cast<unsigned*>(&ge);
#pragma unroll
        for (unsigned i = 0; i < sizeof(GemmArgs) / 4; ++i) d[i] = gp[i];
    }
    asm volatile("" :: "s"(ge.c), "s"(ge.bias_n), "s"(ge.rowvec), "s"(ge.resid), "s"(ge.gn_part), "s"(ge.c_ld), "s"(ge.rowvec_ld), "s"(ge.resid_ld), "s"(ge.gn_quads),
                 "s"(ge.log_rows_per_sample), "s"(ge.scale), "s"(ge.M), "s"(ge.N));
#else
    const GemmArgs ge = g;
#endif
    // every request / read of the last step's look-ahead has landed, the last MFMAs' results are written (inline asm: hipcc does not see the writes
    // it would pad for), and every wave is done with the tiles before the epilogue reuses them
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int tid_e = wave * 64 + lane_e;
    NATINF_CG3_STAMP(42);
    static_assert(EPI == 1 || EPI == 2 || EPI == 5 || EPI == 6, "packed epilogues: plain / + GroupNorm partials / + bf16 residual / both");
    // FIN (round 5): at 16x16 the 256 x 256 tile IS one sample with every channel, so its epilogue can write the consumer's GroupNorm table itself (GemmArgs::fin_*, when the plan
    // asked for it) -- no k_gn_finalize launch behind this one, and no cross-block protocol either: the "last-arriving block" of a 16x16 sample is its only block
    constexpr bool FIN16 = RES == 16 && WM == 2 && WN == 2 && (EPI == 2 || EPI == 6);
    packed_tile_epilogue<WM, WN, 8, 8, typename Cfg::Epi, ACT_NONE, EPI == 2 || EPI == 6, EPI == 5 || EPI == 6, false, false, 1, FIN16, true, true>(ge, smem, acc, m0, n0, 0, tid_e, lane_e, wm, wn);
    NATINF_CG3_STAMP(43);
}

}  // namespace ncsn
