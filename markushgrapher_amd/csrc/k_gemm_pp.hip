// bf16 MFMA GEMM for the encoder-sized shapes: persistent workgroups, ping-pong wave schedule (see the kernel's header comment).
// Operands, epilogues and results are those of gemm_xl_kernel (k_gemm.hip); gemm() dispatches here by variant.
#include "k_gemm_epi.h"

#include <atomic>

namespace mg {

// ---------------------------------------------------------------------------------------------------------
// large-M GEMM, third form: PERSISTENT workgroups with a PING-PONG wave schedule (round 4).
//
// What the two-stage kernel above leaves on the table (measured, profiles/r02_gemm_whatif.txt, r04 notes in DESIGN.md): its 8 waves
// run in lockstep - both waves of a SIMD read fragments at the same time and want the matrix pipe at the same time - and a
// workgroup's HBM-heavy epilogue, its first-stage latency and the next workgroup's start are all paid with the matrix pipe idle.
// Here:
//   * same block tile (64*TI x 256, 8 waves as 2 x 4, 32*TI x 64 per wave) and the same operand format, so the same epilogues;
//   * the two wave rows are two GROUPS (waves 0-3 / 4-7: one wave of each group per SIMD) that run the same phase program one
//     barrier apart: a phase = [LOAD: ds_read the 16-wide k-tile's TI + 2 fragments, issue this wave's copies of a later k-tile,
//     counted vmcnt] barrier [lgkmcnt(0); 2*TI MFMAs at raised priority] barrier.  While group A multiplies, group B loads, and
//     vice versa: the matrix pipe of every SIMD always has one wave feeding it (cdna_hip_programming.md, "8-phase" schedule);
//   * operands travel in a ring of 8 k-tile slots ((2 TI + 8) KiB each = the two 64-deep stages of the kernel above, cut in
//     four): the slot of k-tile g is refilled with k-tile g + 8 as soon as both groups have read it, i.e. the copies of k-tile
//     g + 6 are issued in phase g - 1.5 K-steps (about 3000 cycles) of lead instead of one, with the same LDS footprint;
//   * the workgroup is persistent (one per CU, tiles dealt round-robin inside XCD-contiguous ranges) and the k-tile stream runs
//     ACROSS tiles: the first six k-tiles of the next tile are in flight during the epilogue, and nothing is re-launched.
// Hazards (barrier numbers: group A runs phase g between barriers 2g and 2g+2, group B between 2g+1 and 2g+3):
//   RAW  a wave waits for ITS copies of k-tile g+1 before its first barrier of phase g (A: 2g+1, B: 2g+2); the reads of k-tile g+1
//        start after barrier 2g+2 (A) / 2g+3 (B): every copy has landed and a barrier lies in between;
//   WAR  the reads of k-tile h are complete before barrier 2h+2 (A) / 2h+3 (B); the slot is refilled with k-tile h+8 in phase h+2
//        = after barrier 2h+4 (A) / 2h+5 (B).
// Sums are accumulated in the same order as in gemm_xl_kernel (k ascending per accumulator): results are bit-identical.
// ---------------------------------------------------------------------------------------------------------
constexpr int GP_RING = 8, GP_AHEAD = 6, GP_MAXT = 48;          // ring slots, copy lead (k-tiles), tiles per workgroup at most
// (GP_MAXT = 56 - which would keep the FFN-wi of a 160-image call on this kernel - was measured in round 5: the per-head form <EPI_HEADS> then
//  compiles to code that runs 606 instead of 279 us per launch at the benchmark shape, encoder 43.2 against 35.3 ms per batch; 48 stays)
constexpr int GP_GAIN_MAX = 2048;                               // EPI_RESID_NORM: the next norm's gains are staged in LDS when N <= this

template <int N>
MG_DEV void wait_vmcnt_n() {
#ifndef MG_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
MG_DEV void set_prio_hi() {
#ifndef MG_EMU
    __builtin_amdgcn_s_setprio(1);
#endif
}
MG_DEV void set_prio_lo() {
#ifndef MG_EMU
    __builtin_amdgcn_s_setprio(0);
#endif
}

// XP != 0: timing experiments of the tools build only (MG_PP_EXP; WRONG results): 1 = no operand copies after the prologue, 2 = no
// epilogue, 4 = no fragment reads (registers keep their first contents), 8 = no barriers inside the K loop
template <int EPI, int TI, int XP = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs a) {
    MG_DYN_SMEM(smem);
    constexpr int XT = 2 * TI, FR = XT + 8;                     // fragments per k-tile slot: X row tiles, then 8 W row tiles
    constexpr int KT_BYTES = FR * TILE_BYTES;
    constexpr int NHI = FR - 16;                                // waves 0 .. NHI-1 copy three fragments per k-tile, the others two
    constexpr int BM = 64 * TI;
    static_assert(FR >= 16 && FR <= 24, "two or three copies per wave and k-tile");
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MG_EMU
    const int w = tid >> 6;
#else
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wr = w >> 2, wc = w & 3;
    const int nbn = (a.N + GX_N - 1) / GX_N;
    const int n_list = a.row_tiles ? *a.n_row_tiles : 0;
    const int mt32 = (a.M + 31) >> 5, nt32 = (a.N + 31) >> 5;
    const int rt_all = a.row_tiles ? n_list : mt32;             // 32-row tiles to compute: all of them, or this launch's part (pp_parts > 1)
    const int rt0 = a.pp_parts > 1 ? (int)((long long)a.pp_part * rt_all / a.pp_parts) : 0;
    const int nrt = (a.pp_parts > 1 ? (int)((long long)(a.pp_part + 1) * rt_all / a.pp_parts) : rt_all) - rt0;
    const int KT = a.K >> 4;                                    // 16-wide k-tiles per output tile (a multiple of GP_RING)
    const int G = gridDim.x, b = blockIdx.x;
    // Row blocks of BALANCED height: a persistent workgroup runs whole tiles, so the launch takes ceil(tiles / G) tile times whatever
    // the remainder (QKV at the benchmark's lengths: 1332 tiles of 10 row tiles on 256 workgroups = 5.2 -> 6 rounds).  The row tiles
    // are therefore cut into nbm blocks of floor / ceil(nrt / nbm) <= 2 TI row tiles with nbm the next count that makes nbm * nbn a
    // multiple of G (if that costs at most 30 % more blocks): 1536 tiles of 8.7 row tiles = 6 rounds of 0.87 the height.  A wave whose
    // share of a block is short of TI tiles skips the missing tiles' MFMAs and stores (their fragment slots are filled from a live tile).
    const int need = (nrt + XT - 1) / XT;
    int nbm = need;
    {
        int g = G, r = nbn;                                     // gcd(G, nbn)
        while (r) { const int t = g % r; g = r; r = t; }
        const int step = G / g;
        const int bal = (need + step - 1) / step * step;
        if (a.pp_balance && bal * 10 <= need * 13 && bal <= nrt) nbm = bal;
    }
    (void)BM;
    const int nblk = nbm * nbn;
    // tiles of this workgroup: t_first, t_first + t_step, ... < t_end  (XCD-contiguous ranges: block b runs on XCD b % 8)
    int t_first, t_step, t_end;
    if ((G & 7) == 0) {
        const int T8 = (nblk + 7) >> 3, xcd = b & 7;
        t_first = xcd * T8 + (b >> 3); t_step = G >> 3;
        t_end = (xcd + 1) * T8 < nblk ? (xcd + 1) * T8 : nblk;
    } else {
        t_first = b; t_step = G; t_end = nblk;
    }
    // Order of an XCD's tiles.  Linear (tile = bm * nbn + bn): the 32 workgroups of an XCD cover 32 / nbn row blocks x ALL nbn column tiles at a
    // time, i.e. the whole weight matrix passes through that L2 once per round (FFN-wi: 8 MB per round and XCD against 4 MB of L2).
    // Column groups (pp_colgroup = CG): the full row blocks of the XCD's range are walked CG column tiles at a time - 32 / CG row blocks x CG
    // columns per round, the group's weight slab (CG x 256 x K) stays in L2 over all the range's row blocks, the activation rows are
    // re-read nbn / CG times instead.  Same tiles, same arithmetic per tile: bit-identical.
    int cg_lo = 0, cg_rows = 0;
    const int CG = a.pp_colgroup;
    if (CG > 0 && (G & 7) == 0 && nbn % CG == 0 && nbn > CG) {
        const int T8 = (nblk + 7) >> 3, xcd = b & 7;
        const int r0 = xcd * T8, r1 = (xcd + 1) * T8 < nblk ? (xcd + 1) * T8 : nblk;
        cg_lo = (r0 + nbn - 1) / nbn * nbn;
        cg_rows = (r1 / nbn * nbn - cg_lo) / nbn;
        if (cg_rows < 0) cg_rows = 0;
    }
    auto tile_bm_bn = [&](int tile, int& bm, int& bn) {
        const int q = tile - cg_lo;
        if (cg_rows > 0 && q >= 0 && q < cg_rows * nbn) {
            const int per = cg_rows * CG, g = q / per, rem = q - g * per;
            bm = cg_lo / nbn + rem / CG; bn = g * CG + rem % CG;
        } else {
            bm = tile / nbn; bn = tile - bm * nbn;
        }
    };
    int n_my = t_first < t_end ? (t_end - t_first + t_step - 1) / t_step : 0;
    if (n_my > GP_MAXT) n_my = GP_MAXT;                         // (the launcher keeps tiles / workgroup <= GP_MAXT)
    if (n_my == 0) return;

    // row tiles of my output tiles: rtab[i][wr * TI + ii] = 32-row tile id (>= 0) of wave row wr's ii-th tile, or -1 - (a live tile to
    // read instead) where the block is short; rcnt[i][wr] = tiles of wave row wr.  Block bm = list entries [bm nrt / nbm, (bm+1) nrt / nbm),
    // the first half (rounded up) for wave row 0
    int* rtab = (int*)(smem + GP_RING * KT_BYTES);
    int* rcnt = rtab + GP_MAXT * XT;
    for (int e = tid; e < n_my * XT; e += 512) {
        const int i = e / XT, r = e - i * XT;
        int bm, bn_unused;
        tile_bm_bn(t_first + i * t_step, bm, bn_unused);
        const int e0 = (int)((long long)bm * nrt / nbm), e1 = (int)((long long)(bm + 1) * nrt / nbm);
        const int h = e1 - e0, c0 = (h + 1) >> 1;
        const int wrow = r / TI, ii = r - wrow * TI;
        const int cnt = wrow == 0 ? c0 : h - c0;
        const bool live = ii < cnt;
        const int idx = e0 + (live ? (wrow == 0 ? ii : c0 + ii) : 0);
        const int rt = a.row_tiles ? a.row_tiles[rt0 + idx] : rt0 + idx;
        rtab[e] = live ? rt : -1 - rt;
        if (ii == 0) rcnt[i * 2 + wrow] = cnt;
    }
    float* const gsm = (float*)(rcnt + GP_MAXT * 2);           // EPI_RESID_NORM: gain[0 .. N) (see resid_norm_epilogue_tiles)
    if constexpr (EPI == EPI_RESID_NORM) {                      // (the launcher keeps N <= GP_GAIN_MAX for this epilogue)
        if (a.gain)
            for (int e = tid * 4; e < a.N; e += 512 * 4) *(float4*)(gsm + e) = *(const float4*)(a.gain + e);
    }
    __syncthreads();

    const mg_lds_t sm0 = mg_lds_addr(smem);
    // this wave's copies of a k-tile: fragments w, w + 8 and (w < NHI) 16 + w.  A source = wave-uniform base (X or W) + wave-uniform
    // byte offset of the fragment's row tile (scalar registers) + 16 * lane; offsets fit 31 bits (operands < 2 GiB)
    const unsigned lane16 = lane * 16;
    const bool three = NHI > 0 && w < NHI;
    const char* const bX = (const char*)a.X;
    const char* const bW = (const char*)a.W;
    const bool k1_is_x = w + 8 < XT;                           // fragment w + 8 is an X row tile (TI = 5: waves 0, 1)
    auto frag_off = [&](int i, int k) -> unsigned {            // byte offset of my k-th fragment's row tile in tile i (wave-uniform)
        const int f = k < 2 ? w + 8 * k : 16 + w;
        int bm_unused, bn;
        tile_bm_bn(t_first + i * t_step, bm_unused, bn);
        int rt;
        if (f < XT) { const int v = rtab[i * XT + f]; rt = v >= 0 ? v : -1 - v; }
        else { rt = bn * 8 + (f - XT); rt = rt < nt32 - 1 ? rt : nt32 - 1; }
#ifndef MG_EMU
        rt = __builtin_amdgcn_readfirstlane(rt);
#endif
        return (unsigned)rt * (unsigned)(KT * TILE_BYTES);
    };
    unsigned oc[3], on[3];                                     // current / next tile
#pragma unroll
    for (int k = 0; k < 3; ++k) { oc[k] = frag_off(0, (k < 2 || three) ? k : 0); on[k] = n_my > 1 ? frag_off(1, (k < 2 || three) ? k : 0) : oc[k]; }
    auto issue1 = [&](int k, const unsigned (&off)[3], int kk, int slot) {          // my k-th copy of k-tile kk into ring slot `slot`
        const mg_lds_t dst = sm0 + slot * KT_BYTES;
        const unsigned ko = (unsigned)kk * TILE_BYTES;
        if (k == 0) glds16_async_sv(bX + (size_t)(off[0] + ko), lane16, dst + w * TILE_BYTES);
        else if (k == 1) glds16_async_sv((k1_is_x ? bX : bW) + (size_t)(off[1] + ko), lane16, dst + (w + 8) * TILE_BYTES);
        else if (NHI > 0 && three) glds16_async_sv(bW + (size_t)(off[2] + ko), lane16, dst + (16 + w) * TILE_BYTES);
    };
    auto issue = [&](const unsigned (&off)[3], int kk, int slot) {
        issue1(0, off, kk, slot); issue1(1, off, kk, slot); issue1(2, off, kk, slot);
    };
    // own copies of everything but the 5 most recent k-tiles have landed
    auto wait_groups5 = [&]() { if (three) wait_vmcnt_n<15>(); else wait_vmcnt_n<10>(); };
    auto wait_groups4 = [&]() { if (three) wait_vmcnt_n<12>(); else wait_vmcnt_n<8>(); };

    struct Frags { mg_raw16 x[TI], w[2]; };
    const mg_lds_t lx0 = sm0 + lane * 16 + wr * (TI * TILE_BYTES);
    const mg_lds_t lw0 = sm0 + lane * 16 + (XT + wc * 2) * TILE_BYTES;
    auto rd = [&](int slot, Frags& f) {
        const mg_lds_t lx = lx0 + slot * KT_BYTES, lw = lw0 + slot * KT_BYTES;
        lds_rd16_async<0>(f.w[0], lw);
        lds_rd16_async<TILE_BYTES>(f.w[1], lw);
        lds_rd16_async<0 * TILE_BYTES>(f.x[0], lx);
        lds_rd16_async<1 * TILE_BYTES>(f.x[1], lx);
        lds_rd16_async<2 * TILE_BYTES>(f.x[2], lx);
        lds_rd16_async<3 * TILE_BYTES>(f.x[3], lx);
        if constexpr (TI > 4) lds_rd16_async<4 * TILE_BYTES>(f.x[TI - 1], lx);
    };
    auto landed = [&](Frags& f) {
        MG_WAIT_LGKM_TIE(0, f.w[0]);
        MG_TIE(f.w[1]);
#pragma unroll
        for (int i = 0; i < TI; ++i) MG_TIE(f.x[i]);
    };

    // prologue: k-tiles 0 .. AHEAD-1 of the first tile
#pragma unroll
    for (int j = 0; j < GP_AHEAD; ++j) issue(oc, j, j);
    wait_groups5();                                            // k-tile 0
    MG_BARRIER_RAW();

    for (int i = 0; i < n_my; ++i) {
        const bool has_next = i + 1 < n_my;
        int bm, bn;
        tile_bm_bn(t_first + i * t_step, bm, bn);
        (void)bm;
        const int n0w = bn * GX_N + wc * 64;
        bool tor;
        if (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) tor = false;
        else if (EPI == EPI_HEADS) tor = !heads_region_is_T(a.heads, n0w < a.N ? n0w : 0);
        else tor = true;
        f32x16 acc[TI][2];
#pragma unroll
        for (int ii = 0; ii < TI; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[ii][j] = acc_zero();
#ifdef MG_EMU
        const int cnt = rcnt[i * 2 + wr];
#else
        const int cnt = __builtin_amdgcn_readfirstlane(rcnt[i * 2 + wr]);      // live row tiles of this wave (TI, or fewer in a short block)
#endif
        if (wr == 1) MG_BARRIER_RAW();                         // group B runs one barrier behind group A
        // the K loop exists once per operand order (tor is a property of the whole tile): one branch around the loop, not 64 inside it
        auto kloop = [&](auto tor_c) {
            constexpr bool TOR = decltype(tor_c)::value;
            Frags fkeep;
            // eight phases = k-tiles kb .. kb + 7 in ring slots 0 .. 7.  LAST: the tile's last eight k-tiles - the copies issued from
            // phase 2 on belong to the NEXT tile (k-tiles 0 .. 5), or to nothing when this was the workgroup's last tile
            auto phases = [&](int kb, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
                for (int p = 0; p < GP_RING; ++p) {
                    Frags f;
                    if (!(XP & 4) || (kb == 0 && p == 0 && !LAST)) rd(p, f); else { f = fkeep; }
                    constexpr int slot_mask = GP_RING - 1;
                    // what this phase copies: k-tile kb + p + AHEAD of this tile, k-tile p - 2 of the next one (LAST), or nothing
                    const bool own = !LAST || p < GP_RING - GP_AHEAD;
                    const bool any = own || has_next;
                    // own copies of the NEXT k-tile have landed: after this phase's copies at most 6 groups are in flight (5 when the
                    // wait comes before them, as here: the copies are issued among the MFMAs below)
                    if (any) wait_groups4(); else wait_vmcnt_n<0>();
                    if (!(XP & 8)) MG_BARRIER_RAW();
                    landed(f);
                    if (XP & 4) fkeep = f;
                    MG_SCHED_FENCE();
                    set_prio_hi();
                    {
                        uint4 xw[2], xx[TI];
#pragma unroll
                        for (int j = 0; j < 2; ++j) xw[j] = raw16_get(f.w[j]);
#pragma unroll
                        for (int ii = 0; ii < TI; ++ii) xx[ii] = raw16_get(f.x[ii]);
#pragma unroll
                        for (int ii = 0; ii < TI; ++ii) {
                            if (ii < cnt || ii < TI - 2) {         // (blocks are at least 2 TI - 3 tiles high wherever they are balanced: only the last two tiles of a wave can be missing)
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[ii][j] = TOR ? mfma32(xw[j], xx[ii], acc[ii][j]) : mfma32(xx[ii], xw[j], acc[ii][j]);
                            }
                            // the copies ride in the shadow of the matrix pipe: one after every second pair of MFMAs
                            if (ii < 3 && any && !(XP & 1)) {
                                MG_SCHED_FENCE();
                                if (own) issue1(ii, oc, kb + p + GP_AHEAD, (p + GP_AHEAD) & slot_mask);
                                else issue1(ii, on, p - (GP_RING - GP_AHEAD), (p + GP_AHEAD) & slot_mask);
                                MG_SCHED_FENCE();
                            }
                        }
                    }
                    set_prio_lo();
                    MG_SCHED_FENCE();
                    if (!(XP & 8)) MG_BARRIER_RAW();
                }
            };
            for (int kb = 0; kb < KT - GP_RING; kb += GP_RING) phases(kb, std::false_type{});
            phases(KT - GP_RING, std::true_type{});
        };
        if constexpr (EPI == EPI_F32_STORE || EPI == EPI_F32_RESID) kloop(std::false_type{});
        else if constexpr (EPI == EPI_HEADS) { if (tor) kloop(std::true_type{}); else kloop(std::false_type{}); }
        else kloop(std::true_type{});
        if (wr == 0) MG_BARRIER_RAW();                         // group A waits for group B's last phase: both groups store together
        int mrow[TI];
#pragma unroll
        for (int ii = 0; ii < TI; ++ii) {
            const int v = rtab[i * XT + wr * TI + ii];
            mrow[ii] = v >= 0 ? v * 32 : a.M;                  // past the end: row index M, every store is guarded by m < M
        }
        if (!(XP & 2) || acc[0][0][0] == 123456.789f) xl_epilogue<EPI, TI, 0, true>(a, acc, mrow, n0w, tor, lane, mg_lds_addr(gsm) + (n0w < a.N ? n0w : 0) * 4);
        wait_vmcnt_n<0>();            // stores and loads retire out of order with respect to each other: the counted waits of the next tile start from an empty queue
        // copy sources of the next two tiles, recomputed from the LDS table (nothing of them is live across the epilogue)
        if (has_next) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { oc[k] = frag_off(i + 1, (k < 2 || three) ? k : 0); on[k] = i + 2 < n_my ? frag_off(i + 2, (k < 2 || three) ? k : 0) : oc[k]; }
        }
    }
}
// row tiles of one problem over several launches: -1 = MG_PP_PARTS or 0; 0 never (fall back to the two-stage kernel: the default), 1 when the tile table is too small, 2 .. 8: always that many (tests)
static std::atomic<int> g_pp_parts{-1};
void gemm_pp_set_parts(int mode) { g_pp_parts = mode; }

template <int EPI, int TI>
static bool launch_pp(const GemmArgs& a_in, mgStream_t stream) {
    GemmArgs a = a_in;
    static int balance = -1;
    if (balance < 0) { const char* e = getenv("MG_PP_BALANCE"); balance = e ? atoi(e) : 1; }      // (A/B runs)
    a.pp_balance = balance;
    static int colgroup = -1;
    if (colgroup < 0) { const char* e = getenv("MG_PP_COLGROUP"); colgroup = e ? atoi(e) : 4; }      // (A/B runs; 0: linear order.  4 column tiles: FFN-wi 316 -> 299 us, encoder 35.9 -> 35.4 ms; 2: slower)
    a.pp_colgroup = colgroup;
    constexpr int BM = 64 * TI;
    const int nblk = ((a.M + BM - 1) / BM) * ((a.N + GX_N - 1) / GX_N);
    static int ncu = 0;
#ifndef MG_EMU
    if (!ncu) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256; }
#else
    ncu = 8;
#endif
    int G = nblk < ncu ? nblk : ncu;
    if (G >= 8) G &= ~7;
    if ((a.K & 127) != 0 || (EPI == EPI_RESID_NORM && a.N > GP_GAIN_MAX)) return false;   // (K/16 a multiple of the ring)
    // more tiles per workgroup than the kernel's tile table holds (balanced blocks are up to 30 % more; the FFN input projection of a 160-image
    // call: 10240 tiles): the row tiles in `parts` launches (every output tile is the same arithmetic whichever launch computes it)
    int parts = 1;
    int mode = g_pp_parts.load();
    if (mode < 0) { const char* e = getenv("MG_PP_PARTS"); mode = e ? atoi(e) : 0; g_pp_parts = mode; }      // (A/B runs)
    // Measured (profiles/r05_o_pp_parts_ab.txt, headline regime, same box, two runs each): FFN-wi of the 160-image calls as two launches of this
    // kernel against one launch of the two-stage kernel: encoder phase 190.3 against 189.1 ms per call, 147.1 against 146.8 images/s - no
    // difference, so the default stays 0 (such problems go to the two-stage kernel).
    while (parts < 8 && ((nblk + parts - 1) / parts + (nblk + parts - 1) / parts / 3 + G) > GP_MAXT * G) ++parts;
    if (parts > 1 && (!mode || parts >= 8)) return false;
    if (mode > parts && mode <= 8) parts = mode;                  // (tests: that many launches whatever the size)
    // the kernel addresses its operands by 32-bit byte offsets: an operand of 2 GiB or more goes to the two-stage kernel instead
    if ((size_t)((a.M + 31) / 32) * 32 * (size_t)a.K * 2 > 0x7fffffffull || (size_t)((a.N + 31) / 32) * 32 * (size_t)a.K * 2 > 0x7fffffffull) return false;
    const size_t sh = (size_t)GP_RING * (2 * TI + 8) * TILE_BYTES + (size_t)GP_MAXT * (2 * TI + 2) * sizeof(int) +
                      (EPI == EPI_RESID_NORM ? (size_t)GP_GAIN_MAX * sizeof(float) : 0);
    static bool once = false;
    if (!once) { MG_SET_MAX_SMEM((&gemm_pp_kernel<EPI, TI>), sh); once = true; }
#ifdef MG_TOOLS      // what-if variants with WRONG results: tools builds only
    if constexpr (EPI == EPI_PK || EPI == EPI_F32_RESID) {
        static int xp = -1;
        if (xp < 0) { const char* e = getenv("MG_PP_EXP"); xp = e ? atoi(e) : 0; }
        if (xp) {
#define MG_PX(N) case N: { static bool o = false; if (!o) { MG_SET_MAX_SMEM((&gemm_pp_kernel<EPI, TI, N>), sh); o = true; } \
                             MG_LAUNCH((gemm_pp_kernel<EPI, TI, N>), dim3(G), dim3(512), sh, stream, a); } break;
            switch (xp) { MG_PX(1) MG_PX(2) MG_PX(3) MG_PX(4) MG_PX(7) MG_PX(8) MG_PX(9) MG_PX(11) default: MG_PX(15) }
#undef MG_PX
            return true;
        }
    }
#endif
    a.pp_parts = parts;
    for (int p = 0; p < parts; ++p) {
        a.pp_part = p;
        MG_LAUNCH((gemm_pp_kernel<EPI, TI>), dim3(G), dim3(512), sh, stream, a);
    }
    return true;
}
template <int TI>
static bool launch_pp_epi(const GemmArgs& a, int epi, mgStream_t stream) {
    switch (epi) {
        case EPI_F32_STORE: return launch_pp<EPI_F32_STORE, TI>(a, stream);
        case EPI_F32_RESID: return launch_pp<EPI_F32_RESID, TI>(a, stream);
        case EPI_PK_RELU: return launch_pp<EPI_PK_RELU, TI>(a, stream);
        case EPI_PK_GELU: return launch_pp<EPI_PK_GELU, TI>(a, stream);
        case EPI_PK: return launch_pp<EPI_PK, TI>(a, stream);
        case EPI_PK_BIAS: return launch_pp<EPI_PK_BIAS, TI>(a, stream);
        case EPI_PK_GELU_ERF: return launch_pp<EPI_PK_GELU_ERF, TI>(a, stream);
        case EPI_RESID_NORM: return launch_pp<EPI_RESID_NORM, TI>(a, stream);
        default: return launch_pp<EPI_HEADS, TI>(a, stream);
    }
}

// Measured and rejected (profiles/r04_h_gemm_coresident_rejected.txt): the TI = 4 form compiled to <= 192 registers per lane
// (__attribute__((amdgpu_num_vgpr(96))): the attribute counts half of gfx950's unified register file) so that two of its waves leave a
// quarter of a SIMD's registers and 31 KiB of LDS to the decode kernels of the other batches in flight - 116.0-116.1 images/s against
// 116.2-116.5 for the forms that own their CU: the dispatcher does not turn the free room into throughput.
// Measured and rejected (profiles/r04_c_gemm_duo_rejected.txt; the kernel is in the history of this file): TWO persistent 4-wave
// workgroups per CU with (32 TI) x 256 tiles, meant to put one workgroup's epilogue under the other's K loop.  Bit-identical, but its
// smaller tiles move 1.44 x the operand bytes from L2 (the copies cost 86 us of the QKV projection's 283 against 46 of 239 here:
// L2 -> LDS bandwidth is the co-limit of these GEMMs), and two workgroups started together stay in step over the 2 - 6 tiles each
// gets: QKV 283 us, O 168, wi 372, wo 438 against 254 / 155 / 303 / 392 for this kernel.

bool gemm_pp(const GemmArgs& a, int epi, int ti, mgStream_t stream) {
    return ti == 4 ? launch_pp_epi<4>(a, epi, stream) : launch_pp_epi<5>(a, epi, stream);
}

}  // namespace mg
