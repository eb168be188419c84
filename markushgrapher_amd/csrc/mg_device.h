// Device-side vocabulary shared by all kernels: bf16 helpers, the MFMA fragment tile format, wave shuffles,
// direct global->LDS copies.  Target: gfx950 (MI355X, CDNA4) only.  With -DMG_EMU the same sources compile
// under g++ against tools/simt_emu (test infrastructure; never part of the product build).
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef MG_EMU
#include "simt_emu.h"
#define MG_DYN_SMEM(name) char* name = emu::smem()
#define MG_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch(grid, block, shmem, [=]() { kern(__VA_ARGS__); })
typedef void* mgStream_t;
struct f32x16 {
    float v[16];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
struct f32x4 {
    float v[4];
    float& operator[](int i) { return v[i]; }
    const float& operator[](int i) const { return v[i]; }
};
#else
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <unistd.h>
#define MG_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define MG_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
typedef hipStream_t mgStream_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mg_bf16x2 __attribute__((ext_vector_type(2)));
#endif

// host-side runtime shims (the emulator "device" is host memory)
#ifdef MG_EMU
#include <string.h>
#include <stdlib.h>
static inline int mg_memset_async(void* p, int v, size_t n, mgStream_t) { memset(p, v, n); return 0; }
static inline int mg_memcpy_async(void* d, const void* s, size_t n, mgStream_t) { memmove(d, s, n); return 0; }
static inline int mg_stream_sync(mgStream_t) { return 0; }
static inline int mg_peek_error() { return 0; }
static inline const char* mg_error_string(int) { return "emu"; }
typedef void* mgEvent_t;
static inline int mg_event_create(mgEvent_t* e) { *e = nullptr; return 0; }
static inline int mg_event_record(mgEvent_t, mgStream_t) { return 0; }
static inline float mg_event_elapsed_ms(mgEvent_t, mgEvent_t) { return 0.f; }
static inline void mg_event_destroy(mgEvent_t) {}
// second stream / ordering shims: the emulator executes every launch synchronously, so a "stream" is the null stream,
// every event has already happened and waiting is a no-op
static inline int mg_stream_create(mgStream_t* s, int, const uint32_t*, int) { *s = nullptr; return 0; }
static inline void mg_stream_destroy(mgStream_t) {}
static inline int mg_event_create_notiming(mgEvent_t* e) { *e = nullptr; return 0; }
static inline int mg_stream_wait_event(mgStream_t, mgEvent_t) { return 0; }
static inline int mg_event_done(mgEvent_t) { return 1; }
static inline int mg_event_sync(mgEvent_t) { return 0; }
static inline void* mg_host_alloc(size_t n) { return malloc(n); }
static inline void mg_host_free(void* p) { free(p); }
#else
static inline int mg_track(int rc, const char* what);
static inline int mg_memset_async(void* p, int v, size_t n, mgStream_t st) { return mg_track((int)hipMemsetAsync(p, v, n, st), "hipMemsetAsync"); }
static inline int mg_memcpy_async(void* d, const void* s, size_t n, mgStream_t st) {
    return mg_track((int)hipMemcpyAsync(d, s, n, hipMemcpyDefault, st), "hipMemcpyAsync");
}
// Waiting WITHOUT spinning: hipStreamSynchronize / hipEventSynchronize busy-wait on the host in this runtime (also for events created with
// hipEventBlockingSync: measured, tools/spin_probe.py) - one core per waiting thread, five per rank with four execution contexts in flight.
// Waits of this library poll the event and sleep in between (100 us: nothing here waits for less than a decode step).  MG_SPIN_SYNC=1
// restores the runtime's own waits.
static inline bool mg_spin_sync() { static const bool spin = [] { const char* e = getenv("MG_SPIN_SYNC"); return e && e[0] == '1'; }(); return spin; }
// An event poll returns hipErrorNotReady, which REPLACES whatever error an earlier launch of this thread left in the runtime's
// last-error slot: a failed launch would go unnoticed by the entry point's final check.  Anything pending is therefore taken out of the
// slot before polling and kept (mg_stashed_error) for mg_peek_error.
inline int& mg_stashed_error() { static thread_local int e = 0; return e; }
static inline void mg_stash_pending_error() {
    const hipError_t pre = hipGetLastError();
    if (pre != hipSuccess && pre != hipErrorNotReady && mg_stashed_error() == 0) mg_stashed_error() = (int)pre;
}
static inline int mg_event_wait_sleeping(hipEvent_t ev) {
    mg_stash_pending_error();
    for (int i = 0;; ++i) {
        const hipError_t r = hipEventQuery(ev);
        if (r == hipSuccess) return 0;
        if (r != hipErrorNotReady) return (int)r;
        (void)hipGetLastError();                       // (hipErrorNotReady is sticky in the thread's last-error slot)
        if (i >= 4) usleep(100);
    }
}
static inline int mg_stream_sync(mgStream_t st) {
    static thread_local hipEvent_t ev = nullptr;
    if (!mg_spin_sync() && !ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (mg_spin_sync() || !ev) return mg_track((int)hipStreamSynchronize(st), "hipStreamSynchronize");
    const int rc = mg_track((int)hipEventRecord(ev, st), "hipEventRecord");
    if (rc != 0) return rc;
    return mg_track(mg_event_wait_sleeping(ev), "hipEventQuery");
}
static inline int mg_peek_error() {
    int e = (int)hipGetLastError();
    if (e == (int)hipErrorNotReady) e = 0;
    if (e == 0) e = mg_stashed_error();
    mg_stashed_error() = 0;
    return e;
}
static inline const char* mg_error_string(int e) { return hipGetErrorString((hipError_t)e); }
typedef hipEvent_t mgEvent_t;
static inline int mg_event_create(mgEvent_t* e) { return (int)hipEventCreate(e); }
static inline int mg_event_record(mgEvent_t e, mgStream_t st) { return mg_track((int)hipEventRecord(e, st), "hipEventRecord"); }
static inline float mg_event_elapsed_ms(mgEvent_t a, mgEvent_t b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b); return ms; }
static inline void mg_event_destroy(mgEvent_t e) { (void)hipEventDestroy(e); }
// A second stream for work that overlaps the caller's stream.  low_priority: lowest stream priority (the dispatcher prefers the
// other streams' workgroups when both have some ready); cu_mask (nwords x 32 bits, nullable): restrict the stream to a subset of
// the compute units (hipExtStreamCreateWithCUMask).
static inline int mg_stream_create(mgStream_t* s, int low_priority, const uint32_t* cu_mask, int nwords) {
    if (cu_mask && nwords > 0) return (int)hipExtStreamCreateWithCUMask(s, (uint32_t)nwords, cu_mask);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = numerically greatest = lowest priority
    return (int)hipStreamCreateWithPriority(s, hipStreamNonBlocking, low_priority ? lo : hi);
}
static inline void mg_stream_destroy(mgStream_t s) { (void)hipStreamDestroy(s); }
static inline int mg_event_create_notiming(mgEvent_t* e) { return (int)hipEventCreateWithFlags(e, hipEventDisableTiming); }
static inline int mg_stream_wait_event(mgStream_t s, mgEvent_t e) { return mg_track((int)hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent"); }
static inline int mg_event_done(mgEvent_t e) { mg_stash_pending_error(); const hipError_t r = hipEventQuery(e); if (r == hipErrorNotReady) { (void)hipGetLastError(); return 0; } return 1; }
static inline int mg_event_sync(mgEvent_t e) { return mg_spin_sync() ? mg_track((int)hipEventSynchronize(e), "hipEventSynchronize") : mg_track(mg_event_wait_sleeping(e), "hipEventQuery"); }
static inline void* mg_host_alloc(size_t n) { void* p = nullptr; return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
static inline void mg_host_free(void* p) { (void)hipHostFree(p); }
#endif

// raw workgroup barrier / counted vector-memory wait: let direct global->LDS copies stay in flight ACROSS a barrier
// (a __syncthreads() would drain them with vmcnt(0)); see cdna_hip_programming.md "Pipelining across barriers".
#ifdef MG_EMU
#define MG_BARRIER_RAW() __syncthreads()
#define MG_WAIT_VMCNT(N) ((void)0)
#define MG_WAIT_LGKM0() ((void)0)
#define MG_SCHED_FENCE() ((void)0)
#define MG_SET_MAX_SMEM(kern, bytes) ((void)0)
#else
#define MG_BARRIER_RAW() __builtin_amdgcn_s_barrier()
#define MG_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define MG_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// the instruction scheduler moves nothing across this point (keeps a hand-written read / multiply interleave as written)
#define MG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MG_SET_MAX_SMEM(kern, bytes) (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
#endif

// Stream capture is serialised across host threads (execution contexts of the batches in flight capture their decode step on first
// use, possibly at the same moment): one capture + instantiate at a time, process-wide.  Captures are thread-local
// (hipStreamCaptureModeThreadLocal), so other threads keep launching on their own streams meanwhile.
#include <mutex>
inline std::mutex& mg_capture_mutex() { static std::mutex mu; return mu; }
// First failing runtime call of the current entry point on this thread (the runtime's own "last error" is per thread and sticky across
// callers: an error some other library left behind is drained at entry, MG_ENTRY, and is not ours to report).
struct MgErrSite { int code; const char* what; };
inline MgErrSite& mg_err_site() { static thread_local MgErrSite s{0, nullptr}; return s; }
static inline int mg_track(int rc, const char* what) { if (rc != 0 && mg_err_site().code == 0) mg_err_site() = MgErrSite{rc, what}; return rc; }

#define MG_DEV __device__ __forceinline__
#define MG_HD __host__ __device__ __forceinline__

namespace mg {

constexpr int WAVE = 64;
// Packed "fragment tile" format used for every MFMA operand kept in HBM (weights and activations):
// a matrix X[R][K] (bf16) is stored as [R/32][K/16] tiles of 1 KiB; inside a tile the order is
// [k-half (2)][row (32)][8 consecutive k] so that lane l of a wave reads its v_mfma_f32_32x32x16_bf16
// operand (row l%32, k = 8*(l/32)..+7) as ONE 16-byte load at tile_base + 16*l: a wave-load is 1 KiB
// contiguous, and a global_load_lds copy lands in LDS already in fragment order (conflict-free b128 reads).
constexpr int TILE_ELEMS = 512;
constexpr int TILE_BYTES = 1024;

MG_HD size_t pk_tile_off(int rt, int kt, int K) { return ((size_t)rt * (size_t)(K >> 4) + (size_t)kt) * TILE_ELEMS; }
MG_HD size_t pk_off(int r, int k, int K) {
    return pk_tile_off(r >> 5, k >> 4, K) + (size_t)(((k >> 3) & 1) * 256 + (r & 31) * 8 + (k & 7));
}
MG_HD size_t pk_elems(int R, int K) { return (size_t)((R + 31) / 32) * (size_t)(K / 16) * TILE_ELEMS; }

// fp32 residual stream of the encoder, tiled so that a lane owning one token row touches whole 16-byte groups:
// h[M][N] is stored as [M/32][N/4][32 rows][4 features]; the 32 lanes of a half-wave that own rows 32t..32t+31 read or
// write 512 contiguous bytes per feature group.
MG_HD size_t ht_off(int m, int n, int N) {
    return ((size_t)(m >> 5) * (size_t)(N >> 2) + (size_t)(n >> 2)) * 128 + (size_t)((m & 31) * 4 + (n & 3));
}

MG_DEV float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
MG_DEV float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
MG_DEV float bf16hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
// round-to-nearest-even fp32 -> bf16 (finite inputs)
MG_DEV uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
MG_DEV uint32_t pack_bf16(float lo, float hi);
// one value, same rounding as pack_bf16 (v_cvt_pk_bf16_f32 on the device)
MG_DEV uint16_t f32_to_bf16_rn(float f);
MG_DEV uint32_t pack_bf16(float lo, float hi) {
#ifdef MG_EMU
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#else
    // lowers to ONE v_cvt_pk_bf16_f32 (round-to-nearest-even), schedulable by the compiler unlike inline asm
    typedef float mg_f32x2 __attribute__((ext_vector_type(2)));
    const mg_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_bf16x2));
#endif
}

MG_DEV uint16_t f32_to_bf16_rn(float f) { return (uint16_t)(pack_bf16(f, 0.f) & 0xFFFFu); }

MG_DEV f32x16 acc_zero() {
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    return c;
}

// D = A·B + C for one 32x32x16 bf16 tile.  a: lane holds A[row = l%32][k = 8*(l/32)+0..7];
// b: lane holds B[k = 8*(l/32)+0..7][col = l%32];  result: lane holds D[row = (r%4)+8*(r/4)+4*(l/32)][col = l%32].
MG_DEV f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
#ifdef MG_EMU
    f32x16 d = c;
    emu::mfma_32x32x16_bf16((const uint16_t*)&a, (const uint16_t*)&b, d.v);
    return d;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mg_bf16x8, a), __builtin_bit_cast(mg_bf16x8, b),
                                                   c, 0, 0, 0);
#endif
}
// D = A·B + C for one 16x16x32 bf16 tile.  a: lane holds A[row = l%16][k = 8*(l/16)+0..7];
// b: lane holds B[k = 8*(l/16)+0..7][col = l%16];  result: lane holds D[row = 4*(l/16)+j][col = l%16], j = 0..3.
// In the packed fragment-tile format one operand = the 16-byte chunks (row, k-half) of TWO consecutive 16-wide k-tiles.
MG_DEV f32x4 mfma16(const uint4& a, const uint4& b, const f32x4& c) {
#ifdef MG_EMU
    f32x4 d = c;
    emu::mfma_16x16x32_bf16((const uint16_t*)&a, (const uint16_t*)&b, d.v);
    return d;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mg_bf16x8, a), __builtin_bit_cast(mg_bf16x8, b), c, 0, 0, 0);
#endif
}
// D = A·B + C for one 16x16x16 bf16 tile (v_mfma_f32_16x16x16_bf16).  a: lane holds A[row = l%16][k = 4*(l/16)+0..3];
// b: lane holds B[k = 4*(l/16)+0..3][col = l%16];  result: lane holds D[row = 4*(l/16)+j][col = l%16], j = 0..3 - the k order of
// the operands is the ROW order of a 16x16x32 result, so a transposed score tile feeds the second product without a lane exchange.
MG_DEV f32x4 mfma16k16(const uint2& a, const uint2& b, const f32x4& c) {
#ifdef MG_EMU
    f32x4 d = c;
    emu::mfma_16x16x16_bf16((const uint16_t*)&a, (const uint16_t*)&b, d.v);
    return d;
#else
    typedef short mg_s16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(mg_s16x4, a), __builtin_bit_cast(mg_s16x4, b), c, 0, 0, 0);
#endif
}
// ds_read_b64_tr_b16 (gfx950): every 16 lanes read one [4 rows][16 columns] block of 16-bit elements - lane 4a + b of the group
// supplies the 8-byte-aligned LDS address of row a, columns 4b .. 4b+3 - and receive it transposed: lane i of the group gets column i,
// rows 0 .. 3.  A row-major [key][feature] image thus yields the operand "feature l%16, keys 4*(l/16) .. +3" of mfma16k16.
MG_DEV uint2 lds_read_tr16(const void* lds_lane) {
#ifdef MG_EMU
    uint2 r;
    emu::ds_read_tr16_b64(lds_lane, (uint16_t*)&r);
    return r;
#else
    typedef __bf16 mg_bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) mg_bf16x4* mg_lds_bf16x4_p;
    const mg_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((mg_lds_bf16x4_p)lds_lane);
    return __builtin_bit_cast(uint2, v);
#endif
}
MG_DEV f32x4 acc4_zero() {
    f32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = 0.f;
    return c;
}
// row of D held in accumulator register r by this lane
MG_DEV int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// 16-byte direct global->LDS copy: every lane supplies its own source address, the destination is
// lds_wave_base + 16*lane (wave-uniform base).
MG_DEV void glds16(const void* gsrc_lane, void* lds_wave_base) {
#ifdef MG_EMU
    emu::glds16(gsrc_lane, lds_wave_base);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// Same copy, but issued from inline asm so that hipcc does not know about it: the compiler treats a builtin
// global_load_lds as a pending LDS write and inserts s_waitcnt vmcnt(0) before the next ds_read, which would drain a
// multi-stage pipeline every K-step.  The caller orders it by hand: MG_WAIT_VMCNT(N) then MG_BARRIER_RAW() before any
// wave reads the destination (cdna_hip_programming.md §5.7 "LDS-DMA recipe"; M0 is saved/restored in the statement).
MG_DEV void glds16_async(const void* gsrc_lane, void* lds_wave_base) {
#ifdef MG_EMU
    emu::glds16(gsrc_lane, lds_wave_base);
#else
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(dst) : "memory");
#endif
}

// LDS reads the compiler does not schedule or count (hand-pipelined fragment read-ahead: hipcc's IR passes sink ordinary LDS
// loads to their first use, which undoes "reads of k-tile kt+1 in front of the MFMAs of k-tile kt").  lds_rd16_async issues
// ds_read_b128 from the LDS byte address `addr` (+ the immediate OFF <= 65535); the destination must not be touched before
// MG_WAIT_LGKM_TIE(N, reg) / mg_tie(reg) covers it.  mg_lds_addr: the 32-bit LDS address of a __shared__ pointer.
#ifdef MG_EMU
MG_DEV const char* mg_lds_addr(const void* p) { return (const char*)p; }
typedef const char* mg_lds_t;
#else
typedef unsigned mg_lds_t;
MG_DEV mg_lds_t mg_lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
#endif
// the same copy with the destination given as an LDS address (mg_lds_addr): no generic-pointer conversion per copy
MG_DEV void glds16_async_lds(const void* gsrc_lane, mg_lds_t lds_wave_base) {
#ifdef MG_EMU
    emu::glds16(gsrc_lane, (void*)lds_wave_base);
#else
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(dst) : "memory");
#endif
}

// the same copy with a wave-uniform 64-bit source base (scalar registers) + a 32-bit per-lane byte offset: no per-lane 64-bit
// pointers to keep (global_load_lds_dwordx4 vaddr32, saddr64)
MG_DEV void glds16_async_sv(const char* src_uniform, unsigned lane_off, mg_lds_t lds_wave_base) {
#ifdef MG_EMU
    emu::glds16(src_uniform + lane_off, (void*)lds_wave_base);
#else
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src_uniform), "s"(dst) : "memory");
#endif
}

// the same with the non-temporal hint: data that one CU reads once (a decode stream) should not displace what L2 / MALL could keep
MG_DEV void glds16_async_sv_nt(const char* src_uniform, unsigned lane_off, mg_lds_t lds_wave_base) {
#ifdef MG_EMU
    emu::glds16(src_uniform + lane_off, (void*)lds_wave_base);
#else
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src_uniform), "s"(dst) : "memory");
#endif
}

// 16-byte global load into registers that the compiler does not track (no s_waitcnt inserted on its behalf, it does not
// count against the vmcnt the compiler computes for its own loads): for hand-pipelined prefetch several stages ahead.
// The destination must not be read, copied or moved before MG_WAIT_VMCNT_TIE(N, regs...) has covered the load.
#ifdef MG_EMU
typedef uint4 mg_raw16;
MG_DEV void gld16_async(mg_raw16& dst, const void* p) { dst = *(const uint4*)p; }
template <int OFF> MG_DEV void gld16_async_off(mg_raw16& dst, const void* p) { dst = *(const uint4*)((const char*)p + OFF); }
MG_DEV uint4 raw16_get(const mg_raw16& r) { return r; }
template <int OFF> MG_DEV void lds_rd16_async(mg_raw16& dst, mg_lds_t addr) { dst = *(const uint4*)(addr + OFF); }
#define MG_WAIT_LGKM_TIE(N, r) ((void)0)
#define MG_TIE(r) ((void)0)
#define MG_WAIT_VMCNT_TIE4(N, a, b, c, d) ((void)0)
#else
typedef unsigned int mg_raw16 __attribute__((ext_vector_type(4)));
MG_DEV void gld16_async(mg_raw16& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
// the same with an immediate byte offset (-4096 .. 4095): several loads off one address register pair
template <int OFF> MG_DEV void gld16_async_off(mg_raw16& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
MG_DEV uint4 raw16_get(const mg_raw16& r) { return make_uint4(r.x, r.y, r.z, r.w); }
template <int OFF> MG_DEV void lds_rd16_async(mg_raw16& dst, mg_lds_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
#define MG_WAIT_LGKM_TIE(N, r) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(r) :: "memory")
#define MG_TIE(r) asm volatile("" : "+v"(r))
// counted wait that the four registers depend on: their first use cannot be scheduled above it
#define MG_WAIT_VMCNT_TIE4(N, a, b, c, d) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")
#endif

MG_DEV uint4 ld16(const void* p) { return *(const uint4*)p; }
// streamed-once data (decode K/V, decode weights): non-temporal load, does not displace reusable lines
MG_DEV uint4 ld16_stream(const void* p) {
#ifdef MG_EMU
    return *(const uint4*)p;
#else
    typedef unsigned int mg_u32x4 __attribute__((ext_vector_type(4)));
    const mg_u32x4 v = __builtin_nontemporal_load((const mg_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}
MG_DEV void st16(void* p, const uint4& v) { *(uint4*)p = v; }
// written now, read a decode step later at the earliest (cache appends): non-temporal store, does not displace reusable lines
MG_DEV void st16_stream(void* p, const uint4& v) {
#ifdef MG_EMU
    *(uint4*)p = v;
#else
    typedef unsigned int mg_u32x4 __attribute__((ext_vector_type(4)));
    const mg_u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (mg_u32x4*)p);
#endif
}

// two packed bf16 pairs dotted into an fp32 accumulator
MG_DEV float dot2_bf16(uint32_t a, uint32_t b, float acc) {
#ifdef MG_EMU
    return acc + bf16lo(a) * bf16lo(b) + bf16hi(a) * bf16hi(b);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mg_bf16x2, a), __builtin_bit_cast(mg_bf16x2, b), acc, false);
#endif
}

MG_DEV float fast_exp(float x) {
#ifdef MG_EMU
    return expf(x);
#else
    return __expf(x);
#endif
}

MG_DEV float fast_exp2(float x) {
#ifdef MG_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);       // v_exp_f32 (arguments below -126 flush to 0, which is what a masked score wants)
#endif
}
// true in every lane if the predicate holds in any lane of the wave
MG_DEV bool wave_any(bool p) {
#ifdef MG_EMU
    int v = p ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m);
    return v != 0;
#else
    return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
}

// sum over each aligned group of 8 lanes, result in all 8 (DPP: quad_perm xor 1, quad_perm xor 2, row_half_mirror —
// pure VALU, no LDS crossbar traffic unlike ds_bpermute-based shuffles)
MG_DEV float sum8(float p) {
#ifdef MG_EMU
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    p += __shfl_xor(p, 4);
    return p;
#else
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0xB1, 0xF, 0xF, false));
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x4E, 0xF, 0xF, false));
    p += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x141, 0xF, 0xF, false));
    return p;
#endif
}

// ---- lane exchanges without the LDS crossbar (ds_bpermute costs an address register and ~100 cycles of latency per value) ----
// value of the lane whose index differs in bit STEP (8: DPP row_ror:8; 16 / 32: v_permlane16_swap / v_permlane32_swap, gfx950)
// v_permlane16_swap / v_permlane32_swap exchange the upper rows (16 lanes) / upper half of `a` with the lower rows / lower half
// of `b`, in place.  Inline asm, not the builtin: hipcc 7.2 folds the builtin's second result into the first when both are
// consumed as floats (tools/lane_probe.hip shows it: the halving step came back as own + own).  The two wait states the
// hardware wants between a VALU write of an operand and the swap are inside the string.
template <int STEP>
MG_DEV void lane_swap(unsigned& a, unsigned& b) {
#ifndef MG_EMU
    if constexpr (STEP == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
#endif
}
template <int STEP>
MG_DEV float lane_xor(float v, int lane) {
    static_assert(STEP == 8 || STEP == 16 || STEP == 32, "lane_xor: 8, 16 or 32");
#ifdef MG_EMU
    (void)lane;
    return __shfl_xor(v, STEP);
#else
    if constexpr (STEP == 8) {
        (void)lane;
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
    } else {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        lane_swap<STEP>(a, b);                    // a = {own | partner's}, b = {partner's | own} for the (clear | set) lanes
        return __builtin_bit_cast(float, (lane & STEP) ? a : b);
    }
#endif
}
// One step of a halving reduction over lane bit STEP.  Lanes with the bit clear keep `lo`, lanes with it set keep `hi`; the
// result is own_kept * f_own + partner's_same_part * f_oth.  The swap instructions move exactly the two halves that have to
// travel, so a register pair costs ONE instruction and no select.
template <int STEP>
MG_DEV float lane_halve(float lo, float hi, float f_own, float f_oth, int lane) {
    const bool up = (lane & STEP) != 0;
#ifdef MG_EMU
    const float recv = __shfl_xor(up ? lo : hi, STEP);
    return (up ? hi : lo) * f_own + recv * f_oth;
#else
    if constexpr (STEP == 8) {
        const float recv = lane_xor<8>(up ? lo : hi, lane);
        return (up ? hi : lo) * f_own + recv * f_oth;
    } else {
        unsigned a = __builtin_bit_cast(unsigned, lo), b = __builtin_bit_cast(unsigned, hi);
        lane_swap<STEP>(a, b);                    // a = {own lo | partner's hi}, b = {partner's lo | own hi}
        return __builtin_bit_cast(float, a) * (up ? f_oth : f_own) + __builtin_bit_cast(float, b) * (up ? f_own : f_oth);
    }
#endif
}
// sum over the 8 key-slot groups (lane bits 3..5), result in all lanes
MG_DEV float sum_slots(float v, int lane) {
    v += lane_xor<8>(v, lane);
    v += lane_xor<16>(v, lane);
    v += lane_xor<32>(v, lane);
    return v;
}

MG_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
MG_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// Turn one 32x32 accumulator tile into 16-byte bf16 chunks of 8 consecutive ROW indices for this lane's column.
// In: v[r] = value at D[acc_row(r, half)][lane%32].  Out: chunk[q] (q = 0,1) holds rows 16q + 8*half .. +7 of
// column lane%32 as 8 packed bf16.  (The halves hold interleaved groups of 4 rows; one exchange with lane^32
// completes each group of 8.)
struct PackedAcc { uint32_t p[4][2]; };   // p[g] = rows 8g + 4*half + {0,1},{2,3}
MG_DEV PackedAcc acc_pack(const f32x16& v) {
    PackedAcc o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        o.p[g][0] = pack_bf16(v[4 * g + 0], v[4 * g + 1]);
        o.p[g][1] = pack_bf16(v[4 * g + 2], v[4 * g + 3]);
    }
    return o;
}
MG_DEV void packed_to_chunks(const PackedAcc& a, int half, uint4 chunk[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uint32_t s0 = half ? a.p[2 * q][0] : a.p[2 * q + 1][0];
        uint32_t s1 = half ? a.p[2 * q][1] : a.p[2 * q + 1][1];
        uint32_t r0 = __shfl_xor(s0, 32);
        uint32_t r1 = __shfl_xor(s1, 32);
        if (half == 0) chunk[q] = make_uint4(a.p[2 * q][0], a.p[2 * q][1], r0, r1);
        else chunk[q] = make_uint4(r0, r1, a.p[2 * q + 1][0], a.p[2 * q + 1][1]);
    }
}
// The same exchange on v_permlane32_swap (gfx950): one instruction swaps a register of the upper half-wave with another
// register of the lower half-wave, in place - no LDS crossbar (ds_bpermute), no selects.  With G0 = a.p[2q], G1 = a.p[2q+1]:
// swap(vdst = G0[i], src0 = G1[i]) leaves G0 = {own G0 | lower's G1} and G1 = {upper's G0 | own G1} for the (lower | upper)
// lanes, i.e. chunk q = [G0[0], G0[1], G1[0], G1[1]] = rows 16q + 8*half .. +7 in both halves.
MG_DEV void packed_to_chunks_swap(const PackedAcc& a, int half, uint4 chunk[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uint32_t g0[2], g1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#ifdef MG_EMU
            const uint32_t got = __shfl_xor(half ? a.p[2 * q][i] : a.p[2 * q + 1][i], 32);
            g0[i] = half ? got : a.p[2 * q][i];
            g1[i] = half ? a.p[2 * q + 1][i] : got;
#else
            typedef unsigned int mg_u32x2 __attribute__((ext_vector_type(2)));
            const mg_u32x2 r = __builtin_amdgcn_permlane32_swap(a.p[2 * q][i], a.p[2 * q + 1][i], false, false);
            g0[i] = r.x; g1[i] = r.y;
#endif
        }
        chunk[q] = make_uint4(g0[0], g0[1], g1[0], g1[1]);
    }
}
MG_DEV void acc_to_chunks(const f32x16& v, int half, uint4 chunk[2]) {
    const PackedAcc a = acc_pack(v);
    packed_to_chunks(a, half, chunk);
}

}  // namespace mg
