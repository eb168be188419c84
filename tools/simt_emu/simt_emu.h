// TEST INFRASTRUCTURE ONLY — a minimal SIMT emulator so the HIP kernel sources under
// markushgrapher_amd/csrc can be compiled with g++ (-DMG_EMU) and their index math / data layouts
// checked on a machine without a GPU.  It is never built into, loaded by, or reachable from the product
// (markushgrapher_amd/_lib.py only ever loads libmgrapher_hip.so and fails loudly without it).
//
// Model: one cooperative fiber per thread of a workgroup, workgroups run one after another.
// Wave = 64 consecutive threads.  Collectives (__syncthreads, shuffles, MFMA) are rendezvous points.
// global_load_lds is synchronous here, so the emulator cannot see missing-wait hazards.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <functional>
#include <algorithm>

namespace emu {
struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct Ctx {
    Dim3 tid, bid, bdim, gdim;
    int lin_tid;
};
extern Ctx* cur;
void launch(Dim3 grid, Dim3 block, size_t shmem, const std::function<void()>& body);
void syncthreads();
char* smem();
uint32_t shfl_u32(uint32_t v, int src_lane);
void mfma_32x32x16_bf16(const uint16_t* a8, const uint16_t* b8, float* c16);
void mfma_16x16x32_bf16(const uint16_t* a8, const uint16_t* b8, float* c4);
void mfma_16x16x16_bf16(const uint16_t* a4, const uint16_t* b4, float* c4);
void ds_read_tr16_b64(const void* lds_lane, uint16_t* out4);
void glds16(const void* gsrc_lane, void* lds_wave_base);
}  // namespace emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
typedef emu::Dim3 dim3;

struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

static inline void __syncthreads() { emu::syncthreads(); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __shfl_xor(float v, int m) { return __uint_as_float(emu::shfl_u32(__float_as_uint(v), (emu::cur->lin_tid & 63) ^ m)); }
static inline int __shfl_xor(int v, int m) { return (int)emu::shfl_u32((uint32_t)v, (emu::cur->lin_tid & 63) ^ m); }
static inline uint32_t __shfl_xor(uint32_t v, int m) { return emu::shfl_u32(v, (emu::cur->lin_tid & 63) ^ m); }
static inline float __shfl(float v, int src) { return __uint_as_float(emu::shfl_u32(__float_as_uint(v), src & 63)); }
static inline int __shfl(int v, int src) { return (int)emu::shfl_u32((uint32_t)v, src & 63); }
static inline uint32_t __shfl(uint32_t v, int src) { return emu::shfl_u32(v, src & 63); }
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline void __threadfence() {}
