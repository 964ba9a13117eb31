// TEST INFRASTRUCTURE ONLY — never part of the shipped library.
//
// A small lock-step host emulator for the HIP device code in dags_vae_search_amd/csrc, so that the
// kernels' lane/fragment indexing (64-wide waves, MFMA 16x16x4 f32 operand maps, cross-lane shuffles,
// LDS staging, workgroup barriers) can be checked against the oracle on the CPU-only build container
// — and under gdb — before any GPU time is spent.  It is compiled ONLY by tests/emu/build.py into
// tests/emu/_build/libdvs_emu.so and loaded ONLY by tests (-m "not gpu").  The product
// (dags_vae_search_amd) never references it and has no CPU path.
//
// Model: every thread of a workgroup is a fiber; a workgroup runs on one OS thread, round-robin over its
// fibers.  A wave collective (shuffle, MFMA) is "deposit operand, yield, read peers' operands": one full
// round-robin sweep separates deposit from read, so all 64 lanes have deposited.  Two deposit buffers per
// wave alternate, which is enough because lanes can never be more than one collective apart.  All lanes of
// a wave must execute the same sequence of collectives (true for well-formed wave code).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define DVS_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

struct WaveBuf {
    uint32_t a[2][WAVE];
    uint32_t b[2][WAVE];
    uint32_t a4[2][WAVE][4];      // 8 x bf16 operands of the bf16 MFMA
    uint32_t b4[2][WAVE][4];
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    dim3 tid;
    int lane = 0, wave = 0;
    int parity = 0;
    bool done = false;
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveBuf> waves;
    void* sched_sp = nullptr;
    int nthreads = 0;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    std::function<void()> body;
    char* dyn_smem = nullptr;
};

extern thread_local Block* g_block;
extern thread_local Fiber* g_cur;
extern thread_local dim3 g_blockIdx, g_blockDim, g_gridDim;

extern "C" void dvs_emu_switch(void** save_sp, void* load_sp);

inline void yield() { dvs_emu_switch(&g_cur->sp, g_block->sched_sp); }

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t dyn_smem_bytes);

template <class T> inline uint32_t bits(T v) { uint32_t u; static_assert(sizeof(T) == 4, "32-bit only"); memcpy(&u, &v, 4); return u; }
template <class T> inline T from_bits(uint32_t u) { T v; memcpy(&v, &u, 4); return v; }

// deposit one 32-bit value per lane, sweep, then fetch lane `src`'s deposit
template <class T> inline T exchange(T v, int src) {
    Fiber* f = g_cur;
    WaveBuf& w = g_block->waves[f->wave];
    const int p = f->parity;
    w.a[p][f->lane] = bits(v);
    yield();
    f->parity ^= 1;
    return from_bits<T>(w.a[p][src & (WAVE - 1)]);
}

}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define warpSize 64

// ---- cross-lane ------------------------------------------------------------------------------------
template <class T> inline T __shfl(T v, int src, int width = 64) {
    const int lane = emu::g_cur->lane;
    const int base = lane & ~(width - 1);
    return emu::exchange(v, base + (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = emu::g_cur->lane;
    int src = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu::exchange(v, src);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = emu::g_cur->lane;
    int src = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return emu::exchange(v, src);
}
inline unsigned long long __ballot(int pred) {
    emu::Fiber* f = emu::g_cur;
    emu::WaveBuf& w = emu::g_block->waves[f->wave];
    const int p = f->parity;
    w.a[p][f->lane] = pred ? 1u : 0u;
    emu::yield();
    f->parity ^= 1;
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m |= (unsigned long long)(w.a[p][l] & 1u) << l;
    return m;
}
inline int __builtin_amdgcn_readfirstlane(int v) { return emu::exchange(v, 0); }

// ---- MFMA v_mfma_f32_16x16x4_f32: D = A(16x4) * B(4x16) + C, lane l: A[l&15][l>>4], B[l>>4][l&15],
//      C/D[4*(l>>4)+reg][l&15]; accumulation is a k-ordered fmaf chain (guide §3 'FP32-input MFMA').
typedef float dvs_f32x4 __attribute__((ext_vector_type(4)));
inline dvs_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, dvs_f32x4 c, int, int, int) {
    emu::Fiber* f = emu::g_cur;
    emu::WaveBuf& w = emu::g_block->waves[f->wave];
    const int p = f->parity;
    w.a[p][f->lane] = emu::bits(a);
    w.b[p][f->lane] = emu::bits(b);
    emu::yield();
    f->parity ^= 1;
    const int col = f->lane & 15, g = f->lane >> 4;
    dvs_f32x4 d = c;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = 4 * g + reg;
        float acc = c[reg];
        for (int k = 0; k < 4; ++k)
            acc = fmaf(emu::from_bits<float>(w.a[p][k * 16 + row]), emu::from_bits<float>(w.b[p][k * 16 + col]), acc);
        d[reg] = acc;
    }
    return d;
}

// ---- MFMA v_mfma_f32_16x16x32_bf16: lane l supplies A[row l&15][k = 8*(l>>4) .. +7] and B[k = 8*(l>>4) .. +7][col l&15];
//      C/D as above.  bf16 products are exact in fp32; accumulated here as a k-ordered fp32 chain.
typedef __bf16 dvs_emu_bf8 __attribute__((ext_vector_type(8)));
inline dvs_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(dvs_emu_bf8 a, dvs_emu_bf8 b, dvs_f32x4 c, int, int, int) {
    emu::Fiber* f = emu::g_cur;
    emu::WaveBuf& w = emu::g_block->waves[f->wave];
    const int p = f->parity;
    memcpy(w.a4[p][f->lane], &a, 16);
    memcpy(w.b4[p][f->lane], &b, 16);
    emu::yield();
    f->parity ^= 1;
    const int col = f->lane & 15, g = f->lane >> 4;
    dvs_f32x4 d = c;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = 4 * g + reg;
        float acc = c[reg];
        for (int kb = 0; kb < 4; ++kb) {
            dvs_emu_bf8 va, vb;
            memcpy(&va, w.a4[p][kb * 16 + row], 16);
            memcpy(&vb, w.b4[p][kb * 16 + col], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf((float)va[e], (float)vb[e], acc);
        }
        d[reg] = acc;
    }
    return d;
}

// ---- barriers / fences -----------------------------------------------------------------------------
inline void __syncthreads() {
    emu::Block* b = emu::g_block;
    const unsigned gen = b->bar_gen;
    if (++b->bar_arrived == b->nthreads) {
        b->bar_arrived = 0;
        b->bar_gen++;
    } else {
        while (b->bar_gen == gen) emu::yield();
    }
}
inline void __builtin_amdgcn_wave_barrier() {}
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __threadfence_system() {}
#define __builtin_amdgcn_fence(...) ((void)0)

// ---- atomics (blocks run on several OS threads) -------------------------------------------------------
inline float atomicAdd(float* p, float v) {
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    do {
        nw = emu::bits(emu::from_bits<float>(old) + v);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return emu::from_bits<float>(old);
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- math ------------------------------------------------------------------------------------------
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline unsigned __float_as_uint(float f) { return emu::bits(f); }
inline float __uint_as_float(unsigned u) { return emu::from_bits<float>(u); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline long long __double_as_longlong(double v) { long long d; memcpy(&d, &v, 8); return d; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

// ---- launch ----------------------------------------------------------------------------------------
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emu::launch([=]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(smem))
