// Device-side building blocks shared by every kernel of the PACE-VAE train step (gfx950 / CDNA4).
//
// Execution model: ONE WAVE (64 lanes) OWNS ONE DAG at a time and keeps the DAG's whole 16-token x 64-feature
// activation tile in registers.  Lane l = 16*g + r (r = l & 15, g = l >> 4).  Two register layouts of a
// [16 tokens x 64 features] tile, 16 VGPRs each (4 x float4):
//
//   T-layout  x[t][kk] = X[token r][feature 16t + 4g + kk]        (token on the lane's r, features on t,g,kk)
//   N-layout  x[t][kk] = X[token 4g + kk][feature 16t + r]        (token on g,kk, features on t,r)
//
// They are exactly the operand/result maps of v_mfma_f32_16x16x4_f32 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[i=4*(l>>4)+reg][j=l&15]), so every product of the network chains register-to-register:
//   Y^T(T) = W   * X^T(T)   : mfma(a = W-row-fragment, b = x regs)       (linear layer, stays in T-layout)
//   Y  (N) = X   * W^T      : mfma(a = x regs,         b = W-row-fragment) (same registers, operands swapped)
//   dX^T(T)= W^T * dY^T(T)  : mfma(a = W-column-fragment, b = dy regs)
//   dW    += dY(N)^T x X(N) : mfma(a = dyN regs, b = xN regs), 16 accumulators of 4 regs, D = dW[16ot+4g+reg][16it+r]
// The contraction index of step (t,kk) is feature 16t+4g+kk (a k-permutation of the dot product; exact f32 fma
// chain per MFMA, guide §3 'FP32-input MFMA').  Weights live in LDS as row-major [rows][DVS_LD] images
// (DVS_LD = 68 floats: 16-byte aligned rows, b128 row-fragment reads at most 2-way conflicted, b32 column
// reads conflict-free).  T<->N transposes go through a private per-wave LDS scratch tile.
//
// HBM layout of an activation tile ("frag order"): float4 index (dag*4 + t)*64 + lane holds x[t] of the
// T-layout, so a wave's load/store of one t is one fully coalesced 1 KiB instruction.
#pragma once

#ifndef DVS_EMU
#include <hip/hip_runtime.h>
typedef float dvs_f32x4 __attribute__((ext_vector_type(4)));
#define DVS_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#else
#define DVS_DYN_LDS(name) char* name = emu::g_block->dyn_smem
#endif

#include <stdint.h>

typedef dvs_f32x4 f4;

constexpr int DVS_D = 64;        // d_model = ff_hidden_size (pace.py:1185)
constexpr int DVS_LD = 68;       // LDS row stride in floats
constexpr int DVS_TOK = 16;      // token slots per DAG tile (N = n + 3 <= 16)
constexpr int DVS_TILE = DVS_TOK * DVS_D;   // floats per activation tile (frag order)
constexpr int DVS_SCR = DVS_TOK * DVS_LD;   // floats of one per-wave scratch tile

struct DvsRecord {               // compact per-DAG record written by dvs_pack_features (96 bytes)
    uint8_t label[16];           // class index of token i (vertex_label_features argmax)
    uint8_t pos[16];             // position index of token i (vertex_position_features argmax)
    uint16_t parents[16];        // bit j: adjacency[j][i] == 1  (edge j -> i)
    uint16_t allowed[16];        // bit j: token i may attend token j (target_masks[., i, j] == False)
};

// Work-item / workgroup index behind an optimisation barrier: inside k_bwd_stack's phase loop the compiler otherwise hoists
// every phase's lane-derived LDS addresses out of the loop and spills them (158 VGPR spills); an opaque id keeps each phase's
// address arithmetic inside the phase.
#ifndef DVS_EMU
__device__ __forceinline__ int dvs_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ int dvs_bid() {
    int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x);   // pins the value to a scalar register whatever the caller did with it
    asm volatile("" : "+s"(b));
    return b;
}
#else
__device__ __forceinline__ int dvs_tid() { return (int)threadIdx.x; }
__device__ __forceinline__ int dvs_bid() { return (int)blockIdx.x; }
#endif

struct Lane {
    int lane, r, g, wave, nwaves;
};

__device__ __forceinline__ Lane dvs_lane() {
    Lane L;
    const int t = dvs_tid();
    L.lane = t & 63;
    L.r = L.lane & 15;
    L.g = L.lane >> 4;
    L.wave = t >> 6;
    L.nwaves = blockDim.x >> 6;
    return L;
}

__device__ __forceinline__ f4 dvs_mfma(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Orders this wave's earlier LDS accesses before its later ones as seen by its OTHER lanes (wave-private scratch
// tiles).  The LDS executes one wave's DS instructions in issue order, so on the device this only has to stop the
// compiler from reordering the accesses — no s_waitcnt: the data wait lands where the compiler needs the registers.
// In the host emulator lanes are fibers, so there it is a real wave-wide rendezvous.
__device__ __forceinline__ void dvs_wave_sync() {
#ifdef DVS_EMU
    (void)emu::exchange(0, 0);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also drains the wave's vector-memory queue
// (s_waitcnt vmcnt(0)): behind it a phase tail's prefetch loads (dvs_stage.h) would be waited for by every wave BEFORE the
// barrier, which puts the last wave's load latency back on the critical path.  Global tiles need no workgroup ordering
// here: every tile / statistics line is written and later read by the SAME wave (one in-order memory queue per wave).
__device__ __forceinline__ void dvs_lds_barrier() {
#ifdef DVS_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
}

// Stagger (experiment knob, -DDVS_STAGGER=n: n x 64 cycles): the two waves of a SIMD (one of each wave group) leave a phase's
// opening barrier in lock-step and run the same instruction stream, so both want the matrix pipe, then both the VALU; the
// younger wave's FIRST DAG of a phase took 37 k cycles against 22 k once the two had drifted apart (round 1).
// Holding the younger group back by a fraction of a DAG de-phases them from the start.
#ifndef DVS_STAGGER
#define DVS_STAGGER 0
#endif
__device__ __forceinline__ void dvs_stagger(int wave) {
#if !defined(DVS_EMU) && DVS_STAGGER > 0
    if (wave >= 4) {
        for (int i = 0; i < DVS_STAGGER / 16; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles per step
    }
#endif
}

#ifdef DVS_EMU
#define DVS_SCHED_FENCE() ((void)0)
#else
#define DVS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// 1 / x as ONE v_rcp_f32 (1 ulp) instead of the IEEE-exact division sequence (v_div_scale, v_rcp, 4 x fma, v_div_fmas,
// v_div_fixup: 12 instructions) for gradient terms whose parity bound is 2e-4 of the tensor maximum (sigmoid / softmax
// denominators of the loss head's backward, which is vector-instruction bound)
__device__ __forceinline__ float dvs_rcp(float x) {
#ifdef DVS_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ f4 f4_zero() { return f4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f4 f4_splat(float v) { return f4{v, v, v, v}; }

// ---- weight images in LDS -----------------------------------------------------------------------------
// copy a row-major [rows][cols] global matrix (leading dimension ldg) into an LDS image with stride ldl
__device__ __forceinline__ void dvs_stage_matrix(float* dst, int ldl, const float* __restrict__ src, int ldg, int rows,
                                                 int cols) {
    // Loads in batches of 4 per thread before the first store (a load -> store loop pays one L2 round trip per iteration),
    // and every workgroup starts at a different element: all workgroups of a launch read the same matrix at the same time
    // and would otherwise queue on the same L2 lines (measured on the bf16 images: +5 % DAGs/s).
    const int c4 = cols >> 2, n = rows * c4, step = blockDim.x;
    const int rot = (int)(((unsigned)dvs_bid() * 2654435761u) % (unsigned)n);
    int i = dvs_tid();
    for (; i + 3 * step < n; i += 4 * step) {
        f4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int j = i + u * step + rot;
            j = j >= n ? j - n : j;
            const int row = j / c4, c = (j - row * c4) << 2;
            v[u] = *(const f4*)(src + (size_t)row * ldg + c);   // all sources are 16-byte aligned
            off[u] = row * ldl + c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f4*)(dst + off[u]) = v[u];
    }
    for (; i < n; i += step) {
        int j = i + rot;
        j = j >= n ? j - n : j;
        const int row = j / c4, c = (j - row * c4) << 2;
        *(f4*)(dst + row * ldl + c) = *(const f4*)(src + (size_t)row * ldg + c);
    }
}
__device__ __forceinline__ void dvs_stage_vector(float* dst, const float* __restrict__ src, int n) {
    for (int i = dvs_tid(); i < n; i += blockDim.x) dst[i] = src[i];
}

// ---- head-aligned slot order inside the attention sublayers -------------------------------------------------------
// A T-layout register (g, kk) of tile t normally holds feature 16t + 4g + kk, so one MFMA step (fixed kk, the 4 lane
// groups g as contraction index) mixes features of BOTH heads of the tile (head = feature / 8).  q, k, v and the
// attention output are internal to the sublayer, so their feature order is free: the in-projection ROWS and the
// out-projection COLUMNS are staged into LDS permuted by the 4x4 transpose pi(4a + b) = 4b + a within every block of
// 16.  Slot (g, kk) then holds feature 16t + 4kk + g: MFMA step kk contracts over 4 consecutive features of ONE head
// (head 2t + (kk >> 1)), result rows reg <-> head 2t + (reg >> 1).  Scores need 2 unmasked MFMAs per head instead of
// 4 half-masked ones, and the per-head outputs are merged by register selection.  pi is an involution.
__device__ __forceinline__ int dvs_pi(int i) { return (i & ~15) | ((i & 3) << 2) | ((i >> 2) & 3); }
__device__ __forceinline__ void dvs_stage_vector_perm(float* dst, const float* __restrict__ src, int n) {
    for (int i = dvs_tid(); i < n; i += blockDim.x) dst[i] = src[dvs_pi(i)];
}

// row fragment: element kk = W[row0 + r][16t + 4g + kk]
__device__ __forceinline__ f4 dvs_wrow(const float* W, int ld, int row0, int t, const Lane& L) {
    return *(const f4*)(W + (row0 + L.r) * ld + 16 * t + 4 * L.g);
}
// column fragment: element kk = W[16t + 4g + kk][col0 + r]
__device__ __forceinline__ f4 dvs_wcol(const float* W, int ld, int col0, int t, const Lane& L) {
    const float* p = W + (16 * t + 4 * L.g) * ld + col0 + L.r;
    return f4{p[0], p[ld], p[2 * ld], p[3 * ld]};
}
// feature vector (bias, LayerNorm gamma/beta) in T-layout: element kk = v[16t + 4g + kk]
__device__ __forceinline__ f4 dvs_vecT(const float* v, int t, const Lane& L) { return *(const f4*)(v + 16 * t + 4 * L.g); }

// ---- register-chained products ----------------------------------------------------------------------------
// Both walk the contraction in steps of one 16-feature tile: the weight fragments of step s+1 are fetched from
// LDS (double-buffered, 2 x OT float4) while the 4*OT MFMAs of step s issue; a scheduling barrier per step keeps the
// compiler from hoisting every fragment of the fully unrolled loop to the top (which costs >400 VGPRs and spills).
// Within a step the contraction index kk is outermost and the output tile innermost, so OT independent accumulator
// chains are in flight and the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 never stalls its 32-cycle issue.

// y^T[OT] (T) += W[row0 + 16*OT rows][16*IT cols] * x^T[IT] (T)
template <int OT, int IT>
__device__ __forceinline__ void dvs_mat_T(f4 (&y)[OT], const f4 (&x)[IT], const float* W, int ld, int row0, const Lane& L) {
    f4 w[OT], wn[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) w[ot] = dvs_wrow(W, ld, row0 + 16 * ot, 0, L);
#pragma unroll
    for (int t = 0; t < IT; ++t) {
        if (t + 1 < IT) {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) wn[ot] = dvs_wrow(W, ld, row0 + 16 * ot, t + 1, L);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma(w[ot][kk], x[t][kk], y[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) w[ot] = wn[ot];
        DVS_SCHED_FENCE();
    }
}
// dx^T[IT] (T) += W^T * dy^T[OT] (T), W = [16*OT rows (row0..)][16*IT cols]
template <int IT, int OT>
__device__ __forceinline__ void dvs_mat_Tt(f4 (&dx)[IT], const f4 (&dy)[OT], const float* W, int ld, int row0, const Lane& L) {
    f4 w[IT], wn[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) w[it] = dvs_wcol(W + row0 * ld, ld, 16 * it, 0, L);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        if (ot + 1 < OT) {
#pragma unroll
            for (int it = 0; it < IT; ++it) wn[it] = dvs_wcol(W + row0 * ld, ld, 16 * it, ot + 1, L);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int it = 0; it < IT; ++it) dx[it] = dvs_mfma(w[it][kk], dy[ot][kk], dx[it]);
#pragma unroll
        for (int it = 0; it < IT; ++it) w[it] = wn[it];
        DVS_SCHED_FENCE();
    }
}
// dW[ot][it] += dY(N)[ot]^T (x) X(N)[it] over this DAG's 16 tokens; D = dW[16ot + 4g + reg][16it + r]
template <int OT, int IT>
__device__ __forceinline__ void dvs_outer_acc(f4 (&dw)[OT][IT], const f4 (&dyN)[OT], const f4 (&xN)[IT]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int it = 0; it < IT; ++it) dw[ot][it] = dvs_mfma(dyN[ot][kk], xN[it][kk], dw[ot][it]);
}

// ---- T <-> N transposes through the wave's private scratch tile [16][DVS_LD] ---------------------------------
template <int NT>
__device__ __forceinline__ void dvs_t2n(f4 (&out)[NT], const f4 (&in)[NT], float* scr, const Lane& L) {
#pragma unroll
    for (int t = 0; t < NT; ++t) *(f4*)(scr + L.r * DVS_LD + 16 * t + 4 * L.g) = in[t];
    dvs_wave_sync();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float* p = scr + (4 * L.g) * DVS_LD + 16 * t + L.r;
        out[t] = f4{p[0], p[DVS_LD], p[2 * DVS_LD], p[3 * DVS_LD]};
    }
    dvs_wave_sync();
    // keep all 4*NT reads issued HERE, back to back: left alone, hipcc sinks each one next to the MFMA that consumes
    // it and pays a full LDS round trip (s_waitcnt lgkmcnt(0)) every 8 MFMAs
    DVS_SCHED_FENCE();
}
template <int NT>
__device__ __forceinline__ void dvs_n2t(f4 (&out)[NT], const f4 (&in)[NT], float* scr, const Lane& L) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float* p = scr + (4 * L.g) * DVS_LD + 16 * t + L.r;
        p[0] = in[t][0];
        p[DVS_LD] = in[t][1];
        p[2 * DVS_LD] = in[t][2];
        p[3 * DVS_LD] = in[t][3];
    }
    dvs_wave_sync();
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = *(const f4*)(scr + L.r * DVS_LD + 16 * t + 4 * L.g);
    dvs_wave_sync();
    DVS_SCHED_FENCE();
}

// ---- frag-order HBM tiles -----------------------------------------------------------------------------------
__device__ __forceinline__ void dvs_load_tile(f4 (&x)[4], const float* __restrict__ base, size_t dag, const Lane& L) {
    const f4* p = (const f4*)(base + dag * DVS_TILE) + L.lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = p[t * 64];
}
__device__ __forceinline__ void dvs_store_tile(float* __restrict__ base, size_t dag, const f4 (&x)[4], const Lane& L) {
    f4* p = (f4*)(base + dag * DVS_TILE) + L.lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) p[t * 64] = x[t];
}

// ---- reductions over the 64 features of a token (T-layout: 16 in-lane values x 4 lane groups g) --------------
// v[l] + v[l ^ 16] (resp. ^ 32), max likewise: the partner's value through gfx950's v_permlane16_swap / v_permlane32_swap — a
// VALU operation — instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32 (an LDS-crossbar round trip of ~100+ cycles
// sitting in the softmax / LayerNorm dependency chains: 32 of them per DAG in the attention forward alone, 19 % of its
// loop by tools/attn_stamps.py).  The swap exchanges the odd 16-lane rows of its first operand with the even rows of the
// second (32: the upper half with the lower half), so with both operands = v the two results hold {own, partner} in an
// order that depends on the row — irrelevant to a commutative op, and bitwise the same sum as before.  Inline asm: the
// builtin (__builtin_amdgcn_permlane16_swap) drops the second result on ROCm 7.2 (it emitted v1 + v1).  s_nop 1: the
// VALU-write -> permlane-read wait states hipcc inserts for its own permlanes; nothing pads inside an asm statement.
#ifndef DVS_EMU
__device__ __forceinline__ void dvs_swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void dvs_swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float dvs_add_x16(float v) { float a = v, b = v; dvs_swap16(a, b); return a + b; }
__device__ __forceinline__ float dvs_add_x32(float v) { float a = v, b = v; dvs_swap32(a, b); return a + b; }
__device__ __forceinline__ float dvs_max_x16(float v) { float a = v, b = v; dvs_swap16(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float dvs_max_x32(float v) { float a = v, b = v; dvs_swap32(a, b); return fmaxf(a, b); }
#else
__device__ __forceinline__ float dvs_add_x16(float v) { return v + __shfl_xor(v, 16); }
__device__ __forceinline__ float dvs_add_x32(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float dvs_max_x16(float v) { return fmaxf(v, __shfl_xor(v, 16)); }
__device__ __forceinline__ float dvs_max_x32(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
#endif
__device__ __forceinline__ float dvs_sum_g(float v) { return dvs_add_x32(dvs_add_x16(v)); }
__device__ __forceinline__ float dvs_max_g(float v) { return dvs_max_x32(dvs_max_x16(v)); }
__device__ __forceinline__ float dvs_sum_r(float v) {   // over the 16 lanes r of one g group
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float dvs_sum_wave(float v) { return dvs_sum_g(dvs_sum_r(v)); }
__device__ __forceinline__ float dvs_tile_sum(const f4 (&x)[4]) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (x[t][0] + x[t][1]) + (x[t][2] + x[t][3]);
    return dvs_sum_g(s);
}

// LayerNorm statistics of token r over its 64 features (biased variance, eps 1e-5: torch.nn.LayerNorm)
__device__ __forceinline__ void dvs_ln_stats(const f4 (&x)[4], float& mean, float& rstd) {
    mean = dvs_tile_sum(x) * (1.f / 64.f);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float d = x[t][kk] - mean;
            s += d * d;
        }
    rstd = 1.0f / sqrtf(dvs_sum_g(s) * (1.f / 64.f) + 1e-5f);
}

// ---- counter-based dropout / noise ------------------------------------------------------------------------------
// Stateless: mask bits are a pure function of (seed, site, global DAG index, element), so the backward pass
// regenerates them instead of storing 34 masks, and a batch sharded over ranks draws the same bits as the same
// batch on one GPU.  oracle/rng.py restates these functions in numpy for the dropout-on parity tests.
__device__ __forceinline__ uint32_t dvs_fmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t dvs_site_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t site, uint32_t dag) {
    uint32_t k = dvs_fmix32(seed_lo ^ ((site + 1u) * 0x632BE5ABu));
    k = dvs_fmix32(k ^ seed_hi ^ (dag * 0x9E3779B1u));
    return k;
}
// two Bernoulli(keep) draws for the element pair `pair` of one (site, dag): keep iff 16-bit half >= thr16
__device__ __forceinline__ uint32_t dvs_draw(uint32_t key, uint32_t pair) { return dvs_fmix32(key ^ (pair * 0x9E3779B1u)); }

struct DvsDrop {
    uint32_t thr16;      // round(p * 65536); drop iff half < thr16
    float scale;         // 1 / (1 - thr16/65536)
    int on;              // training && p > 0
};
// element index of (token, feature) within a [tokens][64] site: tok*64 + f; a lane's f4 covers features 16t+4g..+3
// -> pair indices (tok*64 + 16t + 4g)/2 and +1.
// tok0: token index of the tile's row 0 (0 on the one-tile path; 16 * tile-in-DAG on the wide path)
template <int NTILE = 4>
__device__ __forceinline__ void dvs_dropout_tile(f4 (&x)[NTILE], uint32_t key, const DvsDrop& D, const Lane& L, int tok0 = 0) {
    if (!D.on) return;
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        const uint32_t p0 = (uint32_t)((tok0 + L.r) * 64 + 16 * t + 4 * L.g) >> 1;
        const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
        x[t][0] = ((h0 & 0xFFFFu) >= D.thr16) ? x[t][0] * D.scale : 0.f;
        x[t][1] = ((h0 >> 16) >= D.thr16) ? x[t][1] * D.scale : 0.f;
        x[t][2] = ((h1 & 0xFFFFu) >= D.thr16) ? x[t][2] * D.scale : 0.f;
        x[t][3] = ((h1 >> 16) >= D.thr16) ? x[t][3] * D.scale : 0.f;
    }
}
// the same draws as keep-bits (bit 4t + kk <-> x[t][kk]) and their application: a mask that is applied twice (the FFN backward's
// hidden: to the recomputed activation and to its gradient) is drawn once
template <int NTILE = 4>
__device__ __forceinline__ uint32_t dvs_dropout_bits(uint32_t key, const DvsDrop& D, const Lane& L, int tok0 = 0) {
    uint32_t bits = 0xFFFFFFFFu;
    if (D.on) {
        bits = 0;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) {
            const uint32_t p0 = (uint32_t)((tok0 + L.r) * 64 + 16 * t + 4 * L.g) >> 1;
            const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
            bits |= ((h0 & 0xFFFFu) >= D.thr16 ? 1u : 0u) << (4 * t);
            bits |= ((h0 >> 16) >= D.thr16 ? 1u : 0u) << (4 * t + 1);
            bits |= ((h1 & 0xFFFFu) >= D.thr16 ? 1u : 0u) << (4 * t + 2);
            bits |= ((h1 >> 16) >= D.thr16 ? 1u : 0u) << (4 * t + 3);
        }
    }
    return bits;
}
template <int NTILE = 4>
__device__ __forceinline__ void dvs_dropout_apply(f4 (&x)[NTILE], uint32_t bits, const DvsDrop& D) {
    if (!D.on) return;
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) x[t][kk] = ((bits >> (4 * t + kk)) & 1u) ? x[t][kk] * D.scale : 0.f;
}
// single element (used for attention probabilities): element index e
__device__ __forceinline__ float dvs_dropout_elem(float v, uint32_t key, uint32_t e, const DvsDrop& D) {
    const uint32_t h = dvs_draw(key, e >> 1);
    const uint32_t half = (e & 1u) ? (h >> 16) : (h & 0xFFFFu);
    return (half >= D.thr16) ? v * D.scale : 0.f;
}
