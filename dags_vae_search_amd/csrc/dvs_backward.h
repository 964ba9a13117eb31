// Backward kernel argument blocks, launchers and the gradient-slab reduction helpers.
//
// Parameter gradients: every backward kernel runs on exactly `nslab` persistent workgroups.  A wave accumulates its
// DAGs' weight-gradient contributions in registers (MFMA accumulators, dvs_outer_acc), the workgroup's waves are
// summed in a fixed order through LDS, and the workgroup writes ONE partial ("slab") per parameter it owns to
// slab[blockIdx][param offset].  k_reduce_slabs then sums the slabs in a fixed order into the flat gradient buffer:
// no float atomics on HBM, bitwise reproducible gradients.
#pragma once
#include "dvs_kernels.h"
#include "dvs_stage.h"

struct FfnBwdArgs {
    DvsDims dims;
    const float* xin;            // pre-sum feeding this sublayer (or embedding output)
    DvsLN ln;                    // LayerNorm of the producing sublayer (stats null = none)
    const float *l1_w, *l1_b, *l2_w, *l2_b;
    const void* wimg;            // this sublayer's FFN image block (dvs_wimg.h)
    const float* gpre;           // d(pre of this sublayer)  — or d(LayerNorm(pre)) when `own` is set
    // optional: the incoming gradient is w.r.t. LN_own(pre_own); pull it back through that LayerNorm first
    const float* own_pre;
    DvsLN own;
    float* gout;                 // d(pre of the producing sublayer) (or d embedding output)
    int site_hidden, site_post;
    float* slab;
    int64_t P;
    int64_t o_l1_w, o_l1_b, o_l2_w, o_l2_b, o_ln_g, o_ln_b, o_own_g, o_own_b;   // -1 = absent
};

struct ProjBwdArgs {             // backward of 1..3 stacked 64->64 projections sharing one input X
    DvsDims dims;
    const float* xin;
    DvsLN ln;                    // stats null: X is used as is (embedding output or decoder memory)
    const float* w;              // first row of the stacked weight block [64*NPROJ][64]
    const void* wimg;            // W_p^T image pairs of these projections (dvs_wimg.h: DvsAttnImg::WinT + 2 p0 DVS_IMG64)
    const float* gy[3];          // d(projection outputs), T-layout frag tiles
    const float* gres;           // optional residual gradient added to dX before the LayerNorm backward
    float* gout;                 // result: d(pre of producer) / d(X)
    int accumulate_out;          // 1: gout += (decoder memory gradient over layers)
    int slot_order;              // 1: the projections' output rows are in attention slot order (dvs_pi)
    // split mode (3 projections, cross-attention: q of one input, k and v of another — one phase instead of two): projection 0
    // is the backward of xin (with ln / gres / gout as above), projections 1, 2 that of xin2 (used as is), whose gradient goes to
    // gout2 (+= when accumulate_out2).  xin2 == nullptr: all projections share xin.
    const float* xin2;
    float* gout2;
    int accumulate_out2;
    float* slab;
    int64_t P;
    int64_t o_w, o_b, o_ln_g, o_ln_b;
};

struct AttnBwdArgs {
    DvsDims dims;
    const DvsRecord* rec;
    const float* xin;
    DvsLN ln;
    const float* kv;             // null = self-attention
    const float *in_w, *in_b, *out_w, *out_b;
    const void* wimg;            // this sublayer's attention image block (dvs_wimg.h); one-tile path
    const float* gpre;           // d(pre of this sublayer)
    float *gq, *gk, *gv;         // outputs: d(q), d(k), d(v) projections (T-layout frag tiles)
    int site_prob, site_post;
    float* slab;
    int64_t P;
    int64_t o_out_w, o_out_b;
    const float* qkv;            // wide path: the forward's q (scaled), k, v (AttnArgs::qkv)
};

struct LatentBwdArgs {
    DvsDims dims;
    const float* gmem;           // d memory [B][1024]
    const float *mu, *logvar, *epsv;
    const float* limg;           // latent weight images of this step (dvs_wimg.h: DvsLatImg), written by the forward
    const float* gcoef;
    float* gz;                   // [B][64] = d mu | d logvar
    float* genc;                 // [B][1024] d enc_out
};

struct FcDwArgs {
    DvsDims dims;
    const float* gz;             // [B][64]
    const float* xenc;           // [B][1024]
    const float* gmem;           // [B][1024]
    const float* z;              // [B][32]
    float* fcpart;               // [DVS_FC_PARTS][P]: per-batch-quarter partials of the fc1/fc2/fc3 gradients
    int64_t P;
    int64_t o_fc1_w, o_fc1_b, o_fc2_w, o_fc2_b, o_fc3_w, o_fc3_b;
};

struct ReduceArgs {
    const float* slab;           // [nslab][P] per-workgroup partials (all parameters except fc1/fc2/fc3)
    const float* fcpart;         // [DVS_FC_PARTS][P] partials of fc1/fc2 (offsets [fc_lo1, fc_hi1)) and fc3 ([fc_lo2, fc_hi2))
    float* grads;
    int64_t P;
    int nslab;
    int64_t fc_lo1, fc_hi1, fc_lo2, fc_hi2;
    float* sqpart;               // optional [gridDim.x]: this workgroup's sum of squares of the gradient entries it wrote — the
                                 // partials k_adam derives the clip coefficient from (dvs_loss_backward_sq, include/dvs.h)
};

// One phase of a chained backward launch (k_bwd_stack, k_backward.hip).  Plain data: the table travels as the kernel argument.
enum { DVS_PH_FFN = 0, DVS_PH_ATTN = 1, DVS_PH_PROJ1 = 2, DVS_PH_PROJ2 = 3, DVS_PH_PROJ3 = 4 };
#define DVS_STACK_PHASES 9
struct BwdPhase {
    int kind, pad;
    union {
        FfnBwdArgs f;
        AttnBwdArgs a;
        ProjBwdArgs p;
    } u;
};
struct BwdStackArgs {
    int nphase;
    int has_latent;                          // run the latent block's backward (lat) ahead of phase 0 (encoder chain)
    BwdPhase ph[DVS_STACK_PHASES];
    DvsStagePlan plan[DVS_STACK_PHASES];     // filled by dvs_launch_bwd_stack (dvs_stage.h): what each phase keeps in LDS
    LatentBwdArgs lat;
};
static_assert(sizeof(BwdStackArgs) <= 4096, "kernel argument block limit");
// nw: waves per workgroup, 8 or 4 (dvs_api.hip: dvs_waves_per_wg); every launch of one step uses the same value
void dvs_launch_bwd_stack(const BwdStackArgs& s, int tag, int grid, int nw, dvs_stream_t st);   // tag 0 decoder, 1 encoder (profile names)
void dvs_launch_ffn_bwd(const FfnBwdArgs& a, int grid, int nw, dvs_stream_t st);
void dvs_launch_proj_bwd(const ProjBwdArgs& a, int nproj, int grid, int nw, dvs_stream_t st);
void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, int nw, dvs_stream_t st);
void dvs_launch_loss_bwd(const LossArgs& a, int grid, dvs_stream_t st);
void dvs_launch_embed_bwd(const EmbedArgs& a, const float* gout2, int site2, int grid, int nw, dvs_stream_t st);
void dvs_launch_latent_bwd(const LatentBwdArgs& a, dvs_stream_t st);
void dvs_launch_fc_dw(const FcDwArgs& a, dvs_stream_t st);
void dvs_launch_reduce_slabs(const ReduceArgs& a, dvs_stream_t st);
// have_partials: scratch[2 ..] already holds dvs_sq_parts(n) partial sums of squares of `grads` (written by k_reduce_slabs)
void dvs_launch_clip_adam(int64_t n, float* params, float* grads, float* m, float* v, float lr, float b1, float b2,
                          float eps, int64_t step, float max_norm, float* scratch, const float* guard, bool have_partials,
                          dvs_stream_t st);
// workgroups of k_reduce_slabs for an n-float gradient = partial sums of squares it leaves behind
inline int dvs_sq_parts(int64_t n) { return (int)(((n + 3) / 4 + 63) / 64); }

// ---- slab reduction helpers ---------------------------------------------------------------------------------------
// After its DAG loop a workgroup adds its waves' register accumulators and writes ONE partial per parameter to its
// slab.  Two-phase, so that all waves work in parallel and only two barriers are paid per batch of tensors:
//   stage:  every wave dumps its accumulators into its own LDS region          (no barrier)
//   -- __syncthreads() --
//   flush:  all threads add the nwaves regions in fixed wave order and store to the slab (coalesced)
// Region sizes (floats): matrix nwaves*256*OT*IT, vector nwaves*16*NT.  Callers lay the regions out in the
// workgroup's LDS (the weight images are dead by then) and put a barrier between flush and the next stage that
// reuses the space.
template <int OT, int IT>
__device__ __forceinline__ void dvs_stage_dw(float* region, const f4 (&dw)[OT][IT], const Lane& L) {
    constexpr int COLS = 16 * IT;
    float* mine = region + L.wave * (256 * OT * IT);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) mine[(16 * ot + 4 * L.g + reg) * COLS + 16 * it + L.r] = dw[ot][it][reg];
}
// dense [16*OT][16*IT] sum -> dst[row * ld_dst + col] for row < rows, col < cols_used
template <int OT, int IT>
__device__ __forceinline__ void dvs_flush_dw(const float* region, float* dst, const Lane& L, int rows = 16 * OT,
                                             int ld_dst = 16 * IT, int cols_used = 16 * IT, bool rperm = false,
                                             bool cperm = false) {
    constexpr int COLS = 16 * IT, SZ = 256 * OT * IT;
    if (cperm) {                                         // attention slot order -> parameter order (dvs_pi)
        for (int i = dvs_tid(); i < rows * COLS; i += blockDim.x) {
            float s = region[i];
            for (int w = 1; w < L.nwaves; ++w) s += region[w * SZ + i];
            const int row = i / COLS, col = i - row * COLS;
            dst[(size_t)(rperm ? dvs_pi(row) : row) * ld_dst + (cperm ? dvs_pi(col) : col)] = s;
        }
        return;
    }
    if (cols_used == COLS && (ld_dst & 3) == 0) {       // 16-byte path (every 64-wide tensor)
        for (int i = dvs_tid() * 4; i < rows * COLS; i += blockDim.x * 4) {
            f4 s = *(const f4*)(region + i);
            for (int w = 1; w < L.nwaves; ++w) s += *(const f4*)(region + w * SZ + i);
            const int row = i / COLS, col = i - row * COLS;
            *(f4*)(dst + (size_t)(rperm ? dvs_pi(row) : row) * ld_dst + col) = s;
        }
        return;
    }
    for (int i = dvs_tid(); i < rows * COLS; i += blockDim.x) {
        float s = region[i];
        for (int w = 1; w < L.nwaves; ++w) s += region[w * SZ + i];
        const int row = i / COLS, col = i - row * COLS;
        if (col < cols_used) dst[(size_t)row * ld_dst + col] = s;
    }
}
// per-feature vector kept as T-layout per-lane partial sums v[t][kk] (feature 16t+4g+kk, summed over the tokens r
// this lane handled).  The sum over the 16 token lanes goes through the wave's scratch tile: the partials are written
// as a [16 tokens][64 features] tile and lane f adds column f (16 conflict-free LDS reads) — 20 LDS operations instead
// of 64 dependent cross-lane shuffles per vector.  One copy per wave is parked in `region`.
template <int NT>
__device__ __forceinline__ void dvs_stage_vec(float* region, const f4 (&v)[NT], float* scr, const Lane& L) {
#pragma unroll
    for (int t = 0; t < NT; ++t) *(f4*)(scr + L.r * DVS_LD + 16 * t + 4 * L.g) = v[t];
    dvs_wave_sync();
    if (L.lane < 16 * NT) {
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s += scr[rr * DVS_LD + L.lane];
        region[L.wave * (16 * NT) + L.lane] = s;
    }
    dvs_wave_sync();
}
template <int NT>
__device__ __forceinline__ void dvs_flush_vec(const float* region, float* dst, const Lane& L, int n = 16 * NT,
                                              bool perm = false) {
    for (int i = dvs_tid(); i < n; i += blockDim.x) {
        float s = region[i];
        for (int w = 1; w < L.nwaves; ++w) s += region[w * (16 * NT) + i];
        dst[perm ? dvs_pi(i) : i] = s;
    }
}
constexpr int DVS_RED_MAT = 4 * 4096;     // floats of a staged 64x64 matrix for 4 waves
constexpr int DVS_RED_VEC = 4 * 64;       // floats of a staged 64-vector for 4 waves

// LayerNorm backward for token r (T-layout): given dx (gradient w.r.t. the LayerNorm output), xhat, rstd and gamma,
// returns d(pre) in place and accumulates d gamma / d beta partials.
__device__ __forceinline__ void dvs_ln_bwd(f4 (&dx)[4], const f4 (&xhat)[4], float rstd, const float* lg, f4 (&dgam)[4],
                                           f4 (&dbet)[4], const Lane& L) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 g = dvs_vecT(lg, t, L);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            dgam[t][kk] += dx[t][kk] * xhat[t][kk];
            dbet[t][kk] += dx[t][kk];
            const float dxh = dx[t][kk] * g[kk];
            dx[t][kk] = dxh;
            s1 += dxh;
            s2 += dxh * xhat[t][kk];
        }
    }
    s1 = dvs_sum_g(s1) * (1.f / 64.f);
    s2 = dvs_sum_g(s2) * (1.f / 64.f);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dx[t][kk] = rstd * (dx[t][kk] - s1 - xhat[t][kk] * s2);
}

// ---- cooperative weight gradients (8-wave backward kernels) --------------------------------------------------------
// A wave that accumulated a whole 64x64 dW for its own DAGs needs 64 accumulator registers per matrix, which pins the
// backward kernels at one wave per SIMD.  Instead the 8 waves of a workgroup form two independent GROUPS of four
// (waves 0-3 and 4-7; a SIMD hosts one wave of each, so the groups' instruction streams interleave on every SIMD).
// Every wave parks the two operand tiles of its DAG row-major ([token][feature], stride DVS_LD) in its LDS slots, the
// group synchronises on its own LDS counter (an s_barrier would lock both groups into the same phase and forfeit the
// MFMA/VALU overlap), and wave w accumulates rows 16*(w&3).. of dW over the 4 DAGs of its group: 16 accumulator
// registers per matrix instead of 64.  The product itself runs on the bf16 pipe from bf16 hi / lo parked tiles read back
// transposed (dvs_coop_dw_bf, dvs_bf16.h).
struct DvsGroup {
    int* counter;      // LDS word of this wave's group
    int target;        // arrivals expected at the next barrier
};
__device__ __forceinline__ void dvs_group_barrier(DvsGroup& G, const Lane& L) {
    G.target += 4;
#ifdef DVS_EMU
    if (L.lane == 0) atomicAdd(G.counter, 1);
    while (__atomic_load_n(G.counter, __ATOMIC_RELAXED) < G.target) emu::yield();
    (void)emu::exchange(0, 0);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // this wave's slot writes are visible (lgkmcnt(0))
    if (L.lane == 0) atomicAdd(G.counter, 1);
    // bounded spin (~1 s): a lost arrival must never hang the GPU; parity tests catch the wrong result it would give
    for (int spins = 0; __hip_atomic_load(G.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < G.target &&
                        spins < (1 << 24); ++spins)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
// park a T-layout tile row-major in a slot
__device__ __forceinline__ void dvs_park_T(float* slot, const f4 (&v)[4], const Lane& L) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *(f4*)(slot + L.r * DVS_LD + 16 * t + 4 * L.g) = v[t];
}
// sum over the 16 token rows of column `lane` of a parked tile (bias / LayerNorm-parameter gradients: 1 register)
__device__ __forceinline__ float dvs_colsum(const float* slot, const Lane& L) {
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) s += slot[rr * DVS_LD + L.lane];
    return s;
}
// Add the two groups' partial dW (rows 16*(wave&3).. each) and write the 64x64 result to the slab, in two halves around ONE
// workgroup barrier shared by all matrices (and the vector sums) of a phase's epilogue: waves 0-3 stage every matrix in its
// own 4096-float LDS buffer, barrier, waves 4-7 add theirs and store.  (Round 1 paid two barriers per matrix.)
// NW = waves per workgroup: 8 (two groups, as described) or 4 (ONE group — the narrow mapping dvs_api.hip picks for small
// batches, one DAG round on every CU instead of two waves per SIMD on half of them): then the group's accumulators ARE the
// workgroup's partial and go straight to the slab.
template <int NW = 8>
__device__ __forceinline__ void dvs_coop_stage(float* buf, const f4 (&acc)[4], const Lane& L) {
    if (NW == 4 || L.wave >= 4) return;
    const int ot = L.wave & 3;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) buf[(16 * ot + 4 * L.g + reg) * 64 + 16 * it + L.r] = acc[it][reg];
}
template <int NW = 8>
__device__ __forceinline__ void dvs_coop_flush(const float* buf, float* dst, const f4 (&acc)[4], const Lane& L, bool rperm = false,
                                               bool cperm = false, int ld_dst = 64) {
    if (NW == 8 && L.wave < 4) return;
    const int ot = L.wave & 3;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * ot + 4 * L.g + reg, col = 16 * it + L.r;
            const float v = NW == 8 ? acc[it][reg] + buf[row * 64 + col] : acc[it][reg];
            dst[(size_t)(rperm ? dvs_pi(row) : row) * ld_dst + (cperm ? dvs_pi(col) : col)] = v;
        }
}
// LayerNorm backward without the parameter-gradient accumulation (done by column sums of parked tiles)
__device__ __forceinline__ void dvs_ln_bwd_core(f4 (&dx)[4], const f4 (&xhat)[4], float rstd, const float* lg, const Lane& L) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 g = dvs_vecT(lg, t, L);
        dx[t] *= g;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s1 += dx[t][kk];
            s2 += dx[t][kk] * xhat[t][kk];
        }
    }
    s1 = dvs_sum_g(s1) * (1.f / 64.f);
    s2 = dvs_sum_g(s2) * (1.f / 64.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) dx[t] = (dx[t] - s1 - xhat[t] * s2) * rstd;
}

__device__ __forceinline__ void dvs_zero_rows(f4 (&g)[4], int N, const Lane& L) {      // rows of padding tokens (r >= N)
    if (L.r >= N) {
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t] = f4_zero();
    }
}
__device__ __forceinline__ void dvs_load_grad(f4 (&g)[4], const float* base, size_t dag, int N, const Lane& L) {
    dvs_load_tile(g, base, dag, L);
    dvs_zero_rows(g, N, L);
}
