// Kernel argument blocks + launcher prototypes (one launcher per kernel; all enqueue on the given stream).
#pragma once
#include "dvs_device.h"
#include "dvs_host.h"
#include "dvs_stage.h"

#ifndef DVS_EMU
typedef hipStream_t dvs_stream_t;
#else
typedef void* dvs_stream_t;
#endif

struct DvsLN {                   // LayerNorm applied in a consumer's prologue: x = (pre - mean) * rstd * g + b
    const float* stats;          // [tiles][32] (mean[16], rstd[16]); null = identity (embedding output)
    const float* g;
    const float* b;
};

struct PackArgs {
    int B, N, C;
    const float* lab1h;
    const float* pos1h;
    const float* adj;
    const uint8_t* tmask;
    DvsRecord* rec;
    int* status;
};

struct EmbedArgs {
    DvsDims dims;
    const DvsRecord* rec;
    const float *W1, *W2, *lab_w, *lab_b;
    const float* embimg;         // embedding block of this step (dvs_wimg.h: DvsEmbImg); one-tile kernels stage it verbatim
    float* out;                  // [B][1024] frag order
    int site;                    // dropout site of the first dropout (second = site + 1)
    float* out2;                 // optional second embedding of the same graphs under dropout sites (site2, site2 + 1):
    int site2;                   // the decoder-side input of a train-mode step (one-tile forward kernel only)
    // backward only
    const float* gout;           // d out
    float* slab;                 // this kernel's slab base: [grid][P]
    int64_t P;
    int64_t oW1, oW2, olab_w, olab_b;
};

struct AttnArgs {
    DvsDims dims;
    const DvsRecord* rec;
    const float* xin;            // [B][1024] input pre-sum (or embedding output)
    DvsLN ln;                    // LayerNorm of the producing sublayer
    const float* kv;             // null: self-attention; else decoder memory [B][1024] (no LayerNorm)
    const float *in_w, *in_b, *out_w, *out_b;
    const void* wimg;            // this sublayer's attention image block (dvs_wimg.h); one-tile path
    float* out_pre;              // x + dropout(attn(x))
    float* out_stats;
    int site_prob, site_post;
    float* qkv;                  // wide path, optional: q (scaled), k, v of every tile saved for the backward, [B NT][12 output tiles][64 lanes x 4]
};

struct FfnArgs {
    DvsDims dims;
    const float* xin;
    DvsLN ln;
    const float *l1_w, *l1_b, *l2_w, *l2_b;
    const void* wimg;            // this sublayer's FFN image block (dvs_wimg.h)
    float* out_pre;
    float* out_stats;
    float* out_norm;             // optional: LayerNorm(out_pre) with (ng, nb) — encoder output for fc1/fc2
    const float *ng, *nb;
    int site_hidden, site_post;
};

struct LatentArgs {
    DvsDims dims;
    const float* xenc;           // [B][1024] LayerNorm'ed encoder output, frag order
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *fc3_w, *fc3_b;
    const float* limg;           // latent weight images of this step (dvs_wimg.h: DvsLatImg)
    const float* eps_in;         // optional [B][32], already scaled
    float *mu, *logvar, *z, *epsv;   // [B][32]
    float* mem;                  // [B][1024] frag order (null: encode only)
    float* dag_loss;             // [B][2]; KL goes to [.][1]
};

struct LossArgs {
    DvsDims dims;
    const DvsRecord* rec;
    const float* xin;
    DvsLN ln;
    const float *node0_w, *node0_b, *node2_w, *node2_b, *edge0_w, *edge0_b, *edge2_w, *edge2_b;
    const void* wimg;            // loss image block (dvs_wimg.h: DvsLossImg); one-tile kernels
    float* dag_loss;             // [B][2]; NLL goes to [.][0]
    // backward only
    const float* gcoef;          // device [2]
    float* gout;                 // d pre of the last decoder sublayer
    float* slab;
    int64_t P;
    int64_t o_node0_w, o_node0_b, o_node2_w, o_node2_b, o_edge0_w, o_edge0_b, o_edge2_w, o_edge2_b, o_ln_g, o_ln_b;
};

struct FinalizeArgs {
    int B;
    float beta;
    const float* dag_loss;
    int* status;                 // optional validation word of the pack / build call (re-armed when host_tail is given)
    float* losses;               // [DVS_LOSS_FLOATS]
    float* host_tail;            // optional pinned host words [8] (include/dvs.h: dvs_loss_forward_notify)
    uint32_t host_seq;
};

struct BuildArgs {
    int B, N, C;
    const uint8_t* labels;       // [B][n]
    const uint16_t* preds;       // [B][n]
    DvsRecord* rec;
    int* status;
};
void dvs_launch_build_records(const BuildArgs& a, dvs_stream_t st);
void dvs_launch_pack(const PackArgs& a, dvs_stream_t st);
void dvs_launch_embed_fwd(const EmbedArgs& a, int grid, int nw, dvs_stream_t st);
// One phase of a chained forward launch (k_fwd_stack, k_forward.hip): attention or FFN sublayer of the one-tile path.
// DVS_FPH_LATENT (encoder chain only, last phase): the latent block on the 16 DAGs the workgroup owns (dvs_latent.h).
enum { DVS_FPH_ATTN = 0, DVS_FPH_FFN = 1, DVS_FPH_LATENT = 2 };
#define DVS_FWD_STACK_PHASES 9
struct FwdPhase {
    int kind, pad;
    union {
        AttnArgs a;
        FfnArgs f;
        LatentArgs l;
    } u;
};
struct FwdStackArgs {
    int nphase, pad;
    FwdPhase ph[DVS_FWD_STACK_PHASES];
    DvsStagePlan plan[DVS_FWD_STACK_PHASES];     // filled by dvs_launch_fwd_stack (dvs_stage.h)
};
static_assert(sizeof(FwdStackArgs) <= 4096, "kernel argument block limit");
// nw: waves per workgroup, 8 or 4 (dvs_api.hip: dvs_waves_per_wg); grid sized for nw DAGs per workgroup and pass
void dvs_launch_fwd_stack(const FwdStackArgs& s, int tag, int grid, int nw, dvs_stream_t st);   // tag 0 encoder, 1 decoder (profile names)
void dvs_launch_attn_fwd(const AttnArgs& a, int grid, int nw, dvs_stream_t st);
int dvs_attn_fwd_waves();
void dvs_launch_ffn_fwd(const FfnArgs& a, int grid, int nw, dvs_stream_t st);
void dvs_launch_latent_fwd(const LatentArgs& a, dvs_stream_t st);
void dvs_launch_loss_fwd(const LossArgs& a, int grid, int nw, dvs_stream_t st);
void dvs_launch_finalize(const FinalizeArgs& a, dvs_stream_t st);
void dvs_launch_unfrag(const float* frag, float* out, int B, dvs_stream_t st);

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg):
// dvs_profile_enable(1) ... launches ... dvs_profile_collect(): see include/dvs.h.
void dvs_prof_begin(const char* name, dvs_stream_t st);
void dvs_prof_end(dvs_stream_t st);
// A launch the runtime refuses (dynamic-LDS request above the limit, bad grid, ...) must not look like success: the
// first failure inside one entry-point call is kept per thread (dvs_note_hip_error) and becomes that call's return code
// and dvs_last_error() message (dvs_api.hip: call_begin / call_end).
void dvs_note_hip_error(const char* what, int hip_error, const char* hip_message);
// `name`: what the launch is called in the per-kernel timing table and in error messages (template instances that differ
// only in their workgroup width share one name)
#define DVS_LAUNCH_AS(name, kernel, grid, block, lds, st, ...)                                 \
    do {                                                                                       \
        dvs_prof_begin(name, st);                                                              \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                         \
        const hipError_t dvs_le_ = hipGetLastError();                                          \
        if (dvs_le_ != hipSuccess) dvs_note_hip_error(name, (int)dvs_le_, hipGetErrorString(dvs_le_)); \
        dvs_prof_end(st);                                                                      \
    } while (0)
#define DVS_LAUNCH(kernel, grid, block, lds, st, ...) DVS_LAUNCH_AS(#kernel, kernel, grid, block, lds, st, __VA_ARGS__)

#ifndef DVS_EMU
// Raise a kernel's dynamic-LDS limit once per (call site, device): hipFuncSetAttribute costs several microseconds of
// host time, which shows up as GPU idle at the start of every step (the host is not yet ahead of the device there).
// A refused request is recorded like a refused launch and is NOT cached, so the next call reports it again.
#define DVS_SET_LDS(kernel, bytes)                                                                                   \
    do {                                                                                                             \
        static size_t dvs_lds_set_[16] = {0};                                                                        \
        int dvs_dev_ = 0;                                                                                            \
        if (hipGetDevice(&dvs_dev_) != hipSuccess) dvs_dev_ = -1;                                                    \
        if (dvs_dev_ < 0 || dvs_dev_ >= 16 || dvs_lds_set_[dvs_dev_] < (size_t)(bytes)) {                            \
            const hipError_t dvs_ae_ = hipFuncSetAttribute((const void*)kernel,                                      \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (dvs_ae_ != hipSuccess)                                                                               \
                dvs_note_hip_error("hipFuncSetAttribute(" #kernel ", max dynamic LDS)", (int)dvs_ae_,               \
                                   hipGetErrorString(dvs_ae_));                                                      \
            else if (dvs_dev_ >= 0 && dvs_dev_ < 16)                                                                 \
                dvs_lds_set_[dvs_dev_] = (size_t)(bytes);                                                            \
        }                                                                                                            \
    } while (0)
#else
#define DVS_SET_LDS(kernel, bytes) ((void)0)
#endif

// ---- tiles ------------------------------------------------------------------------------------------------------
// Activation buffers hold B*NT frag-order tiles; tile `tile` is rows tok0 .. tok0+15 of DAG `dag`, of which the first
// Nl are real tokens.  NT == 1 (N <= 16): tile == dag.  The token-local kernels (FFN, projection backward) loop over
// tiles and are oblivious to which DAG a tile belongs to except for the dropout key and element index.
struct DvsTile {
    int dag, tok0, Nl;
};
__device__ __forceinline__ DvsTile dvs_tile_of(int tile, const DvsDims& d) {
    DvsTile t;
    if (d.NT == 1) {
        t.dag = tile;
        t.tok0 = 0;
        t.Nl = d.N;
    } else {
        t.dag = tile / d.NT;
        t.tok0 = 16 * (tile - t.dag * d.NT);
        const int nl = d.N - t.tok0;
        t.Nl = nl < 0 ? 0 : (nl > 16 ? 16 : nl);
    }
    return t;
}

__device__ __forceinline__ DvsDrop dvs_drop_of(const DvsDims& d) {
    DvsDrop D;
    D.thr16 = d.drop.thr16;
    D.scale = d.drop.scale;
    D.on = d.drop.on;
    return D;
}

// Prologue shared by every sublayer kernel: load the producer's pre-sum and apply its LayerNorm.
// Rows of padding tokens (r >= N) are forced to zero.  xhat (optional) receives the normalised value.
// Two halves, so that a chained phase can put its opening workgroup barrier BETWEEN them (DVS_PHASE_GATE below): `issue` is
// global loads only, `finish` is the first reader of the LayerNorm parameters in LDS.
struct DvsRawX {
    f4 raw[4];
    float mean, rs;
};
__device__ __forceinline__ void dvs_load_x_issue(DvsRawX& r, const float* xin, const DvsLN& ln, size_t dag, const Lane& L) {
    dvs_load_tile(r.raw, xin, dag, L);
    r.mean = 0.f;
    r.rs = 1.f;
    if (ln.stats != nullptr) {
        r.mean = ln.stats[dag * 32 + L.r];
        r.rs = ln.stats[dag * 32 + 16 + L.r];
    }
}
template <bool WANT_XHAT>
__device__ __forceinline__ void dvs_load_x_finish(f4 (&x)[4], f4 (&xhat)[4], float& rstd, const DvsRawX& r, const DvsLN& ln,
                                                  const float* lg, const float* lb, int N, const Lane& L) {
    // whole-vector arithmetic only (element-wise updates of the loaded vectors inside the branch made hipcc route
    // them through scratch memory)
    const float vm = L.r < N ? 1.f : 0.f;
    const bool has_ln = ln.stats != nullptr;
    rstd = r.rs;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 g = has_ln ? dvs_vecT(lg, t, L) : f4_splat(1.f);
        const f4 b = has_ln ? dvs_vecT(lb, t, L) : f4_zero();
        const f4 xh = (r.raw[t] - r.mean) * (r.rs * vm);
        if (WANT_XHAT) xhat[t] = has_ln ? xh : f4_zero();
        x[t] = xh * g + b * vm;
    }
}
template <bool WANT_XHAT>
__device__ __forceinline__ void dvs_load_x(f4 (&x)[4], f4 (&xhat)[4], float& rstd, const float* xin, const DvsLN& ln,
                                           const float* lg, const float* lb, size_t dag, int N, const Lane& L) {
    DvsRawX r;
    dvs_load_x_issue(r, xin, ln, dag, L);
    dvs_load_x_finish<WANT_XHAT>(x, xhat, rstd, r, ln, lg, lb, N, L);
}

// Opening barrier of a chained phase.  The previous phase's tail commits this phase's images and vectors to LDS and the
// workgroup barrier that publishes them used to sit between the two phase functions; every wave then started its first DAG
// with a cold global load behind that barrier.  Now the phase takes the barrier itself, in its FIRST DAG round, right behind
// that round's global loads (nothing in LDS has been touched yet — the loads are the only work ahead of it), so their latency
// overlaps the wait for the slowest wave of the previous phase and for the commit.  `pending` is false when the phase staged
// its own images (first phase of a launch, per-phase kernels); a workgroup without a DAG takes the barrier behind its loop.
// -DDVS_GATE_UPFRONT (A/B builds, tools/build_variant.sh): every phase takes the barrier before its DAG loop, as round 2 did.
#ifdef DVS_GATE_UPFRONT
#define DVS_PHASE_GATE_INIT(pending) DVS_PHASE_GATE(pending)
#else
#define DVS_PHASE_GATE_INIT(pending) ((void)0)
#endif
#define DVS_PHASE_GATE(pending)      \
    do {                             \
        if (pending) {               \
            dvs_lds_barrier();       \
            pending = false;         \
        }                            \
    } while (0)

// Epilogue: LayerNorm statistics of the new pre-sum; store tile + stats.
__device__ __forceinline__ void dvs_store_pre(float* out_pre, float* out_stats, size_t dag, const f4 (&pre)[4], const Lane& L) {
    float mean, rstd;
    dvs_ln_stats(pre, mean, rstd);
    dvs_store_tile(out_pre, dag, pre, L);
    if (L.g == 0) {
        out_stats[dag * 32 + L.r] = mean;
        out_stats[dag * 32 + 16 + L.r] = rstd;
    }
}

// ---- embedding selectors -----------------------------------------------------------------------------------------
// The positional encoder's first layer acts on one-hot rows, i.e. it gathers rows of W1 (pace.py:214-215):
//   e1pre[i] = W1[pos_i] + sum_{j parent of i} W1[N + pos_j]
// Written as products with two 0/1 selector matrices over positions p, Sel[p][i] = [pos_i == p] and
// Par[p][i] = [the vertex at position p is a parent of i], both the gather (forward) and its scatter (weight
// gradient) are MFMA chains: e1pre^T = W1a^T Sel + W1b^T Par ; dW1a = Sel de1 ; dW1b = Par de1.
// selB/parB: B-operand form [k <-> p = 4g+kk][col i = r]; selA/parA: A-operand form [row p = r][k <-> i = 4g+kk].
constexpr int EMB_LDW2 = 36;
struct EmbSel {
    f4 selB, parB, selA, parA, labA;   // labA: [k <-> i = 4g+kk][col c = r] = [label_i == c] (B-operand of the label scatter)
};
__device__ __forceinline__ EmbSel dvs_emb_selectors(const DvsRecord* rec, int N, float* scr, const Lane& L) {
    int* inv = (int*)scr;                              // inv[p] = vertex at position p
    if (L.g == 0) inv[L.r] = 0;
    dvs_wave_sync();
    const int pos_r = rec->pos[L.r];
    if (L.g == 0 && L.r < N) inv[pos_r] = L.r;
    dvs_wave_sync();
    const unsigned par_r = rec->parents[L.r];
    const int inv_r = inv[L.r];
    EmbSel s;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int p = 4 * L.g + kk;                    // as position index (B forms) and as token index (A forms)
        const bool vr = L.r < N, vp = p < N;
        s.selB[kk] = (vr && pos_r == p) ? 1.f : 0.f;
        s.parB[kk] = (vr && vp && ((par_r >> inv[p]) & 1u)) ? 1.f : 0.f;
        s.selA[kk] = (vp && rec->pos[p] == L.r) ? 1.f : 0.f;
        s.parA[kk] = (vp && vr && ((rec->parents[p] >> inv_r) & 1u)) ? 1.f : 0.f;
        s.labA[kk] = (vp && rec->label[p] == L.r) ? 1.f : 0.f;
    }
    dvs_wave_sync();
    return s;
}
// hidden of the positional encoder, T-layout [64 x tok], post-ReLU, before dropout.  W1: LDS image [32][DVS_LD] whose
// rows >= 2N are zero.
__device__ __forceinline__ void dvs_emb_hidden(f4 (&e1)[4], const float* W1, int N, const EmbSel& s, const Lane& L) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) e1[ct] = f4_zero();
    f4 wa[4], wb[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        wa[ct] = dvs_wcol(W1, DVS_LD, 16 * ct, 0, L);
        wb[ct] = dvs_wcol(W1 + N * DVS_LD, DVS_LD, 16 * ct, 0, L);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            e1[ct] = dvs_mfma(wa[ct][kk], s.selB[kk], e1[ct]);
            e1[ct] = dvs_mfma(wb[ct][kk], s.parB[kk], e1[ct]);
        }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) e1[ct][kk] = fmaxf(e1[ct][kk], 0.f);
}
