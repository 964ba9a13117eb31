// Kernel argument blocks + launcher prototypes (one launcher per kernel; all enqueue on the given stream).
#pragma once
#include "dvs_device.h"
#include "dvs_host.h"

#ifndef DVS_EMU
typedef hipStream_t dvs_stream_t;
#else
typedef void* dvs_stream_t;
#endif

struct DvsLN {                   // LayerNorm applied in a consumer's prologue: x = (pre - mean) * rstd * g + b
    const float* stats;          // [B][32] (mean[16], rstd[16]); null = identity (embedding output)
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
    float* out;                  // [B][1024] frag order
    int site;                    // dropout site of the first dropout (second = site + 1)
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
    float* out_pre;              // x + dropout(attn(x))
    float* out_stats;
    int site_prob, site_post;
};

struct FfnArgs {
    DvsDims dims;
    const float* xin;
    DvsLN ln;
    const float *l1_w, *l1_b, *l2_w, *l2_b;
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
    float* losses;               // [4]
};

void dvs_launch_pack(const PackArgs& a, dvs_stream_t st);
void dvs_launch_embed_fwd(const EmbedArgs& a, int grid, dvs_stream_t st);
void dvs_launch_attn_fwd(const AttnArgs& a, int grid, dvs_stream_t st);
void dvs_launch_ffn_fwd(const FfnArgs& a, int grid, dvs_stream_t st);
void dvs_launch_latent_fwd(const LatentArgs& a, dvs_stream_t st);
void dvs_launch_loss_fwd(const LossArgs& a, int grid, dvs_stream_t st);
void dvs_launch_finalize(const FinalizeArgs& a, dvs_stream_t st);
void dvs_launch_unfrag(const float* frag, float* out, int B, dvs_stream_t st);

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg):
// dvs_profile_enable(1) ... launches ... dvs_profile_collect(): see include/dvs.h.
void dvs_prof_begin(const char* name, dvs_stream_t st);
void dvs_prof_end(dvs_stream_t st);
#define DVS_LAUNCH(kernel, grid, block, lds, st, ...)                      \
    do {                                                                   \
        dvs_prof_begin(#kernel, st);                                       \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);     \
        dvs_prof_end(st);                                                  \
    } while (0)

#ifndef DVS_EMU
#define DVS_SET_LDS(kernel, bytes) \
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
#else
#define DVS_SET_LDS(kernel, bytes) ((void)0)
#endif

__device__ __forceinline__ DvsDrop dvs_drop_of(const DvsDims& d) {
    DvsDrop D;
    D.thr16 = d.drop.thr16;
    D.scale = d.drop.scale;
    D.on = d.drop.on;
    return D;
}

// Prologue shared by every sublayer kernel: load the producer's pre-sum and apply its LayerNorm.
// Rows of padding tokens (r >= N) are forced to zero.  xhat (optional) receives the normalised value.
template <bool WANT_XHAT>
__device__ __forceinline__ void dvs_load_x(f4 (&x)[4], f4 (&xhat)[4], float& rstd, const float* xin, const DvsLN& ln,
                                           const float* lg, const float* lb, size_t dag, int N, const Lane& L) {
    dvs_load_tile(x, xin, dag, L);
    const bool valid = L.r < N;
    rstd = 1.f;
    if (ln.stats) {
        const float mean = ln.stats[dag * 32 + L.r];
        rstd = ln.stats[dag * 32 + 16 + L.r];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f4 g = dvs_vecT(lg, t, L), b = dvs_vecT(lb, t, L);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float xh = valid ? (x[t][kk] - mean) * rstd : 0.f;
                if (WANT_XHAT) xhat[t][kk] = xh;
                x[t][kk] = valid ? xh * g[kk] + b[kk] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                x[t][kk] = valid ? x[t][kk] : 0.f;
                if (WANT_XHAT) xhat[t][kk] = 0.f;
            }
    }
}

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
