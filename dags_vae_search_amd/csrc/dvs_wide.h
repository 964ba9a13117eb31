// "Wide" path: DAGs of 17..48 tokens (alarm-size, n = 37 -> N = 40; BASELINE config 5).
//
// A DAG is NT = ceil(N/16) consecutive frag-order tiles of 16 tokens.  Everything token-local (the 64x64 linears,
// LayerNorm, dropout, residuals: k_ffn_*, k_proj_bwd, the latent GEMMs) runs on the SAME kernels as the one-tile path,
// looping over tiles (dvs_tile_of).  The three cross-token operations get workgroup-per-DAG kernels here: wave w of a
// 4-wave workgroup owns tile w (its MFMA chains are those of the one-tile kernels), and the tiles of a DAG meet in LDS:
//   * positional embedding: every token gathers the W1 rows of its parents' positions from an LDS image of W1
//     (pace.py:214 adj^T @ pos_onehot) — the parent-hidden gather BASELINE config 5 stresses;
//   * attention core (8-wave workgroups): q, k, v of all tiles parked row-major in LDS, wave h = head h on the fp32 matrix pipe
//     over the non-empty (query tile, key tile) pairs of the DAG's mask; the backward reads q, k, v back from the forward,
//     walks the query tiles once (dS / P' tiles transposed through a per-wave LDS scratch) and keeps two DAGs in flight per
//     workgroup (k_wide_fwd.hip, k_wide_bwd.hip);
//   * edge-pair head: U, V of all tokens parked in LDS, lane (token i) walks j < i; the walk steps of all tiles are shared
//     evenly by the four waves.
// Workgroups are persistent (grid = #CU) so that weight gradients keep the slab scheme of dvs_backward.h.
#pragma once
#include "dvs_backward.h"
#include "dvs_bf16.h"

constexpr int DVS_WNT = 3;                       // tiles per DAG at most
constexpr int DVS_WSCR = DVS_WTOK * DVS_LD;      // floats of a [48][DVS_LD] row-major DAG tile set in LDS

struct DvsRecordW {              // compact per-DAG record of the wide path (864 bytes)
    uint8_t label[DVS_WTOK];
    uint8_t pos[DVS_WTOK];
    uint64_t parents[DVS_WTOK];  // bit j: edge j -> i
    uint64_t allowed[DVS_WTOK];  // bit j: token i may attend token j
};

__device__ __forceinline__ int dvs_ctz64(uint64_t v) { return __builtin_ctzll(v); }

// rows of real tokens in tile `w` of an N-token DAG
__device__ __forceinline__ int dvs_rows_of(int N, int w) {
    const int nl = N - 16 * w;
    return nl < 0 ? 0 : (nl > 16 ? 16 : nl);
}

// T-layout / N-layout register tiles of rows tok0.. of a row-major [tokens][DVS_LD] LDS buffer
__device__ __forceinline__ void dvs_lds_T(f4 (&x)[4], const float* buf, int tok0, const Lane& L) {
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = *(const f4*)(buf + (tok0 + L.r) * DVS_LD + 16 * t + 4 * L.g);
}
__device__ __forceinline__ void dvs_lds_N(f4 (&x)[4], const float* buf, int tok0, const Lane& L) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* p = buf + (tok0 + 4 * L.g) * DVS_LD + 16 * t + L.r;
        x[t] = f4{p[0], p[DVS_LD], p[2 * DVS_LD], p[3 * DVS_LD]};
    }
}

// ---- LDS images shared by the forward and backward kernels ---------------------------------------------------------
constexpr int EMBW_LABLD = DVS_WTOK;             // labw image [32][48]
struct EmbWLds {
    float *W1, *W2, *labw, *labb;
    uint32_t* posl;              // [4 waves][16]: the positions (bytes) of the DAG a wave works on (embw_hidden)
};
__device__ __forceinline__ EmbWLds embw_lds(char* smem) {
    EmbWLds l;
    l.W1 = (float*)smem;                         // [2*48][LD]
    l.W2 = l.W1 + 2 * DVS_WTOK * DVS_LD;         // [64][36]
    l.labw = l.W2 + 64 * EMB_LDW2;               // [32][48]
    l.labb = l.labw + 32 * EMBW_LABLD;
    l.posl = (uint32_t*)(l.labb + 32);
    return l;
}
constexpr size_t EMBW_FLOATS = 2 * DVS_WTOK * DVS_LD + 64 * EMB_LDW2 + 32 * EMBW_LABLD + 32 + 4 * 16;

__device__ __forceinline__ void embw_stage(const EmbWLds& l, const EmbedArgs& a) {
    const int N = a.dims.N, C = a.dims.C;
    dvs_stage_matrix(l.W1, DVS_LD, a.W1, 64, 2 * N, 64);
    dvs_stage_matrix(l.W2, EMB_LDW2, a.W2, 32, 64, 32);
    for (int i = threadIdx.x; i < 32 * EMBW_LABLD; i += blockDim.x) {
        const int f = i / EMBW_LABLD, c = i - f * EMBW_LABLD;
        l.labw[i] = c < C ? a.lab_w[f * C + c] : 0.f;
    }
    dvs_stage_vector(l.labb, a.lab_b, 32);
}

// hidden of the positional encoder for token tok0 + r (T-layout), post-ReLU, before dropout; zero for padding rows
// posl: this wave's 16 words of EmbWLds::posl — the DAG's positions are copied there first: the walk below reads one per parent,
// and from global memory every one of them was a dependent load in the chain
__device__ __forceinline__ void embw_hidden(f4 (&e1)[4], const float* W1, const DvsRecordW* rec, uint32_t* posl, int N, int tok0,
                                            int Nl, const Lane& L) {
    const bool valid = L.r < Nl;
    const int i = valid ? tok0 + L.r : 0;
    uint64_t pm = valid ? rec->parents[i] : 0ull;
    dvs_wave_sync();             // (the wave's previous use of posl is over)
    if (L.lane < DVS_WTOK / 4) posl[L.lane] = ((const uint32_t*)rec->pos)[L.lane];
    dvs_wave_sync();
    const uint8_t* pos = (const uint8_t*)posl;
    const float* row = W1 + pos[i] * DVS_LD + 4 * L.g;
#pragma unroll
    for (int t = 0; t < 4; ++t) e1[t] = valid ? *(const f4*)(row + 16 * t) : f4_zero();
    while (pm) {
        const int j = dvs_ctz64(pm);
        pm &= pm - 1;
        const float* prow = W1 + (N + pos[j]) * DVS_LD + 4 * L.g;
#pragma unroll
        for (int t = 0; t < 4; ++t) e1[t] += *(const f4*)(prow + 16 * t);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) e1[t][kk] = fmaxf(e1[t][kk], 0.f);
}

constexpr int LOSSW_LDN2 = 36;
struct LossWLds {
    float *Wn1, *Wn2, *Wa, *Wb, *bn1, *bn2, *be1, *w2, *b2, *lg, *lb, *V, *U, *dlm, *part, *scr, *pU, *pV;
};
__device__ __forceinline__ LossWLds lossw_lds(char* smem) {
    LossWLds l;
    l.Wn1 = (float*)smem;                       // [32][LD]
    l.Wn2 = l.Wn1 + 32 * DVS_LD;                // [48][36], rows >= C zero
    l.Wa = l.Wn2 + DVS_WTOK * LOSSW_LDN2;
    l.Wb = l.Wa + 64 * DVS_LD;
    l.bn1 = l.Wb + 64 * DVS_LD;                 // 32
    l.bn2 = l.bn1 + 32;                         // 48
    l.be1 = l.bn2 + DVS_WTOK;                   // 64
    l.w2 = l.be1 + 64;                          // 64
    l.b2 = l.w2 + 64;                           // 16
    l.lg = l.b2 + 16;
    l.lb = l.lg + 64;
    l.V = l.lb + 64;                            // [48][LD] shared
    l.U = l.V + DVS_WSCR;                       // [48][LD] shared (backward)
    l.dlm = l.U + DVS_WSCR;                     // [48][49] d logit matrix (backward)
    l.part = l.dlm + DVS_WTOK * (DVS_WTOK + 1); // 16
    l.scr = l.part + 16;                        // 4 per-wave transpose tiles (backward); then the parked h tiles of the DAG
    l.pU = l.scr + 4 * DVS_SCR;                 // 4 blocks of 16 rows: parked dU tiles as bf16 [hi | lo] pairs (backward)
    l.pV = l.pU + 4 * DVS_SCR;                  // likewise dV
    return l;
}
// forward: the images, V, U and 16 floats of per-wave partial sums (in the first words of what is the d logit matrix in the
// backward) — 78 KB, two workgroups per CU (one wave per SIMD: the kernel is latency-bound)
static inline size_t dvs_lossw_lds_floats(bool backward = true) {
    const size_t head = 32 * DVS_LD + DVS_WTOK * LOSSW_LDN2 + 128 * DVS_LD + 32 + DVS_WTOK + 64 + 64 + 16 + 128;
    if (!backward) return head + 2 * (size_t)DVS_WSCR + 16;
    return head + 2 * (size_t)DVS_WSCR + DVS_WTOK * (DVS_WTOK + 1) + 16 + 12 * (size_t)DVS_SCR;
}
__device__ __forceinline__ void lossw_stage(const LossWLds& l, const LossArgs& a, bool backward = true) {
    const int C = a.dims.C;
    dvs_stage_matrix(l.Wn1, DVS_LD, a.node0_w, 64, 32, 64);
    for (int i = threadIdx.x; i < DVS_WTOK * 32; i += blockDim.x) {
        const int c = i >> 5, k = i & 31;
        l.Wn2[c * LOSSW_LDN2 + k] = c < C ? a.node2_w[c * 32 + k] : 0.f;
    }
    dvs_stage_matrix(l.Wa, DVS_LD, a.edge0_w, 128, 64, 64);
    dvs_stage_matrix(l.Wb, DVS_LD, a.edge0_w + 64, 128, 64, 64);
    dvs_stage_vector(l.bn1, a.node0_b, 32);
    for (int i = threadIdx.x; i < DVS_WTOK; i += blockDim.x) l.bn2[i] = i < C ? a.node2_b[i] : 0.f;
    dvs_stage_vector(l.be1, a.edge0_b, 64);
    dvs_stage_vector(l.w2, a.edge2_w, 64);
    if (threadIdx.x == 0) l.b2[0] = a.edge2_b[0];
    dvs_stage_vector(l.lg, a.ln.g, 64);
    dvs_stage_vector(l.lb, a.ln.b, 64);
    for (int i = threadIdx.x; i < 2 * DVS_WSCR; i += blockDim.x) l.V[i] = 0.f;
}
// backward only: the per-wave transpose tiles and the parked dU / dV blocks start as zeros (blocks of tiles the DAG does not
// have — tile 3 always — are never written and are contracted as they are)
__device__ __forceinline__ void lossw_zero_parks(const LossWLds& l) {
    for (int i = threadIdx.x; i < 12 * DVS_SCR; i += blockDim.x) l.scr[i] = 0.f;
}

struct BuildWArgs {
    int B, N, C;
    const uint8_t* labels;       // [B][n]
    const uint64_t* preds;       // [B][n]
    DvsRecordW* rec;
    int* status;
};

void dvs_launch_pack_w(const PackArgs& a, dvs_stream_t st);
void dvs_launch_build_records_w(const BuildWArgs& a, dvs_stream_t st);
void dvs_launch_embed_fwd_w(const EmbedArgs& a, int grid, dvs_stream_t st);
void dvs_launch_attn_fwd_w(const AttnArgs& a, int grid, dvs_stream_t st);
void dvs_launch_loss_fwd_w(const LossArgs& a, int grid, dvs_stream_t st);
void dvs_launch_embed_bwd_w(const EmbedArgs& a, const float* gout2, int site2, int grid, dvs_stream_t st);
void dvs_launch_attn_bwd_w(const AttnBwdArgs& a, int grid, dvs_stream_t st);
void dvs_launch_loss_bwd_w(const LossArgs& a, int grid, dvs_stream_t st);

// Cooperative weight gradient of ONE DAG of up to 3 tiles (dvs_coop_dw_bf, dvs_bf16.h, restated): acc (rows 16 (wave & 3) ..
// of dW) += sum over the DAG's tiles of dY^T X and accb += column sums of dY, from the tiles parked as bf16 [hi | lo] pairs in
// consecutive 16-row blocks (2 DVS_SCR bf16 each) of `abase` (dY) and `bbase` (X).  The K = 32 blocks are the tile pairs
// (0, 1) and (2, 3); tiles >= NT (tile 3 always: it lies behind the buffers) are contracted against zeros.  Four waves share a
// product: 16 accumulator registers per matrix instead of the 64 of a per-wave outer product.
__device__ __forceinline__ void dvsw_coop_dw(f4 (&acc)[4], f4& accb, const dvs_bf16* abase, const dvs_bf16* bbase, int NT,
                                                const Lane& L) {
    const int ot = L.wave & 3;
    constexpr int STRIDE = 2 * DVS_SCR;               // bf16 elements per 16-row block ([hi | lo] pair)
    bf8 ones, zero;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ones[i] = (dvs_bf16)1.0f;
        zero[i] = (dvs_bf16)0.0f;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int d = 2 * m + (L.g >> 1);
        const bool real = d < NT;                     // tiles >= NT: never parked (tile 3: outside the buffers)
        const int dc = real ? d : 0;
        const dvs_bf16* ta = abase + dc * STRIDE;
        const dvs_bf16* tb = bbase + dc * STRIDE;
        bf8 ah = dvs_tr_frag(ta, 16 * ot, L), al = dvs_tr_frag(ta + DVS_PKB, 16 * ot, L);
        ah = real ? ah : zero;
        al = real ? al : zero;
        accb = dvs_mfma_bf(al, ones, accb);
        accb = dvs_mfma_bf(ah, ones, accb);
        bf8 bh[4], bl[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            bh[it] = dvs_tr_frag(tb, 16 * it, L);
            bl[it] = dvs_tr_frag(tb + DVS_PKB, 16 * it, L);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = dvs_mfma_bf(al, bh[it], acc[it]);
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = dvs_mfma_bf(ah, bl[it], acc[it]);
#pragma unroll
        for (int it = 0; it < 4; ++it) acc[it] = dvs_mfma_bf(ah, bh[it], acc[it]);
        DVS_SCHED_FENCE();
    }
}


// one output tile (16 features ot) of a T-layout product, parked row-major: rows tok0 + r, columns 16 ot + 4g ..
__device__ __forceinline__ void dvs_park_col(float* buf, int tok0, int ot, const f4& v, const Lane& L) {
    *(f4*)(buf + (tok0 + L.r) * DVS_LD + 16 * ot + 4 * L.g) = v;
}
