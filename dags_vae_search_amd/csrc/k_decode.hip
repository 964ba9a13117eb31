// Batched generation: PaceVaeV3.decode (pace.py:1666-1749) as a device-resident autoregressive loop.
//
// The reference grows one igraph object per DAG on the host: per step it rebuilds dense features from the graph
// objects (prepare_features_v2, 1480-1611), runs the decoder, pulls probabilities to the host, samples with
// np.random.choice / torch.rand_like, mutates the graphs and re-sorts them topologically (~26 s per 32-DAG batch of
// 10x10 decodes at n = 12, experiments/03_synthetic_12/main.py:236-239).  Here the grown graphs live on the GPU as
// bit rows (DvsDecodeState); each of the N-2 steps is: records of the partial graphs -> the SAME embedding / decoder
// kernels as the train step (eval mode) -> k_decode_step: node-type and edge probabilities of the newest vertex,
// sampling, graph update, and the next step's record (FIFO-Kahn positions, ancestor closure) — no host round trip.
//
// Padding tokens of a partial graph (label `output`, position nv) attend only themselves here (the reference lets them
// attend each other): no real token can attend a padding token in either, so the rows that are read are identical.
#include "dvs_wide.h"

#include "dvs_decode.h"

// per-wave scratch for the record build (lane 0) and the sampling (all lanes)
struct DecScratch {
    uint64_t child[DVS_WTOK], reach[DVS_WTOK];
    unsigned char indeg[DVS_WTOK], order[DVS_WTOK];
    float hi[64], hv[64], t1[32], p[DVS_WTOK], score[DVS_WTOK];
};

// record of a partial graph: real tokens as in k_build_records, padding tokens isolated
template <class Rec, class Row>
__device__ void decode_build_record(const DvsDecodeState* S, Rec* rec, int N, DecScratch* sc, int slots) {
    const int nv = S->nv;
    for (int v = 0; v < nv; ++v) sc->child[v] = 0;
    for (int i = 0; i < nv; ++i)
        for (uint64_t m = S->parents[i]; m; m &= m - 1) sc->child[dvs_ctz64(m)] |= 1ull << i;
    for (int v = 0; v < nv; ++v) sc->indeg[v] = (unsigned char)__popcll(S->parents[v]);
    int tail = 0;
    for (int v = 0; v < nv; ++v)
        if (sc->indeg[v] == 0) sc->order[tail++] = (unsigned char)v;
    for (int head = 0; head < tail && head < nv; ++head)
        for (uint64_t m = sc->child[sc->order[head]]; m; m &= m - 1) {
            const int v = dvs_ctz64(m);
            if (--sc->indeg[v] == 0 && tail < DVS_WTOK) sc->order[tail++] = (unsigned char)v;
        }
    for (int v = 0; v < nv; ++v) sc->reach[v] = sc->child[v] | (1ull << v);
    for (int k = 0; k < nv; ++k) {
        const uint64_t rk = sc->reach[k];
        for (int v = 0; v < nv; ++v)
            if ((sc->reach[v] >> k) & 1ull) sc->reach[v] |= rk;
    }
    for (int i = 0; i < slots; ++i) {
        int label = 0, pos = 0;
        uint64_t parents = 0, allowed = 1ull << i;
        if (i < nv) {
            label = S->label[i];
            pos = sc->order[i];
            parents = S->parents[i];
            allowed = 0;
            for (int j = 0; j < nv; ++j)
                if ((sc->reach[j] >> i) & 1ull) allowed |= 1ull << j;
        } else if (i < N) {
            label = DEC_OUT;
            pos = nv;
        }
        rec->label[i] = (uint8_t)label;
        rec->pos[i] = (uint8_t)pos;
        rec->parents[i] = (Row)parents;
        rec->allowed[i] = (Row)allowed;
    }
}

__device__ __forceinline__ void decode_record(const DecodeArgs& a, int dag, DecScratch* sc) {
    if (a.wide) decode_build_record<DvsRecordW, uint64_t>(a.state + dag, (DvsRecordW*)a.rec + dag, a.dims.N, sc, DVS_WTOK);
    else decode_build_record<DvsRecord, uint16_t>(a.state + dag, (DvsRecord*)a.rec + dag, a.dims.N, sc, 16);
}

__global__ __launch_bounds__(256) void k_decode_init(DecodeArgs a) {
    __shared__ DecScratch scratch[4];
    const Lane L = dvs_lane();
    const int dag = blockIdx.x * 4 + L.wave;
    if (dag >= a.dims.B || L.lane != 0) return;
    DvsDecodeState* S = a.state + dag;
    for (int i = 0; i < DVS_WTOK; ++i) {
        S->parents[i] = 0;
        S->label[i] = 0;
    }
    S->label[0] = 2;             // graph_label_start (pace.py:1681)
    S->label[1] = 0;             // graph_label_input (1685)
    S->nv = 2;
    S->finished = 0;
    decode_record(a, dag, &scratch[L.wave]);
}

// hidden row `tok` of DAG `dag` after the final LayerNorm, feature = lane
__device__ __forceinline__ float decode_hidden(const DecodeArgs& a, const float* lg, const float* lb, int dag, int tok, int f) {
    const size_t tile = (size_t)dag * a.dims.NT + (tok >> 4);
    const int r = tok & 15, t = f >> 4, g = (f >> 2) & 3, kk = f & 3;
    const float pre = a.xin[(tile * 4 + t) * 256 + (16 * g + r) * 4 + kk];
    const float mean = a.ln.stats[tile * 32 + r], rstd = a.ln.stats[tile * 32 + 16 + r];
    return (pre - mean) * rstd * lg[f] + lb[f];
}

__device__ __forceinline__ float decode_uniform(const DecodeArgs& a, int dag, int idx, int k) {
    if (a.uniforms) return a.uniforms[((size_t)dag * a.dims.N + idx) * a.dims.N + k];
    const uint32_t key = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, 200u, a.dims.dag_offset + dag);
    return (float)(dvs_draw(key, (uint32_t)(idx * 64 + k)) >> 8) * (1.0f / 16777216.0f);
}

// Weights as TRANSPOSED LDS images (WT[k][out]) so that lane = output feature reads consecutive words.
__global__ __launch_bounds__(256) void k_decode_step(DecodeArgs a) {
    DVS_DYN_LDS(smem);
    const int C = a.dims.C, N = a.dims.N, idx = a.idx;
    float* Wn1T = (float*)smem;                  // [64][32]
    float* Wn2T = Wn1T + 64 * 32;                // [32][48]
    float* WaT = Wn2T + 32 * DVS_WTOK;           // [64][64]
    float* WbT = WaT + 64 * 64;                  // [64][64]
    float* vec = WbT + 64 * 64;                  // bn1[32] bn2[48] be1[64] w2[64] lg[64] lb[64] b2
    DecScratch* scratch = (DecScratch*)(vec + 32 + DVS_WTOK + 64 * 4 + 16);
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) Wn1T[(i & 63) * 32 + (i >> 6)] = a.node0_w[i];
    for (int i = threadIdx.x; i < DVS_WTOK * 32; i += blockDim.x) {
        const int c = i >> 5, k = i & 31;
        Wn2T[k * DVS_WTOK + c] = c < C ? a.node2_w[c * 32 + k] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int f = i >> 6, k = i & 63;
        WaT[k * 64 + f] = a.edge0_w[f * 128 + k];
        WbT[k * 64 + f] = a.edge0_w[f * 128 + 64 + k];
    }
    float *bn1 = vec, *bn2 = bn1 + 32, *be1 = bn2 + DVS_WTOK, *w2 = be1 + 64, *lg = w2 + 64, *lb = lg + 64, *b2 = lb + 64;
    for (int i = threadIdx.x; i < 32; i += blockDim.x) bn1[i] = a.node0_b[i];
    for (int i = threadIdx.x; i < DVS_WTOK; i += blockDim.x) bn2[i] = i < C ? a.node2_b[i] : 0.f;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        be1[i] = a.edge0_b[i];
        w2[i] = a.edge2_w[i];
        lg[i] = a.ln.g[i];
        lb[i] = a.ln.b[i];
    }
    if (threadIdx.x == 0) b2[0] = a.edge2_b[0];
    __syncthreads();
    const Lane L = dvs_lane();
    DecScratch* sc = scratch + L.wave;
    const int f = L.lane;
    for (int dag = blockIdx.x * 4 + L.wave; dag < a.dims.B; dag += gridDim.x * 4) {
        DvsDecodeState* S = a.state + dag;
        if (S->finished) continue;                       // wave-uniform: a finished graph is never touched again
        // ---- node type of the new vertex from the hidden of vertex idx-1 (pace.py:1707-1713) -------------------------
        sc->hi[f] = decode_hidden(a, lg, lb, dag, idx - 1, f);
        dvs_wave_sync();
        if (f < 32) {
            float s = bn1[f];
            for (int k = 0; k < 64; ++k) s = fmaf(Wn1T[k * 32 + f], sc->hi[k], s);
            sc->t1[f] = fmaxf(s, 0.f);
        }
        dvs_wave_sync();
        float logit = -3.0e38f;
        if (f < C) {
            logit = bn2[f];
            for (int k = 0; k < 32; ++k) logit = fmaf(Wn2T[k * DVS_WTOK + f], sc->t1[k], logit);
        }
        float m2 = logit;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) m2 = fmaxf(m2, __shfl_xor(m2, sh));
        const float ex = f < C ? __expf(logit - m2) : 0.f;
        const float se = dvs_sum_wave(ex);
        if (f < DVS_WTOK) sc->p[f] = ex / se;
        // ---- edge scores of (new vertex <- vertex vi+1), vi = 0 .. idx-2 (pace.py:1716-1717) -----------------------
        float U = 0.f;
        for (int k = 0; k < 64; ++k) U = fmaf(WaT[k * 64 + f], sc->hi[k], U);
        for (int vi = 0; vi <= idx - 2; ++vi) {
            sc->hv[f] = decode_hidden(a, lg, lb, dag, vi, f);
            dvs_wave_sync();
            float V = be1[f];
            for (int k = 0; k < 64; ++k) V = fmaf(WbT[k * 64 + f], sc->hv[k], V);
            const float e = dvs_sum_wave(w2[f] * fmaxf(U + V, 0.f)) + b2[0];
            if (f == 0) sc->score[vi] = 1.0f / (1.0f + __expf(-e));
            dvs_wave_sync();
        }
        dvs_wave_sync();
        // ---- sampling + graph update (lane 0; pace.py:1712, 1719-1741) ---------------------------------------------
        if (f == 0) {
            // np.random.choice: cdf = cumsum(p) / sum; first index with cdf > u
            double tot = 0.0;
            for (int c = 0; c < C; ++c) tot += (double)sc->p[c];
            const double u = (double)decode_uniform(a, dag, idx, 0) * tot;
            double run = 0.0;
            int new_type = C - 1;
            for (int c = 0; c < C; ++c) {
                run += (double)sc->p[c];
                if (run > u) { new_type = c; break; }
            }
            const int nw = S->nv;
            S->label[nw] = (uint8_t)(idx < N - 1 ? new_type : DEC_OUT);
            S->nv = nw + 1;
            uint64_t par = 0;
            if (new_type == DEC_OUT) {
                uint64_t has_out = 0;
                for (int i = 0; i < nw; ++i) has_out |= S->parents[i];
                par = ~has_out & ((1ull << nw) - 1ull);          // every loose end feeds the output vertex
                S->finished = 1;
            } else {
                for (int vi = idx - 2; vi >= 0; --vi)
                    if (decode_uniform(a, dag, idx, 1 + vi) < sc->score[vi]) par |= 1ull << (vi + 1);
            }
            S->parents[nw] = par;
            decode_record(a, dag, sc);
        }
        dvs_wave_sync();
    }
}

// memory = fc3(z) (pace.py:1675), written as frag-order tiles
__global__ __launch_bounds__(256) void k_decode_memory(DvsDims d, const float* z, const float* w, const float* b, float* mem) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_dag = (size_t)d.NT * 1024;
    if (i >= (size_t)d.B * per_dag) return;
    const size_t dag = i / per_dag;
    const int e = (int)(i - dag * per_dag);
    const int tk = e >> 10, q = e & 1023;
    const int t = q >> 8, lane = (q >> 2) & 63, kk = q & 3;
    const int tok = 16 * tk + (lane & 15), f = 16 * t + 4 * (lane >> 4) + kk;
    float s = 0.f;
    if (tok < d.N) {
        const float* wr = w + ((size_t)tok * 64 + f) * 32;
        s = b[tok * 64 + f];
        for (int o = 0; o < 32; ++o) s = fmaf(wr[o], z[dag * 32 + o], s);
    }
    mem[i] = s;
}

size_t dvs_decode_lds_bytes() {
    return (64 * 32 + 32 * DVS_WTOK + 2 * 64 * 64 + 32 + DVS_WTOK + 64 * 4 + 16) * sizeof(float) + 4 * sizeof(DecScratch);
}
void dvs_launch_decode_init(const DecodeArgs& a, dvs_stream_t st) {
    DVS_LAUNCH(k_decode_init, dim3((a.dims.B + 3) / 4), dim3(256), 0, st, a);
}
void dvs_launch_decode_step(const DecodeArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = dvs_decode_lds_bytes();
    DVS_SET_LDS(k_decode_step, lds);
    DVS_LAUNCH(k_decode_step, dim3(grid), dim3(256), lds, st, a);
}
void dvs_launch_decode_memory(const DvsDims& d, const float* z, const float* w, const float* b, float* mem, dvs_stream_t st) {
    const size_t n = (size_t)d.B * d.NT * 1024;
    DVS_LAUNCH(k_decode_memory, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, z, w, b, mem);
}
