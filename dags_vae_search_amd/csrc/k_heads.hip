// Latent block (fc1/fc2 -> reparameterize -> KL -> fc3), loss head (node NLL + edge BCE) and loss finalisation.
#include "dvs_kernels.h"
#include "dvs_wimg.h"
#include "dvs_latent.h"

// ---------------------------------------------------------------------------------------------------------
// Latent block forward (pace.py:1639-1641, 1649-1664, 1997, 2030).  One wave owns 16 DAGs:
//   out^T[64 x 16 dags] = [fc1;fc2][64 x N*64] * Xenc^T     (contraction walks the frag-order tile in 16-float
//                                                            chunks; chunk (m,g) is 4 consecutive features of
//                                                            token 4(m&3)+g, so weights are read in place)
//   z^T = mu^T + eps * exp(logvar/2);  KL per DAG;  mem^T[N*64 x 16 dags] = fc3 * z^T + b3 (stored frag order)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * LAT_WAVES) void k_latent_fwd(LatentArgs a) {
    DVS_DYN_LDS(smem);
    dvs_latent_fwd_group(a, smem, (int)blockIdx.x * 16, (int)blockIdx.x * 16 + 8);
}

void dvs_launch_latent_fwd(const LatentArgs& a, dvs_stream_t st) {
    const int ngroups = (a.dims.B + 15) / 16;
    const size_t lds = (size_t)LAT_WAVES * 4 * 64 * sizeof(f4);
    DVS_SET_LDS(k_latent_fwd, lds);
    DVS_LAUNCH(k_latent_fwd, dim3(ngroups), dim3(64 * LAT_WAVES), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Loss head forward (loss_log_likelihood_full_vectorized, pace.py:1880-1972).
//   node: log_softmax(add_node(h))[i, label_{i+1}] for i < N-1
//   edge: logit(i,j) = w2 . relu(Wa h_i + Wb h_j + b1) + b2 for j < i <= N-2 (W1 cat(h_i,h_j) split into Wa|Wb),
//         truth = adj[j+1][i+1]; BCE-with-logits, summed.
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_LDN2 = 36;
struct LossLds {
    dvs_bf16 *Wa, *Wb;           // bf16x6 image triples of the two halves of add_edge.0.weight (dvs_wimg.h)
    float *Wn1, *Wn2, *bn1, *bn2, *be1, *w2, *b2, *lg, *lb, *scr;
};
__device__ __forceinline__ LossLds loss_lds(char* smem) {
    LossLds l;
    l.Wa = (dvs_bf16*)smem;
    l.Wb = l.Wa + 3 * DVS_IMG64;
    l.Wn1 = (float*)(l.Wb + 3 * DVS_IMG64);
    l.Wn2 = l.Wn1 + 32 * DVS_LD;
    l.bn1 = l.Wn2 + 16 * LOSS_LDN2;
    l.bn2 = l.bn1 + 32;
    l.be1 = l.bn2 + 16;
    l.w2 = l.be1 + 64;
    l.b2 = l.w2 + 64;
    l.lg = l.b2 + 16;
    l.lb = l.lg + 64;
    l.scr = l.lb + 64;
    return l;
}
size_t dvs_loss_lds_floats(int nwaves, int tiles_per_wave) {      // forward layout: the two x6 image triples first
    return 6 * DVS_IMG64 / 2 + 32 * DVS_LD + 16 * LOSS_LDN2 + 32 + 16 + 64 + 64 + 16 + 128 + (size_t)nwaves * tiles_per_wave * DVS_SCR;
}

__device__ __forceinline__ void loss_stage(const LossLds& l, const LossArgs& a) {
    const int C = a.dims.C;
    dvs_stage_matrix(l.Wn1, DVS_LD, a.node0_w, 64, 32, 64);
    for (int i = threadIdx.x; i < 16 * 32; i += blockDim.x) {
        const int c = i >> 5, k = i & 31;
        l.Wn2[c * LOSS_LDN2 + k] = c < C ? a.node2_w[c * 32 + k] : 0.f;
    }
    dvs_copy_image(l.Wa, (const dvs_bf16*)a.wimg + DvsLossImg::Wa, (int)(6 * DVS_IMG64));      // Wa, Wb triples
    dvs_stage_vector(l.bn1, a.node0_b, 32);
    for (int i = threadIdx.x; i < 16; i += blockDim.x) l.bn2[i] = i < C ? a.node2_b[i] : 0.f;
    dvs_stage_vector(l.be1, a.edge0_b, 64);
    dvs_stage_vector(l.w2, a.edge2_w, 64);
    if (threadIdx.x == 0) l.b2[0] = a.edge2_b[0];
    dvs_stage_vector(l.lg, a.ln.g, 64);
    dvs_stage_vector(l.lb, a.ln.b, 64);
}

__global__ __launch_bounds__(512) void k_loss_fwd(LossArgs a) {
    DVS_DYN_LDS(smem);
    const LossLds l = loss_lds(smem);
    loss_stage(l, a);
    __syncthreads();
    const Lane L = dvs_lane();
    const int N = a.dims.N, C = a.dims.C;
    float* scr = l.scr + L.wave * DVS_SCR;
    const float b2 = l.b2[0];
    for (int dag = blockIdx.x * L.nwaves + L.wave; dag < a.dims.B; dag += gridDim.x * L.nwaves) {
        f4 h[4], dummy[4];
        float rstd;
        dvs_load_x<false>(h, dummy, rstd, a.xin, a.ln, l.lg, l.lb, dag, N, L);
        const DvsRecord* rec = a.rec + dag;
        // ---- node head -------------------------------------------------------------------------------
        f4 t1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) t1[t] = dvs_vecT(l.bn1, t, L);
        dvs_mat_T<2, 4>(t1, h, l.Wn1, DVS_LD, 0, L);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) t1[t][kk] = fmaxf(t1[t][kk], 0.f);
        f4 lgt[1] = {*(const f4*)(l.bn2 + 4 * L.g)};
        dvs_mat_T<1, 2>(lgt, t1, l.Wn2, LOSS_LDN2, 0, L);
        float mx = -3.0e38f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mx = (4 * L.g + reg < C) ? fmaxf(mx, lgt[0][reg]) : mx;
        mx = dvs_max_g(mx);
        float se = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) se += (4 * L.g + reg < C) ? __expf(lgt[0][reg] - mx) : 0.f;
        se = dvs_sum_g(se);
        const float lse = mx + __logf(se);
        const int target = rec->label[(L.r + 1) & 15];
        float nll = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            nll -= (4 * L.g + reg == target && L.r < N - 1) ? (lgt[0][reg] - lse) : 0.f;
        // ---- edge head -------------------------------------------------------------------------------
        f4 U[4], V[4], w2v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            U[t] = f4_zero();
            V[t] = dvs_vecT(l.be1, t, L);
            w2v[t] = dvs_vecT(l.w2, t, L);
        }
        {   // fp32-accurate bf16x6 products; k_loss_bwd recomputes U, V with the same sequence (their sum's sign is a ReLU mask)
            const Split3T hs = dvs_split3_T(h);
            dvs_matb3<4>(U, hs, l.Wa, 64, 0, L);
            dvs_matb3<4>(V, hs, l.Wb, 64, 0, L);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) *(f4*)(scr + L.r * DVS_LD + 16 * t + 4 * L.g) = V[t];
        dvs_wave_sync();
        const unsigned par = rec->parents[(L.r + 1) & 15];
        float enll = 0.f;
#pragma unroll 2
        for (int j = 0; j < N - 2; ++j) {
            float e = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 vj = *(const f4*)(scr + j * DVS_LD + 16 * t + 4 * L.g);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) e += w2v[t][kk] * fmaxf(U[t][kk] + vj[kk], 0.f);
            }
            const float logit = dvs_sum_g(e) + b2;
            const bool pv = (L.r > j) && (L.r <= N - 2);
            const float truth = (float)((par >> (j + 1)) & 1u);
            const float bce = fmaxf(logit, 0.f) - logit * truth + log1pf(__expf(-fabsf(logit)));
            enll += pv ? bce : 0.f;
        }
        dvs_wave_sync();
        nll += (L.g == 0) ? enll : 0.f;
        nll = dvs_sum_wave(nll);
        if (L.lane == 0) a.dag_loss[(size_t)dag * 2] = nll;
    }
}

void dvs_launch_loss_fwd(const LossArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = dvs_loss_lds_floats(8, 1) * 4;
    DVS_SET_LDS(k_loss_fwd, lds);
    DVS_LAUNCH(k_loss_fwd, dim3(grid), dim3(512), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Deterministic reduction of the per-DAG losses: recon = sum NLL, kld = sum KL, total = recon + beta * kld
// (pace.py:2030-2035).  losses[3] = 1 if anything is non-finite (replaces the per-layer isnan host sync,
// pace.py:97-98, by one device-side flag per step); losses[4] = 1 if the records' validation word is set.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs a) {
    __shared__ float s0[256], s1[256];
    float r = 0.f, k = 0.f;
    for (int i = threadIdx.x; i < a.B; i += 256) {
        r += a.dag_loss[(size_t)i * 2];
        k += a.dag_loss[(size_t)i * 2 + 1];
    }
    s0[threadIdx.x] = r;
    s1[threadIdx.x] = k;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s0[threadIdx.x] += s0[threadIdx.x + s];
            s1[threadIdx.x] += s1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float recon = s0[0], kld = s1[0];
        const float total = recon + a.beta * kld;
        a.losses[0] = total;
        a.losses[1] = recon;
        a.losses[2] = kld;
        a.losses[3] = (total - total == 0.f) ? 0.f : 1.f;
        a.losses[4] = (a.status && *a.status != 0) ? 1.f : 0.f;      // invalid-features flag (travels with the scalars)
    }
}

void dvs_launch_finalize(const FinalizeArgs& a, dvs_stream_t st) {
    DVS_LAUNCH(k_finalize, dim3(1), dim3(256), 0, st, a);
}
