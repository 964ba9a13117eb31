// Native BIC scorer for discrete Bayesian networks: replaces the per-graph `Rscript bnlearn` subprocess
// of BNLearnWrapper.score (src/problem/bn/bnlearn.py:27-61, bnlearn_scripts/bnlearn_score.R:25-39) by a batched
// contingency-count kernel.  bnlearn's discrete BIC is decomposable:
//   BIC(G) = sum_v [ sum_{j,k} N_vjk log(N_vjk / N_vj) - (log S / 2)(r_v - 1) q_v ]
// One workgroup per (DAG, variable): the S samples (4-bit level codes, 16 variables per 64-bit word, sample-major:
// one coalesced 8-byte load per lane and word; the whole asia / sachs data set is 40 KB and lives in L2) are binned
// into an LDS histogram over (parent configuration, state) with integer atomics (or, for parent sets whose table does
// not fit LDS, sorted as 64-bit keys), then the log-likelihood terms are summed in fp64 in a fixed order (integer
// counts are exact, so the score does not depend on the atomics' order).
// Integer/byte work bound by LDS atomics and L2 reads — no MFMA.
#include "dvs_kernels.h"

constexpr int BIC_MAX_BINS = 36864;          // q_v * r_v histogram bins that fit LDS (144 KB of u32 counters)

struct BicArgs {
    int B, n, S, words;
    const uint64_t* data;        // [S][words]
    const uint8_t* card;         // [n] levels of each variable (2..16)
    const uint64_t* parents;     // [B][n]: bit u of parents[b][v] <=> edge u -> v (dataset variable indices)
    double* local;               // [B][n] scratch: local scores
    double* out;                 // [B]
    int* status;
};

__device__ __forceinline__ int bic_level(const uint64_t* row, int var) { return (int)((row[var >> 4] >> (4 * (var & 15))) & 15ull); }

constexpr int BIC_MAX_SORT = 16384;          // samples the sort path can hold in LDS (64-bit keys)

__device__ __forceinline__ int bic_bits(int card) {   // bits needed for a level code 0 .. card-1
    int b = 0;
    while ((1 << b) < card) ++b;
    return b;
}
// first index in sorted keys[0..n) with keys[i] >= x
__device__ __forceinline__ int bic_lower_bound(const uint64_t* keys, int n, uint64_t x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Two counting strategies, chosen per (DAG, variable):
//   histogram: q_v * r_v <= BIC_MAX_BINS cells fit LDS -> integer atomics into the dense table (the common case);
//   sort:      larger parent sets (sachs: >= 9 ternary parents) are sparse — at most S cells are occupied — so the
//              bit-packed 64-bit (configuration, state) keys of the S samples are bitonic-sorted in LDS and the counts
//              are run lengths found by binary search.
__global__ __launch_bounds__(256) void k_bic_local(BicArgs a) {
    DVS_DYN_LDS(smem);
    __shared__ double red[256];
    __shared__ int par_id[48], par_stride[48], par_shift[48];
    __shared__ int s_np, s_mode, s_rbits;
    __shared__ double s_q;
    const int v = blockIdx.x % a.n, dag = blockIdx.x / a.n;
    const int r = a.card[v];
    if (threadIdx.x == 0) {
        uint64_t pm = a.parents[(size_t)dag * a.n + v] & ~(1ull << v);
        double q = 1.0;
        long long qi = 1;
        int np = 0, bits = bic_bits(r), mode = 0;          // mode 0 histogram, 1 sort, -1 unsupported
        for (; pm; pm &= pm - 1) {
            const int p = __builtin_ctzll(pm);
            if (p >= a.n) { mode = -1; break; }
            par_id[np] = p;
            par_stride[np] = (int)qi;                      // histogram: mixed radix, lowest variable id fastest
            par_shift[np] = bits;                          // sort: bit-packed above the state's bits
            bits += bic_bits(a.card[p]);
            q *= a.card[p];
            if (mode == 0) {
                qi *= a.card[p];
                if (qi * r > BIC_MAX_BINS) mode = 1;
            }
            ++np;
        }
        if (mode == 1 && (bits > 63 || a.S > BIC_MAX_SORT)) mode = -1;
        s_np = np;
        s_q = q;
        s_mode = mode;
        s_rbits = bic_bits(r);
    }
    __syncthreads();
    const int np = s_np, mode = s_mode;
    if (mode < 0) {
        if (threadIdx.x == 0) {
            atomicOr(a.status, 16);
            a.local[(size_t)dag * a.n + v] = __longlong_as_double(0x7ff8000000000000LL);
        }
        return;
    }
    double acc = 0.0;
    if (mode == 0) {
        unsigned* hist = (unsigned*)smem;
        const int q = (int)s_q, bins = q * r;
        for (int i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
        for (int s = threadIdx.x; s < a.S; s += blockDim.x) {
            const uint64_t* row = a.data + (size_t)s * a.words;
            int key = 0;
            for (int i = 0; i < np; ++i) key += bic_level(row, par_id[i]) * par_stride[i];
            atomicAdd(&hist[key * r + bic_level(row, v)], 1u);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < q; j += blockDim.x) {
            unsigned nj = 0;
            for (int k = 0; k < r; ++k) nj += hist[j * r + k];
            if (nj == 0) continue;
            const double dn = (double)nj;
            for (int k = 0; k < r; ++k) {
                const unsigned c = hist[j * r + k];
                if (c) acc += (double)c * log((double)c / dn);
            }
        }
    } else {
        uint64_t* keys = (uint64_t*)smem;
        const int S = a.S, rbits = s_rbits;
        int spad = 1;
        while (spad < S) spad <<= 1;
        for (int s = threadIdx.x; s < spad; s += blockDim.x) {
            uint64_t key = ~0ull;
            if (s < S) {
                const uint64_t* row = a.data + (size_t)s * a.words;
                key = (uint64_t)bic_level(row, v);
                for (int i = 0; i < np; ++i) key |= (uint64_t)bic_level(row, par_id[i]) << par_shift[i];
            }
            keys[s] = key;
        }
        __syncthreads();
        for (int k = 2; k <= spad; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < spad; i += blockDim.x) {
                    const int p = i ^ j;
                    if (p > i) {
                        const uint64_t x = keys[i], y = keys[p];
                        if ((x > y) == ((i & k) == 0)) {
                            keys[i] = y;
                            keys[p] = x;
                        }
                    }
                }
                __syncthreads();
            }
        for (int i = threadIdx.x; i < S; i += blockDim.x) {
            const uint64_t key = keys[i];
            if (i > 0 && keys[i - 1] == key) continue;                 // not the first sample of its (j, k) cell
            const int c = bic_lower_bound(keys, S, key + 1) - i;
            const uint64_t j0 = (key >> rbits) << rbits;
            const int nj = bic_lower_bound(keys, S, j0 + (1ull << rbits)) - bic_lower_bound(keys, S, j0);
            acc += (double)c * log((double)c / (double)nj);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        a.local[(size_t)dag * a.n + v] = red[0] - 0.5 * log((double)a.S) * (double)(r - 1) * s_q;
}

__global__ __launch_bounds__(256) void k_bic_sum(BicArgs a) {
    const int dag = blockIdx.x * blockDim.x + threadIdx.x;
    if (dag >= a.B) return;
    double s = 0.0;
    for (int v = 0; v < a.n; ++v) s += a.local[(size_t)dag * a.n + v];
    a.out[dag] = s;
}

void dvs_launch_bic(const BicArgs& a, dvs_stream_t st) {
    const size_t lds = (size_t)BIC_MAX_BINS * sizeof(unsigned);
    DVS_SET_LDS(k_bic_local, lds);
    DVS_LAUNCH(k_bic_local, dim3((unsigned)a.B * a.n), dim3(256), lds, st, a);
    DVS_LAUNCH(k_bic_sum, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
}

extern "C" int dvs_bic_scores_impl(int B, int n, int S, const uint64_t* data, const uint8_t* card, const uint64_t* parents,
                                   double* local, double* out, int* status, void* stream) {
    BicArgs a;
    a.B = B;
    a.n = n;
    a.S = S;
    a.words = (n + 15) / 16;
    a.data = data;
    a.card = card;
    a.parents = parents;
    a.local = local;
    a.out = out;
    a.status = status;
    dvs_launch_bic(a, (dvs_stream_t)stream);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// GP predictor, predictive mean: mean(x*) = c + sum_m o exp(-|x* - z_m|^2 / (2 l^2)) alpha_m for the
// reference's SGPR model (src/predictors/gp.py:13-32: ConstantMean + InducingPointKernel(ScaleKernel(RBF))), with alpha
// solved once at fit time.  One wave per query: its 32-dim latent in registers (lane = dimension pair), the M inducing
// points streamed from global/L2 ([M][32] fp32 = 64 KB), fp64 accumulation: SGPR weights alternate in sign and are
// orders of magnitude larger than the result.  B x M x 32 MACs — tiny; the kernel exists so that the encode -> predict
// -> decode loop of latent-space search never leaves the device.
// ---------------------------------------------------------------------------------------------------------
struct GpArgs {
    int B, M, D;
    const float* x;              // [B][D] queries
    const float* z;              // [M][D] inducing points
    const double* alpha;         // [M]
    double outputscale, inv2l2, constant;
    double* out;                 // [B]
};
__global__ __launch_bounds__(256) void k_gp_predict(GpArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= a.B) return;
    // lane m-stride: each lane owns inducing points lane, lane + 64, ...; the query is read once per lane (L1 broadcast)
    double acc = 0.0;
    for (int m = lane; m < a.M; m += 64) {
        const float* zm = a.z + (size_t)m * a.D;
        const float* xq = a.x + (size_t)q * a.D;
        float d2 = 0.f;
        for (int k = 0; k < a.D; ++k) {
            const float d = xq[k] - zm[k];
            d2 = fmaf(d, d, d2);
        }
        acc += a.alpha[m] * exp(-(double)d2 * a.inv2l2);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        // 64-bit shuffle as two 32-bit halves
        long long bits = __double_as_longlong(acc);
        int lo = (int)(bits & 0xffffffffLL), hi = (int)(bits >> 32);
        lo = __shfl_xor(lo, s);
        hi = __shfl_xor(hi, s);
        acc += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    if (lane == 0) a.out[q] = a.constant + a.outputscale * acc;
}

extern "C" int dvs_gp_predict_impl(int B, int M, int D, const float* x, const float* z, const double* alpha,
                                   double outputscale, double lengthscale, double constant, double* out, void* stream) {
    GpArgs a;
    a.B = B;
    a.M = M;
    a.D = D;
    a.x = x;
    a.z = z;
    a.alpha = alpha;
    a.outputscale = outputscale;
    a.inv2l2 = 0.5 / (lengthscale * lengthscale);
    a.constant = constant;
    a.out = out;
    DVS_LAUNCH(k_gp_predict, dim3((B + 3) / 4), dim3(256), 0, (dvs_stream_t)stream, a);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// GP predictor, hyper-parameter training (reference loop src/predictors/gp.py:55-81 =
// experiments/01_bn_asia/main.py:315-393: Adam on -ExactMarginalLogLikelihood of the SGPR model).  The two kernel-specific
// pieces of one training iteration; the dense M x M / M x n factorisations in between are library calls of the host side
// (predictor.py), the parameter update is dvs_clip_adam.
//   k_gp_kernel      K[a][b] = o exp(-|xa_a - xb_b|^2 / (2 l^2)), float64 (K_uu needs all of it: its Cholesky factor is
//                    ill-conditioned at M = 500), float32 points.
//   k_gp_kernel_bwd  given G = dF/dK: row a's pull-backs  dxa_a = sum_b G'_ab K_ab (xb_b - xa_a) / l^2  (G' = G + G^T when
//                    xa and xb are the same point set, K_uu) and the per-row partial sums of dF/dl = sum G K d^2 / l^3 and
//                    dF/do = sum G K / o.  One wave per row a, lanes stride over b, fixed-order reduction: bitwise
//                    reproducible.  K is recomputed from the points (a 32-dim distance) instead of read.
// ---------------------------------------------------------------------------------------------------------
struct GpKernArgs {
    int na, nb, D, symmetric;
    const float* xa;
    const float* xb;
    double outputscale, inv2l2, inv_l2, inv_l3, inv_o;
    double* K;                   // forward: [na][nb]
    const double* G;             // backward: [na][nb]
    double* dxa;                 // backward: [na][D]
    double* rows;                // backward: [na][2]: d/dl, d/do partials
};
constexpr int DVS_GP_MAXD = 32;
__global__ __launch_bounds__(256) void k_gp_kernel(GpKernArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.na * a.nb) return;
    const int ra = (int)(i / a.nb), rb = (int)(i - (size_t)ra * a.nb);
    const float* pa = a.xa + (size_t)ra * a.D;
    const float* pb = a.xb + (size_t)rb * a.D;
    double d2 = 0.0;
    for (int k = 0; k < a.D; ++k) {
        const double d = (double)pa[k] - (double)pb[k];
        d2 = fma(d, d, d2);
    }
    a.K[i] = a.outputscale * exp(-d2 * a.inv2l2);
}
__device__ __forceinline__ double dvs_wave_sum_f64(double v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const long long bits = __double_as_longlong(v);
        int lo = (int)(bits & 0xffffffffLL), hi = (int)(bits >> 32);
        lo = __shfl_xor(lo, s);
        hi = __shfl_xor(hi, s);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}
__global__ __launch_bounds__(256) void k_gp_kernel_bwd(GpKernArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ra = blockIdx.x * 4 + wave;
    if (ra >= a.na) return;
    const float* pa = a.xa + (size_t)ra * a.D;
    double xa[DVS_GP_MAXD], acc[DVS_GP_MAXD];
#pragma unroll
    for (int k = 0; k < DVS_GP_MAXD; ++k) {
        xa[k] = k < a.D ? (double)pa[k] : 0.0;
        acc[k] = 0.0;
    }
    double sl = 0.0, so = 0.0;
    for (int rb = lane; rb < a.nb; rb += 64) {
        const float* pb = a.xb + (size_t)rb * a.D;
        double diff[DVS_GP_MAXD], d2 = 0.0;
#pragma unroll
        for (int k = 0; k < DVS_GP_MAXD; ++k) {
            diff[k] = k < a.D ? (double)pb[k] - xa[k] : 0.0;
            d2 = fma(diff[k], diff[k], d2);
        }
        const double Kab = a.outputscale * exp(-d2 * a.inv2l2);
        const double g = a.G[(size_t)ra * a.nb + rb];
        const double w = g * Kab;
        sl += w * d2;
        so += w;
        const double wr = a.symmetric ? (g + a.G[(size_t)rb * a.nb + ra]) * Kab : w;
#pragma unroll
        for (int k = 0; k < DVS_GP_MAXD; ++k) acc[k] = fma(wr, diff[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < DVS_GP_MAXD; ++k) {
        const double s = dvs_wave_sum_f64(acc[k]);
        if (lane == 0 && k < a.D) a.dxa[(size_t)ra * a.D + k] = s * a.inv_l2;
    }
    sl = dvs_wave_sum_f64(sl);
    so = dvs_wave_sum_f64(so);
    if (lane == 0) {
        a.rows[2 * (size_t)ra] = sl * a.inv_l3;
        a.rows[2 * (size_t)ra + 1] = so * a.inv_o;
    }
}

static GpKernArgs gp_kern_args(int na, int nb, int D, const float* xa, const float* xb, double outputscale, double lengthscale) {
    GpKernArgs a = {};
    a.na = na;
    a.nb = nb;
    a.D = D;
    a.xa = xa;
    a.xb = xb;
    a.outputscale = outputscale;
    a.inv2l2 = 0.5 / (lengthscale * lengthscale);
    a.inv_l2 = 1.0 / (lengthscale * lengthscale);
    a.inv_l3 = 1.0 / (lengthscale * lengthscale * lengthscale);
    a.inv_o = 1.0 / outputscale;
    return a;
}
extern "C" int dvs_gp_kernel_impl(int na, int nb, int D, const float* xa, const float* xb, double outputscale, double lengthscale,
                                  double* K, void* stream) {
    GpKernArgs a = gp_kern_args(na, nb, D, xa, xb, outputscale, lengthscale);
    a.K = K;
    const size_t n = (size_t)na * nb;
    DVS_LAUNCH(k_gp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (dvs_stream_t)stream, a);
    return 0;
}
extern "C" int dvs_gp_kernel_backward_impl(int na, int nb, int D, int symmetric, const float* xa, const float* xb,
                                           double outputscale, double lengthscale, const double* G, double* dxa, double* rows,
                                           void* stream) {
    GpKernArgs a = gp_kern_args(na, nb, D, xa, xb, outputscale, lengthscale);
    a.symmetric = symmetric;
    a.G = G;
    a.dxa = dxa;
    a.rows = rows;
    DVS_LAUNCH(k_gp_kernel_bwd, dim3((unsigned)((na + 3) / 4)), dim3(256), 0, (dvs_stream_t)stream, a);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Row codec -> BIC parent masks on the device (the relabelling of BNLearnWrapper.score, src/problem/bn/bnlearn.py:34-45:
// graph vertex v stands for data-set variable labels[v]): parents[b][labels[v]] = OR over predecessors u of (1 << labels[u]).
// One thread per (DAG, vertex).  status bit 5: the labels of a DAG are not a permutation of 0..n-1 (the reference asserts,
// bnlearn.py:35) — that DAG's masks are zeroed.
// ---------------------------------------------------------------------------------------------------------
struct BicMaskArgs {
    int B, n, wide;
    const uint8_t* labels;       // [B][n]
    const void* preds;           // [B][n] u16 (wide == 0) or u64
    uint64_t* parents;           // [B][n]
    int* status;
};
__global__ __launch_bounds__(256) void k_bic_parent_masks(BicMaskArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.n) return;
    const int b = i / a.n, v = i - b * a.n;
    const uint8_t* lab = a.labels + (size_t)b * a.n;
    uint64_t seen = 0;
    bool ok = true;
    for (int u = 0; u < a.n; ++u) {
        ok = ok && lab[u] < a.n;
        seen |= 1ull << (lab[u] & 63);
    }
    ok = ok && seen == (a.n == 64 ? ~0ull : (1ull << a.n) - 1ull);
    const uint64_t pr = a.wide ? ((const uint64_t*)a.preds)[i] : (uint64_t)((const uint16_t*)a.preds)[i];
    uint64_t m = 0;
    for (int u = 0; u < v; ++u)
        if ((pr >> u) & 1ull) m |= 1ull << (lab[u] & 63);
    if (!ok) {
        atomicOr(a.status, 32);
        a.parents[(size_t)b * a.n + v] = 0;
        return;
    }
    a.parents[(size_t)b * a.n + lab[v]] = m;
}
extern "C" int dvs_bic_parent_masks_impl(int B, int n, int wide, const uint8_t* labels, const void* preds, uint64_t* parents,
                                         int* status, void* stream) {
    BicMaskArgs a;
    a.B = B;
    a.n = n;
    a.wide = wide;
    a.labels = labels;
    a.preds = preds;
    a.parents = parents;
    a.status = status;
    DVS_LAUNCH(k_bic_parent_masks, dim3((unsigned)((B * n + 255) / 256)), dim3(256), 0, (dvs_stream_t)stream, a);
    return 0;
}
