// clip_grad_norm_ + Adam over the flat parameter buffer (experiments/03_synthetic_12/main.py:115-116).
#include "dvs_backward.h"

// Deterministic global L2 norm in two fixed-order stages: 256 workgroups write partial sums of squares to
// scratch[2..258), one workgroup adds them.  scratch[0] = sum of squares, scratch[1] = clip coefficient
// min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_).
constexpr int SQ_PARTS = 256;
__global__ __launch_bounds__(256) void k_sqnorm_part(const float* g, int64_t n, float* scratch) {
    __shared__ float part[256];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)SQ_PARTS * 256) {
        const f4 v = *(const f4*)(g + 4 * i);
        s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0)
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) scratch[2 + blockIdx.x] = part[0];
}
// Every workgroup of k_adam re-derives the clip coefficient from the 256 partials (same tree, same order: bitwise the same
// value everywhere) instead of waiting for a one-workgroup launch in between; workgroup 0 publishes it.
// nparts: SQ_PARTS after k_sqnorm_part, dvs_sq_parts(n) when k_reduce_slabs left the partials (dvs_clip_adam_from_partials).
__global__ __launch_bounds__(256) void k_adam(int64_t n, float* p, float* g, float* m, float* v, float lr, float b1, float b2,
                                              float eps, float bc1, float bc2_sqrt, float max_norm, float* scratch,
                                              const float* guard, int nparts) {
    __shared__ float part[256];
    {
        float s = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) s += scratch[2 + i];      // fixed order
        part[threadIdx.x] = s;
    }
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    const float ss = part[0];
    float coef = 1.0f;
    if (max_norm > 0.f) {
        coef = max_norm / (sqrtf(ss) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scratch[0] = ss;
        scratch[1] = coef;
    }
    // non-finite loss or invalid features: the reference raises inside loss_direct, before backward / clip / step, so
    // neither the weights nor the moments move (uniform branch: every thread reads the same two words)
    if (guard && (guard[0] != 0.f || guard[1] != 0.f)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * coef;
    g[i] = gi;                                        // clip_grad_norm_ scales .grad in place
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

void dvs_launch_clip_adam(int64_t n, float* params, float* grads, float* m, float* v, float lr, float b1, float b2,
                          float eps, int64_t step, float max_norm, float* scratch, const float* guard, bool have_partials,
                          dvs_stream_t st) {
    if (!have_partials) DVS_LAUNCH(k_sqnorm_part, dim3(SQ_PARTS), dim3(256), 0, st, (const float*)grads, n, scratch);
    const float bc1 = 1.0f - powf(b1, (float)step);
    const float bc2 = 1.0f - powf(b2, (float)step);
    DVS_LAUNCH(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, params, grads, m, v, lr, b1, b2, eps,
                       bc1, sqrtf(bc2), max_norm, scratch, guard, have_partials ? dvs_sq_parts(n) : SQ_PARTS);
}

// ---------------------------------------------------------------------------------------------------------
// Per-step weight images (dvs_wimg.h): one workgroup per (job, 16-row slice).
// ---------------------------------------------------------------------------------------------------------
#include "dvs_wimg.h"
__device__ __forceinline__ float dvs_loss_head_value(const DvsLossHeadArgs& h, int i) {      // element i of the head block
    if (i < 32 * 68) {
        const int r = i / 68, c = i % 68;
        return c < 64 ? h.node0_w[r * 64 + c] : 0.f;
    }
    i -= 32 * 68;
    if (i < 16 * 36) {
        const int c = i / 36, k = i % 36;
        return (c < h.C && k < 32) ? h.node2_w[c * 32 + k] : 0.f;
    }
    i -= 16 * 36;
    if (i < 32) return h.node0_b[i];
    i -= 32;
    if (i < 16) return i < h.C ? h.node2_b[i] : 0.f;
    i -= 16;
    if (i < 64) return h.edge0_b[i];
    i -= 64;
    if (i < 64) return h.edge2_w[i];
    i -= 64;
    if (i < 16) return i == 0 ? h.edge2_b[0] : 0.f;
    i -= 16;
    if (i < 64) return h.ln_g[i];
    return h.ln_b[i - 64];
}
__device__ __forceinline__ float dvs_emb_block_value(const DvsLossHeadArgs& h, int i) {      // element i of the embedding block
    if (i < 32 * 68) {
        const int r = i / 68, c = i % 68;
        return (r < 2 * h.N && c < 64) ? h.W1[r * 64 + c] : 0.f;
    }
    i -= 32 * 68;
    if (i < 64 * 36) {
        const int f = i / 36, k = i % 36;
        return k < 32 ? h.W2[f * 32 + k] : 0.f;
    }
    i -= 64 * 36;
    if (i < 512) {
        const int f = i >> 4, c = i & 15;
        return c < h.C ? h.lab_w[f * h.C + c] : 0.f;
    }
    i -= 512;
    return i < 32 ? h.lab_b[i] : 0.f;
}
__global__ __launch_bounds__(256) void k_prepare_images(DvsImgJobs jobs, const float* params, dvs_bf16* wimg, DvsLatImgArgs lat,
                                                        DvsLossHeadArgs head) {
    const int nlat = 1024 * lat.NT / 64;                             // latent blocks come FIRST: their serial transposes overlap the rest
    const int first_job = nlat + (head.dst ? 2 : 0);                 // then the loss head block and the embedding block, then the jobs
    if (head.dst) {
        if ((int)blockIdx.x == nlat + 1) {
            float v[DvsEmbImg::FLOATS / 256];
#pragma unroll
            for (int u = 0; u < DvsEmbImg::FLOATS / 256; ++u) v[u] = dvs_emb_block_value(head, u * 256 + (int)threadIdx.x);
#pragma unroll
            for (int u = 0; u < DvsEmbImg::FLOATS / 256; ++u) head.dst_emb[u * 256 + threadIdx.x] = v[u];
            return;
        }
        if ((int)blockIdx.x == nlat) {
            float v[DvsLossImg::HEAD_FLOATS / 256];
#pragma unroll
            for (int u = 0; u < DvsLossImg::HEAD_FLOATS / 256; ++u) v[u] = dvs_loss_head_value(head, u * 256 + (int)threadIdx.x);
#pragma unroll
            for (int u = 0; u < DvsLossImg::HEAD_FLOATS / 256; ++u) head.dst[u * 256 + threadIdx.x] = v[u];
            return;
        }
    }
    if ((int)blockIdx.x < nlat) {
        // latent images (DvsLatImg): one workgroup per 64 consecutive frag positions k'; both orientations of every matrix are
        // written as contiguous rows through an LDS tile (4-byte scatters at 4 KB stride cost this launch 7 us)
        __shared__ float tile[64][65];
        const int K = 1024 * lat.NT, kp0 = (int)blockIdx.x * 64;
        const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
        const size_t ldw = (size_t)lat.N * 64;
        auto col_of = [&](int kp, bool& valid) {
            const int tile_i = kp >> 10, q = (kp & 1023) >> 2, kk = kp & 3, t = q >> 6, ln = q & 63;
            const int tok = 16 * tile_i + (ln & 15), feat = 16 * t + 4 * (ln >> 4) + kk;
            valid = tok < lat.N;
            return (size_t)(valid ? tok : 0) * 64 + feat;
        };
        bool valid;
        const size_t col = col_of(kp0 + l, valid);                 // this lane's position when lanes run over k'
        float va[16];                                                // all loads first: a load -> store loop pays a round trip each
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = w + 4 * i;
            va[i] = !valid ? 0.f : (o < 32 ? lat.fc1_w[o * ldw + col] : lat.fc2_w[(o - 32) * ldw + col]);
        }
        float v3[8];                                                 // lanes run over o (two k' per wave pass)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bool v2;
            const size_t c2 = col_of(kp0 + 2 * (w + 4 * i) + (l >> 5), v2);
            v3[i] = v2 ? lat.fc3_w[c2 * 32 + (l & 31)] : 0.f;
        }
        const float vb3 = valid ? lat.fc3_b[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = w + 4 * i;
            tile[o][l] = va[i];
            lat.img[DvsLatImg::A(lat.NT) + (size_t)o * DvsLatImg::LD(lat.NT) + kp0 + l] = va[i];
        }
        __syncthreads();
        for (int j = w; j < 64; j += 4) lat.img[DvsLatImg::AT(lat.NT) + (size_t)(kp0 + j) * 64 + l] = tile[l][j];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int jj = 2 * (w + 4 * i) + (l >> 5), o = l & 31;
            tile[o][jj] = v3[i];
            lat.img[DvsLatImg::W3(lat.NT) + (size_t)(kp0 + jj) * 32 + o] = v3[i];
        }
        __syncthreads();
        for (int o = w; o < 32; o += 4) lat.img[DvsLatImg::W3T(lat.NT) + (size_t)o * DvsLatImg::LD(lat.NT) + kp0 + l] = tile[o][l];
        if (w == 0) lat.img[DvsLatImg::B3(lat.NT) + kp0 + l] = vb3;
        return;
    }
    const int jb = ((int)blockIdx.x - first_job) / 12, slice = ((int)blockIdx.x - first_job) % 12;         // up to 192 rows = 12 slices of 16
    const DvsImgJob J = jobs.job[jb];
    const bool transposed = J.flags & 1, rperm = J.flags & 2, cperm = J.flags & 4;
    const float* src = params + J.src;
    dvs_bf16* dst = wimg + J.dst;
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) {
        const int row = 16 * slice + (i >> 6), col = i & 63;
        if (row >= J.rows) break;
        const float v = src[(size_t)(rperm ? dvs_pi(row) : row) * ((J.flags & 8) ? 128 : 64) + (cperm ? dvs_pi(col) : col)];
        if (!transposed) {                       // x6: [3][rows][LDB], img[row][kperm(col)]
            dvs_bf16 h, m, l;
            dvs_split3_1(v, h, m, l);
            const size_t o = (size_t)row * DVS_LDB + dvs_kperm(col), part = (size_t)J.rows * DVS_LDB;
            dst[o] = h;
            dst[part + o] = m;
            if (!(J.flags & 16)) dst[2 * part + o] = l;
        } else {                                 // x3 transposed: [2][64][LDB], img[col][kperm(row)]
            dvs_bf16 h, l;
            dvs_split1(v, h, l);
            const size_t o = (size_t)col * DVS_LDB + dvs_kperm(row);
            dst[o] = h;
            dst[DVS_IMG64 + o] = l;
        }
    }
}
void dvs_launch_prepare_images(const DvsImgJobs& jobs, const float* params, dvs_bf16* wimg, const DvsLatImgArgs& lat,
                               const DvsLossHeadArgs& head, dvs_stream_t st) {
    DVS_LAUNCH(k_prepare_images, dim3(jobs.count * 12 + 1024 * lat.NT / 64 + (head.dst ? 2 : 0)), dim3(256), 0, st, jobs, params,
               wimg, lat, head);
}
