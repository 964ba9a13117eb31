// Latent block forward as a device function over one 8-wave workgroup: the standalone kernel (k_heads.hip: wide path, encode,
// per-phase profiles) and the last phase of the chained encoder forward (k_fwd_stack, k_forward.hip) share it.
#pragma once
#include "dvs_kernels.h"
#include "dvs_wimg.h"

constexpr int LAT_WAVES = 8;     // waves per 16-DAG group
constexpr int LAT_UB = 8;        // chunks per load batch
#ifndef DVS_LAT_STAMP
#define DVS_LAT_STAMP(id) ((void)0)
#endif
__device__ __forceinline__ float dvs_normal(uint32_t key, uint32_t e) {
    const uint32_t h1 = dvs_draw(key, 2 * e), h2 = dvs_draw(key, 2 * e + 1);
    const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * cosf(6.283185307179586f * u2);
}

// One 8-wave workgroup per group of 16 DAGs — DAGs base0 .. base0 + 7 and base1 .. base1 + 7: the standalone kernel takes 16
// consecutive ones, the chained forward (k_fwd_stack) the two 8-DAG runs its workgroup owns —: the contraction (fc1/fc2) and the
// output rows (fc3) are split over the waves by 16-float chunk index m; the fc1/fc2 partial sums meet in LDS (32 KB at `smem`)
// and are added in wave order.
// (measured on the standalone kernel: 4 waves 37 us, 8 waves 31 us, 16 waves 45 us)
__device__ __forceinline__ void dvs_latent_fwd_group(const LatentArgs& a, char* smem, int base0, int base1) {
    f4 (*part)[4][64] = (f4 (*)[4][64])smem;            // [LAT_WAVES][4][64]
    const Lane L = dvs_lane();
    const int B = a.dims.B, N = a.dims.N;
    const int dag = (L.r < 8 ? base0 : base1) + (L.r & 7);
    const bool dvalid = dag < B;
    DVS_LAT_STAMP(0);
    f4 acc[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
    // chunk m of the DAG's NT frag-order tiles: tile m >> 6, 16-float chunk mm = m & 63 of that tile
    const int NT = a.dims.NT, mch = 64 * NT / LAT_WAVES;
    const size_t dstride = (size_t)NT * DVS_TILE;
    const int m0 = mch * L.wave;
    const int K = 1024 * NT;
    // A operand: row 16 ot + r of the [fc1; fc2] image, contraction positions 16 m + 4 g .. + 3 (frag order, as the activations)
    const size_t LDA = DvsLatImg::LD(NT);
    const float* const wrow = a.limg + DvsLatImg::A(NT) + (size_t)L.r * LDA + 4 * L.g;
    // Batches of LAT_UB chunks with every load issued before the first MFMA, and no load under a condition: with
    // `dvalid ? load : 0` in the loop hipcc branches round the load, refuses to unroll ("loop not unrolled") and the walk pays
    // one full memory round trip per chunk — 8 in a row were 2/3 of this kernel's 24 us.  mch = 8 NT is a multiple of LAT_UB.
    const float* const xrow = a.xenc + (size_t)(dvalid ? dag : 0) * dstride + 4 * L.g;
    for (int mi = 0; mi < mch; mi += LAT_UB) {
        f4 xb[LAT_UB], wa[LAT_UB][4];
#pragma unroll
        for (int u = 0; u < LAT_UB; ++u) {
            const int m = m0 + mi + u;
            xb[u] = *(const f4*)(xrow + 16 * m);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) wa[u][ot] = *(const f4*)(wrow + (size_t)16 * ot * LDA + 16 * m);
        }
#pragma unroll
        for (int u = 0; u < LAT_UB; ++u) {
            if (!dvalid) xb[u] = f4_zero();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) acc[ot] = dvs_mfma(wa[u][ot][kk], xb[u][kk], acc[ot]);
        }
    }
    DVS_LAT_STAMP(1);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) part[L.wave][ot][L.lane] = acc[ot];
    __syncthreads();
    DVS_LAT_STAMP(2);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
        acc[ot] = *(const f4*)((ot < 2 ? a.fc1_b + 16 * ot : a.fc2_b + 16 * (ot - 2)) + 4 * L.g);
#pragma unroll
        for (int w = 0; w < LAT_WAVES; ++w) acc[ot] += part[w][ot][L.lane];
    }
    // acc[ot][reg] = out[o = 16(ot&1) + 4g + reg][dag r]; ot 0,1 = mu, ot 2,3 = logvar
    float kl = 0.f;
    f4 z[2];
    const uint32_t key = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, 100u, a.dims.dag_offset + dag);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f4 ev = f4_zero();
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float mu = acc[t][reg], lv = acc[t + 2][reg];
            const float elv = __expf(lv);
            kl += -0.5f * (1.0f + lv - mu * mu - elv);
            float zz = mu;
            if (a.dims.training) {
                const int o = 16 * t + 4 * L.g + reg;
                const float e = a.eps_in ? (dvalid ? a.eps_in[(size_t)dag * 32 + o] : 0.f)
                                         : dvs_normal(key, (uint32_t)o) * a.dims.eps_scale;
                ev[reg] = e;
                zz = mu + e * __expf(0.5f * lv);
            }
            z[t][reg] = zz;
        }
        if (dvalid && L.wave == 0) {
            const size_t o4 = (size_t)dag * 32 + 16 * t + 4 * L.g;
            *(f4*)(a.mu + o4) = acc[t];
            *(f4*)(a.logvar + o4) = acc[t + 2];
            *(f4*)(a.z + o4) = z[t];
            *(f4*)(a.epsv + o4) = ev;
        }
    }
    kl = dvs_sum_g(kl);
    if (L.wave == 0 && L.g == 0 && dvalid && a.dag_loss) a.dag_loss[(size_t)dag * 2 + 1] = kl;
    DVS_LAT_STAMP(3);
    if (!a.mem) return;                                  // uniform
    // mem^T chunk m (rows k' = 16 m ..): A = fc3 image rows (one 128-byte row per lane r, 16 bytes per (t, g)), bias in place
    const float* const w3 = a.limg + DvsLatImg::W3(NT) + (size_t)L.r * 32 + 4 * L.g;
    const float* const b3 = a.limg + DvsLatImg::B3(NT) + 4 * L.g;
    for (int mi = 0; mi < mch; mi += LAT_UB) {
        f4 o[LAT_UB], wa[LAT_UB][2];
#pragma unroll
        for (int u = 0; u < LAT_UB; ++u) {
            const int m = m0 + mi + u;
            o[u] = *(const f4*)(b3 + 16 * m);
#pragma unroll
            for (int t = 0; t < 2; ++t) wa[u][t] = *(const f4*)(w3 + (size_t)16 * m * 32 + 16 * t);
        }
#pragma unroll
        for (int u = 0; u < LAT_UB; ++u) {
            const int m = m0 + mi + u, mm = m & 63;
            const int tokD = 16 * (m >> 6) + 4 * (mm & 3) + L.g;        // token of this lane's 4 result rows
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) o[u] = dvs_mfma(wa[u][t][kk], z[t][kk], o[u]);
            if (dvalid) *(f4*)(a.mem + (size_t)dag * dstride + 16 * m + 4 * L.g) = tokD < N ? o[u] : f4_zero();
        }
    }
}
