// Latent block backward, part 1, as a device function over one workgroup of NW waves: the standalone kernel
// (k_backward_heads.hip: wide path, per-phase profiles; 16 waves) and the chained encoder backward (k_bwd_stack, k_backward.hip:
// 8 waves, ahead of its first phase) share it.  DAGs base0 .. base0 + 7 and base1 .. base1 + 7 form the MFMA group of 16.
#pragma once
#include "dvs_backward.h"
#include "dvs_wimg.h"

template <int NW>
__device__ __forceinline__ void dvs_latent_bwd_group(const LatentBwdArgs& a, f4 (*part)[2][64], int base0, int base1) {
    const Lane L = dvs_lane();
    const int B = a.dims.B, N = a.dims.N;
    const float gkl = a.gcoef[1];
    const int dag = (L.r < 8 ? base0 : base1) + (L.r & 7);
    const bool dvalid = dag < B;
    const int NT = a.dims.NT, mch = 64 * NT / NW;   // chunk m: tile m >> 6, chunk m & 63 of it (k_latent_fwd)
    const size_t dstride = (size_t)NT * DVS_TILE;
    const int m0 = mch * L.wave;
    f4 dz[2] = {f4_zero(), f4_zero()};
    const int K = 1024 * NT;
    // d z^T = fc3^T d mem^T: A = row 16 t + r of the transposed fc3 image, contraction positions 16 m + 4 g .. + 3
    const size_t LDA = DvsLatImg::LD(NT);
    const float* const w3t = a.limg + DvsLatImg::W3T(NT) + (size_t)L.r * LDA + 4 * L.g;
    // load batches of UB chunks, nothing under a condition (dvs_latent.h)
    constexpr int UB = 8;
    const float* const grow = a.gmem + (size_t)(dvalid ? dag : 0) * dstride + 4 * L.g;
    for (int mi = 0; mi < mch; mi += UB) {
        f4 gb[UB], wa[UB][2];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int m = m0 + mi + u;
            gb[u] = *(const f4*)(grow + 16 * m);
#pragma unroll
            for (int t = 0; t < 2; ++t) wa[u][t] = *(const f4*)(w3t + (size_t)16 * t * LDA + 16 * m);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (!dvalid) gb[u] = f4_zero();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < 2; ++t) dz[t] = dvs_mfma(wa[u][t][kk], gb[u][kk], dz[t]);
        }
    }
    part[L.wave][0][L.lane] = dz[0];
    part[L.wave][1][L.lane] = dz[1];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        dz[t] = part[0][t][L.lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) dz[t] += part[w][t][L.lane];
    }
    // dz[t][reg] = d z[o = 16t + 4g + reg][dag r]
    f4 dout[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const size_t o4 = (size_t)(dvalid ? dag : 0) * 32 + 16 * t + 4 * L.g;
        const f4 mu = *(const f4*)(a.mu + o4), lv = *(const f4*)(a.logvar + o4), ev = *(const f4*)(a.epsv + o4);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float dzz = dvalid ? dz[t][reg] : 0.f;
            float dmu = dzz + gkl * mu[reg];
            float dlv = gkl * 0.5f * (__expf(lv[reg]) - 1.0f);
            if (a.dims.training) dlv += dzz * ev[reg] * 0.5f * __expf(0.5f * lv[reg]);
            dout[t][reg] = dvalid ? dmu : 0.f;
            dout[t + 2][reg] = dvalid ? dlv : 0.f;
        }
        if (dvalid && L.wave == 0) {
            *(f4*)(a.gz + (size_t)dag * 64 + 16 * t + 4 * L.g) = dout[t];
            *(f4*)(a.gz + (size_t)dag * 64 + 32 + 16 * t + 4 * L.g) = dout[t + 2];
        }
    }
    // d enc_out^T[k'][dag] = sum_o Wfc[o][k'] dout^T[o][dag]: A = rows 16 m + r of the transposed [fc1; fc2] image (256 bytes each)
    const float* const wat = a.limg + DvsLatImg::AT(NT) + (size_t)L.r * 64 + 4 * L.g;
    for (int mi = 0; mi < mch; mi += 4) {
        f4 wa[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) wa[u][t] = *(const f4*)(wat + (size_t)16 * (m0 + mi + u) * 64 + 16 * t);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + mi + u, mm = m & 63;
            const int tokD = 16 * (m >> 6) + 4 * (mm & 3) + L.g;
            f4 o0 = f4_zero(), o1 = f4_zero();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                o0 = dvs_mfma(wa[u][0][kk], dout[0][kk], o0);
                o1 = dvs_mfma(wa[u][1][kk], dout[1][kk], o1);
                o0 = dvs_mfma(wa[u][2][kk], dout[2][kk], o0);
                o1 = dvs_mfma(wa[u][3][kk], dout[3][kk], o1);
            }
            if (dvalid) *(f4*)(a.genc + (size_t)dag * dstride + 16 * m + 4 * L.g) = tokD < N ? o0 + o1 : f4_zero();
        }
    }
}
