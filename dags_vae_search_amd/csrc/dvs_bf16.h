// 64-deep products on the bf16 matrix pipe with fp32-class accuracy ("bf16x3").
//
// v_mfma_f32_16x16x4_f32 delivers 64 FLOP/clk/SIMD; v_mfma_f32_16x16x32_bf16 delivers 16x that (16 cycles for K = 32
// against 8 x 32 cycles).  Every operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <=
// 2^-18 |x|) and a product is three MFMAs, hi*hi + hi*lo + lo*hi, accumulated in fp32: relative error ~1e-5 per term,
// which leaves the ELBO within ~2e-6 of the fp32 path on every golden case (contract: 1e-4) — measured on the oracle
// with all linears replaced, and asserted by the parity tests.  5.3x fewer matrix-pipe cycles for the 64-wide linears.
//
// Operand layout: lane l = 16g + r supplies 8 consecutive k of row/column r.  Any k-permutation is allowed as long as A
// and B use the same one, so a K = 32 block is the tile pair (2p, 2p+1) of the T-layout: k = 8g + 4(t&1) + kk  <->
// feature 16t + 4g + kk.  An activation tile therefore splits register-locally (dvs_split_T), and weight images are
// bf16 [rows][DVS_LDB] with their columns in that order (dvs_kperm), hi and lo separately (built once per step by
// k_prepare_images, dvs_wimg.h, and copied into LDS): a lane's fragment is one 16-byte read.  The accumulator layout is that of the f32 16x16 MFMA, so results chain unchanged.
#pragma once
#include "dvs_device.h"

typedef __bf16 dvs_bf16;
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int DVS_LDB = 72;          // bf16 per image row: 144-byte stride -> the 16 rows of a fragment read hit 16 distinct 16-byte slots

__device__ __forceinline__ f4 dvs_mfma_bf(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// column of feature f inside the permuted image row
__device__ __forceinline__ int dvs_kperm(int f) {
    const int t = f >> 4, g = (f >> 2) & 3, kk = f & 3;
    return 32 * (t >> 1) + 8 * g + 4 * (t & 1) + kk;
}

__device__ __forceinline__ void dvs_split1(float v, dvs_bf16& hi, dvs_bf16& lo) {
    hi = (dvs_bf16)v;
    lo = (dvs_bf16)(v - (float)hi);
}
// two T-layout float4 (tiles 2p, 2p+1) -> the lane's 8-wide operand slice of K-block p
__device__ __forceinline__ void dvs_split8(const f4& a, const f4& b, bf8& hi, bf8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dvs_bf16 h, l;
        dvs_split1(a[i], h, l);
        hi[i] = h;
        lo[i] = l;
        dvs_split1(b[i], h, l);
        hi[4 + i] = h;
        lo[4 + i] = l;
    }
}
struct SplitT {                      // a 16-token x 64-feature tile as bf16x3 operand (8 VGPRs per half)
    bf8 hi[2], lo[2];
};
__device__ __forceinline__ SplitT dvs_split_T(const f4 (&x)[4]) {
    SplitT s;
    dvs_split8(x[0], x[1], s.hi[0], s.lo[0]);
    dvs_split8(x[2], x[3], s.hi[1], s.lo[1]);
    return s;
}

__device__ __forceinline__ bf8 dvs_wfrag(const dvs_bf16* img, int row, int p, const Lane& L) {
    return *(const bf8*)(img + row * DVS_LDB + 32 * p + 8 * L.g);
}

// y^T[OT] (T) += W[row0 + 16*OT rows][64] * x^T   (A = weight row fragments, B = activation)
template <int OT>
__device__ __forceinline__ void dvs_matb_T(f4 (&y)[OT], const SplitT& x, const dvs_bf16* Wh, const dvs_bf16* Wl, int row0,
                                           const Lane& L) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        bf8 ah[OT], al[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            ah[ot] = dvs_wfrag(Wh, row0 + 16 * ot + L.r, p, L);
            al[ot] = dvs_wfrag(Wl, row0 + 16 * ot + L.r, p, L);
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(ah[ot], x.hi[p], y[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(ah[ot], x.lo[p], y[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(al[ot], x.hi[p], y[ot]);
        DVS_SCHED_FENCE();
    }
}
// y[OT] (N) += x * W^T   (A = activation, B = weight row fragments): y[dt][reg] = Y[token 4g+reg][16dt + r]
template <int OT>
__device__ __forceinline__ void dvs_matb_N(f4 (&y)[OT], const SplitT& x, const dvs_bf16* Wh, const dvs_bf16* Wl, int row0,
                                           const Lane& L) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        bf8 bh[OT], bl[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            bh[ot] = dvs_wfrag(Wh, row0 + 16 * ot + L.r, p, L);
            bl[ot] = dvs_wfrag(Wl, row0 + 16 * ot + L.r, p, L);
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(x.hi[p], bh[ot], y[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(x.lo[p], bh[ot], y[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) y[ot] = dvs_mfma_bf(x.hi[p], bl[ot], y[ot]);
        DVS_SCHED_FENCE();
    }
}

// ---- bf16x6: three-way split, fp32-accurate -------------------------------------------------------------------------
// x = hi + mid + lo carries 24 significant bits (each part 8), i.e. the whole fp32 mantissa.  Six MFMAs per product
// (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the dropped terms are <= 2^-24 of the product) reproduce the fp32 result
// to ~1e-7 relative at 96 matrix-pipe cycles per K = 32 instead of 256: this is the variant the FORWARD may use, where
// the 1e-4 ELBO contract leaves no room for bf16x3's 1e-5.
struct Split3T {
    bf8 hi[2], mid[2], lo[2];
};
__device__ __forceinline__ void dvs_split3_1(float v, dvs_bf16& hi, dvs_bf16& mid, dvs_bf16& lo) {
    hi = (dvs_bf16)v;
    const float r1 = v - (float)hi;
    mid = (dvs_bf16)r1;
    lo = (dvs_bf16)(r1 - (float)mid);
}
__device__ __forceinline__ Split3T dvs_split3_T(const f4 (&x)[4]) {
    Split3T s;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dvs_bf16 h, m, l;
            dvs_split3_1(x[2 * p][i], h, m, l);
            s.hi[p][i] = h;
            s.mid[p][i] = m;
            s.lo[p][i] = l;
            dvs_split3_1(x[2 * p + 1][i], h, m, l);
            s.hi[p][4 + i] = h;
            s.mid[p][4 + i] = m;
            s.lo[p][4 + i] = l;
        }
    return s;
}
// y^T[OT] (T) += W[row0 + 16*OT rows][64] * x^T, W given as an image triple [hi | mid | lo][rows][DVS_LDB] (dvs_wimg.h)
template <int OT, bool N_LAYOUT = false>
__device__ __forceinline__ void dvs_matb3(f4 (&y)[OT], const Split3T& x, const dvs_bf16* img, int rows, int row0, const Lane& L) {
    const dvs_bf16* Wh = img;
    const dvs_bf16* Wm = img + rows * DVS_LDB;
    const dvs_bf16* Wl = Wm + rows * DVS_LDB;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        bf8 wh[OT], wm[OT], wl[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            wh[ot] = dvs_wfrag(Wh, row0 + 16 * ot + L.r, p, L);
            wm[ot] = dvs_wfrag(Wm, row0 + 16 * ot + L.r, p, L);
            wl[ot] = dvs_wfrag(Wl, row0 + 16 * ot + L.r, p, L);
        }
        // smallest terms first; N_LAYOUT swaps the operands (result [token][feature] instead of [feature][token])
#define DVS_M3(wa, xa)                                                                                     \
    _Pragma("unroll") for (int ot = 0; ot < OT; ++ot)                                                      \
        y[ot] = N_LAYOUT ? dvs_mfma_bf(xa[p], wa[ot], y[ot]) : dvs_mfma_bf(wa[ot], xa[p], y[ot]);
        DVS_M3(wl, x.hi)
        DVS_M3(wh, x.lo)
        DVS_M3(wm, x.mid)
        DVS_M3(wm, x.hi)
        DVS_M3(wh, x.mid)
        DVS_M3(wh, x.hi)
#undef DVS_M3
        DVS_SCHED_FENCE();
    }
}
