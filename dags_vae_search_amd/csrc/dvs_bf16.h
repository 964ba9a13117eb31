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

// ---- parked bf16 tiles and transposing LDS reads: cooperative weight gradients on the bf16 pipe -------------------------
// dW = dY^T X contracts over TOKENS, so an MFMA operand is 8 consecutive tokens of one feature per lane — the transpose of
// how a tile is parked ([token][feature]).  gfx950's ds_read_b64_tr_b16 delivers exactly that: per 16-lane group it reads a
// block of 4 rows x 16 columns of 16-bit elements and hands lane i column i of the 4 rows (cdna_hip_programming.md T10).
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
constexpr int DVS_PLD = 68;                  // bf16 per parked row (136 B): a [hi | lo] pair is exactly one fp32 scratch tile (DVS_SCR)
constexpr int DVS_PKB = 16 * DVS_PLD;        // bf16 elements of one parked part [16 tokens][DVS_PLD]

// park a T-layout tile as hi / lo bf16 images (lo directly behind hi), row-major [token][feature]: 8-byte stores
__device__ __forceinline__ void dvs_park_bf(dvs_bf16* img, const f4 (&x)[4], const Lane& L) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        bf4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dvs_bf16 hh, ll;
            dvs_split1(x[t][i], hh, ll);
            h[i] = hh;
            l[i] = ll;
        }
        *(bf4*)(img + L.r * DVS_PLD + 16 * t + 4 * L.g) = h;
        *(bf4*)(img + DVS_PKB + L.r * DVS_PLD + 16 * t + 4 * L.g) = l;
    }
}
// lane 4q+p of a 16-lane group passes the address of row q, columns 4p..4p+3 of the block; lane i gets column i, rows 0..3.
// All 64 lanes must be active.
__device__ __forceinline__ bf4 dvs_tr_read(const dvs_bf16* p) {
#ifndef DVS_EMU
    typedef short s4_ __attribute__((ext_vector_type(4)));
    const s4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4_ __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(bf4, v);
#else
    const unsigned long long a = (unsigned long long)p;
    const int lane = (int)(threadIdx.x & 63), grp = lane & 48, i = lane & 15;
    bf4 out;
    for (int q = 0; q < 4; ++q) {
        const int src = grp + 4 * q + (i >> 2);
        const unsigned lo = __shfl((unsigned)(a & 0xFFFFFFFFull), src), hi = __shfl((unsigned)(a >> 32), src);
        const dvs_bf16* row = (const dvs_bf16*)(((unsigned long long)hi << 32) | lo);
        out[q] = row[i & 3];
    }
    return out;
#endif
}
// the lane's 8-token operand slice of feature column fcol0 + r: tokens 8*(g&1) .. +7 of the parked part `tile`
__device__ __forceinline__ bf8 dvs_tr_frag(const dvs_bf16* tile, int fcol0, const Lane& L) {
    const dvs_bf16* base = tile + (8 * (L.g & 1) + (L.r >> 2)) * DVS_PLD + fcol0 + 4 * (L.r & 3);
    const bf4 a = dvs_tr_read(base), b = dvs_tr_read(base + 4 * DVS_PLD);
    bf8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[i] = a[i];
        f[4 + i] = b[i];
    }
    return f;
}
// Wave d of the workgroup parked dY at abase + d*stride and X at bbase + d*stride (hi part, lo part DVS_PKB behind).
// acc[it][reg] = dW[16*(wave&3) + 4g + reg][16*it + r] and accb[reg] = sum over tokens of dY[.][16*(wave&3) + 4g + reg] (any
// column r), both over the 4 DAGs of this wave's group: two K = 32 blocks, each the 16-token tiles of two DAGs.
__device__ __forceinline__ void dvs_coop_dw_bf(f4 (&acc)[4], f4& accb, const dvs_bf16* abase, const dvs_bf16* bbase, int stride,
                                               const Lane& L) {
    const int ot = L.wave & 3, d0 = L.wave & 4;
    bf8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (dvs_bf16)1.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int d = d0 + 2 * m + (L.g >> 1);
        const dvs_bf16* ta = abase + d * stride;
        const dvs_bf16* tb = bbase + d * stride;
        const bf8 ah = dvs_tr_frag(ta, 16 * ot, L), al = dvs_tr_frag(ta + DVS_PKB, 16 * ot, L);
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
