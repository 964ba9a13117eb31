// Backward kernels of the transformer stack (phases: dvs_bwd_phases.h) and the slab reduce.
#include "dvs_bwd_phases.h"

__global__ __launch_bounds__(512) void k_ffn_bwd(FfnBwdArgs a) {
    DVS_DYN_LDS(smem);
    dvs_ffn_bwd_phase(a, smem);
}
void dvs_launch_ffn_bwd(const FfnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = ffnb_lds_bytes();
    DVS_SET_LDS(k_ffn_bwd, lds);
    DVS_LAUNCH(k_ffn_bwd, dim3(grid), dim3(512), lds, st, a);
}

template <int NPROJ>
__global__ __launch_bounds__(512) void k_proj_bwd(ProjBwdArgs a) {
    DVS_DYN_LDS(smem);
    dvs_proj_bwd_phase<NPROJ>(a, smem);
}

void dvs_launch_proj_bwd(const ProjBwdArgs& a, int nproj, int grid, dvs_stream_t st) {
    const size_t bytes = projb_lds_bytes(nproj);
    if (nproj == 3) {
        DVS_SET_LDS(k_proj_bwd<3>, bytes);
        DVS_LAUNCH(k_proj_bwd<3>, dim3(grid), dim3(512), bytes, st, a);
    } else if (nproj == 2) {
        DVS_SET_LDS(k_proj_bwd<2>, bytes);
        DVS_LAUNCH(k_proj_bwd<2>, dim3(grid), dim3(512), bytes, st, a);
    } else {
        DVS_SET_LDS(k_proj_bwd<1>, bytes);
        DVS_LAUNCH(k_proj_bwd<1>, dim3(grid), dim3(512), bytes, st, a);
    }
}

__global__ __launch_bounds__(512) void k_attn_bwd(AttnBwdArgs a) {
    DVS_DYN_LDS(smem);
    dvs_attn_bwd_phase(a, smem);
}

void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = attnb_lds_floats() * 4;
    DVS_SET_LDS(k_attn_bwd, lds);
    DVS_LAUNCH(k_attn_bwd, dim3(grid), dim3(512), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Up to DVS_STACK_PHASES consecutive phases of the stack in ONE launch (one-tile path).  A launch boundary costs ~4-5 us
// on this part (dispatch + end-of-kernel cache write-back) against 20-50 us of work per phase; the phases of a chain
// need nothing from each other but the workgroup's own tiles.  TAG only names the launch in profiles.
// ---------------------------------------------------------------------------------------------------------
template <int TAG>
__global__ __launch_bounds__(512) void k_bwd_stack(BwdStackArgs s) {
    DVS_DYN_LDS(smem);
    for (int i = 0; i < s.nphase; ++i) {
        const BwdPhase& ph = s.ph[i];
        switch (ph.kind) {
            case DVS_PH_FFN: dvs_ffn_bwd_phase(ph.u.f, smem); break;
            case DVS_PH_ATTN: dvs_attn_bwd_phase(ph.u.a, smem); break;
            case DVS_PH_PROJ1: dvs_proj_bwd_phase<1>(ph.u.p, smem); break;
            case DVS_PH_PROJ2: dvs_proj_bwd_phase<2>(ph.u.p, smem); break;
            default: dvs_proj_bwd_phase<3>(ph.u.p, smem); break;
        }
        __syncthreads();     // the epilogue's LDS scratch is the next phase's weight image; global tiles: same wave, same CU
    }
}

void dvs_launch_bwd_stack(const BwdStackArgs& s, int tag, int grid, dvs_stream_t st) {
    size_t lds = 0;
    for (int i = 0; i < s.nphase; ++i) {
        const int k = s.ph[i].kind;
        const size_t b = k == DVS_PH_FFN ? ffnb_lds_bytes() : k == DVS_PH_ATTN ? attnb_lds_floats() * 4 : projb_lds_bytes(k - DVS_PH_PROJ1 + 1);
        lds = b > lds ? b : lds;
    }
    if (tag == 0) {
        DVS_SET_LDS(k_bwd_stack<0>, lds);
        DVS_LAUNCH(k_bwd_stack<0>, dim3(grid), dim3(512), lds, st, s);
    } else {
        DVS_SET_LDS(k_bwd_stack<1>, lds);
        DVS_LAUNCH(k_bwd_stack<1>, dim3(grid), dim3(512), lds, st, s);
    }
}

// ---------------------------------------------------------------------------------------------------------
// grads[p] = sum over slabs (fixed order)
// ---------------------------------------------------------------------------------------------------------
// One workgroup = 64 float4 columns x 4 slab quarters: four times the loads in flight of a thread-per-column walk
// (the kernel is a 230 MB strided read: 66 -> 51 us; eight ranges: no further gain); the quarters meet in LDS and are
// added in fixed order.
__global__ __launch_bounds__(256) void k_reduce_slabs(ReduceArgs a) {
    __shared__ f4 part[4][64];
    const int q = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int64_t i4 = ((int64_t)blockIdx.x * 64 + c) * 4;
    const bool in = i4 < a.P;
    f4 s = f4_zero();
    if (in) {
        const bool fc = (i4 >= a.fc_lo1 && i4 < a.fc_hi1) || (i4 >= a.fc_lo2 && i4 < a.fc_hi2);
        const float* src = fc ? a.fcpart : a.slab;
        const int n = fc ? DVS_FC_PARTS : a.nslab;
        const int per = (n + 3) / 4;
        const int k1 = (q + 1) * per < n ? (q + 1) * per : n;
        int k = q * per;
        for (; k + 8 <= k1; k += 8) {      // 8 independent 16-byte loads in flight per lane
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f4*)(src + (size_t)(k + u) * a.P + i4);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < k1; ++k) s += *(const f4*)(src + (size_t)k * a.P + i4);
    }
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && in) *(f4*)(a.grads + i4) = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

void dvs_launch_reduce_slabs(const ReduceArgs& a, dvs_stream_t st) {
    const int64_t n4 = (a.P + 3) / 4;
    DVS_LAUNCH(k_reduce_slabs, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, a);
}
