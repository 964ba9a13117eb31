// Backward kernels of the transformer stack (phases: dvs_bwd_phases.h) and the slab reduce.
#include "dvs_bwd_phases.h"
#include "dvs_latent_bwd.h"

// NW = waves per workgroup (dvs_api.hip: dvs_waves_per_wg): 8 — two cooperative groups, two waves per SIMD — or 4, the narrow
// mapping for batches that do not fill the chip at 8 DAGs per workgroup (one group, one wave per SIMD, twice the workgroups).
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ffn_bwd(FfnBwdArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    dvs_ffn_bwd_phase<NW>(a, smem, &plan, true, &plan, false);
}
void dvs_launch_ffn_bwd(const FfnBwdArgs& a, int grid, int nw, dvs_stream_t st) {
    const size_t lds = ffnb_lds_bytes();
    DvsStagePlan plan;
    ffnb_plan(plan, a, DVS_FAKE_LDS);
    if (nw == 4) {
        DVS_SET_LDS(k_ffn_bwd<4>, lds);
        DVS_LAUNCH_AS("k_ffn_bwd", k_ffn_bwd<4>, dim3(grid), dim3(256), lds, st, a, plan);
    } else {
        DVS_SET_LDS(k_ffn_bwd<8>, lds);
        DVS_LAUNCH_AS("k_ffn_bwd", k_ffn_bwd<8>, dim3(grid), dim3(512), lds, st, a, plan);
    }
}

template <int NPROJ, int NW>
__global__ __launch_bounds__(64 * NW) void k_proj_bwd(ProjBwdArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    dvs_proj_bwd_phase<NPROJ, NW>(a, smem, &plan, true, &plan, false);
}

#define DVS_PROJ_LAUNCH(NP, NWV)                                                               \
    do {                                                                                       \
        DVS_SET_LDS((k_proj_bwd<NP, NWV>), bytes);                                             \
        DVS_LAUNCH_AS("k_proj_bwd<" #NP ">", (k_proj_bwd<NP, NWV>), dim3(grid), dim3(64 * NWV), bytes, st, a, plan); \
    } while (0)
void dvs_launch_proj_bwd(const ProjBwdArgs& a, int nproj, int grid, int nw, dvs_stream_t st) {
    const size_t bytes = projb_lds_bytes(nproj);
    DvsStagePlan plan;
    projb_plan(plan, a, nproj, DVS_FAKE_LDS);
    if (nw == 4) {
        if (nproj == 3) DVS_PROJ_LAUNCH(3, 4);
        else if (nproj == 2) DVS_PROJ_LAUNCH(2, 4);
        else DVS_PROJ_LAUNCH(1, 4);
    } else {
        if (nproj == 3) DVS_PROJ_LAUNCH(3, 8);
        else if (nproj == 2) DVS_PROJ_LAUNCH(2, 8);
        else DVS_PROJ_LAUNCH(1, 8);
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_bwd(AttnBwdArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    dvs_attn_bwd_phase<NW>(a, smem, &plan, true, &plan, false);
}

void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, int nw, dvs_stream_t st) {
    const size_t lds = attnb_lds_floats() * 4;
    DvsStagePlan plan;
    attnb_plan(plan, a, DVS_FAKE_LDS);
    if (nw == 4) {
        DVS_SET_LDS(k_attn_bwd<4>, lds);
        DVS_LAUNCH_AS("k_attn_bwd", k_attn_bwd<4>, dim3(grid), dim3(256), lds, st, a, plan);
    } else {
        DVS_SET_LDS(k_attn_bwd<8>, lds);
        DVS_LAUNCH_AS("k_attn_bwd", k_attn_bwd<8>, dim3(grid), dim3(512), lds, st, a, plan);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Up to DVS_STACK_PHASES consecutive phases of the stack in ONE launch (one-tile path).  A launch boundary costs ~4-5 us
// on this part (dispatch + end-of-kernel cache write-back) against 20-50 us of work per phase; the phases of a chain
// need nothing from each other but the workgroup's own tiles.  TAG only names the launch in profiles.
// ---------------------------------------------------------------------------------------------------------
template <int TAG, int NW>
__global__ __launch_bounds__(64 * NW) void k_bwd_stack(BwdStackArgs s) {
    DVS_DYN_LDS(smem);
    // the plan table is read straight from the kernel-argument segment (dvs_stage.h); the struct is the kernel's only
    // explicit argument, so it starts at offset 0 of the segment
#ifndef DVS_EMU
    const DvsPlanK plans =
        ((const __attribute__((address_space(4))) BwdStackArgs*)__builtin_amdgcn_kernarg_segment_ptr())->plan;
#else
    const DvsPlanK plans = s.plan;
#endif
    if (TAG == 1 && NW == 8 && s.has_latent) {
        // latent block on the workgroup's DAGs, two 8-DAG runs (= two rounds of the phases' DAG loops) per MFMA group; its
        // d enc_out tiles are read by other waves of this workgroup in phase 0: __syncthreads waits for the stores (vmcnt)
        const int step = (int)gridDim.x * 8, B = s.lat.dims.B;
        for (int base = dvs_bid() * 8; base < B; base += 2 * step) {
            dvs_latent_bwd_group<8>(s.lat, (f4 (*)[2][64])smem, base, base + step);
            __syncthreads();
        }
    }
    for (int i = 0; i < s.nphase; ++i) {
        const BwdPhase& ph = s.ph[i];
        const bool first = i == 0;                       // later phases: staged by the tail of the one before (dvs_stage.h)
        const bool more = i + 1 < s.nphase;
        const DvsPlanK mine = plans + i;
        const DvsPlanK next = plans + (more ? i + 1 : i);
        switch (ph.kind) {
            case DVS_PH_FFN: dvs_ffn_bwd_phase<NW>(ph.u.f, smem, mine, first, next, more); break;
            case DVS_PH_ATTN: dvs_attn_bwd_phase<NW>(ph.u.a, smem, mine, first, next, more); break;
            case DVS_PH_PROJ1: dvs_proj_bwd_phase<1, NW>(ph.u.p, smem, mine, first, next, more); break;
            case DVS_PH_PROJ2: dvs_proj_bwd_phase<2, NW>(ph.u.p, smem, mine, first, next, more); break;
            default: dvs_proj_bwd_phase<3, NW>(ph.u.p, smem, mine, first, next, more); break;
        }
        // the barrier that publishes the next phase's staged images (tail commit) is taken by that phase itself, behind its
        // first round's global loads (DVS_PHASE_GATE, dvs_kernels.h); global tiles: same wave, same queue
    }
}

void dvs_launch_bwd_stack(const BwdStackArgs& s_in, int tag, int grid, int nw, dvs_stream_t st) {
    BwdStackArgs s = s_in;
    for (int i = 0; i < s.nphase; ++i) {
        dvs_bwd_plan(s.plan[i], s.ph[i], DVS_FAKE_LDS);
        s.plan[i].phase = i;
    }
#ifdef DVS_STAMPS
    static int stamp_seq = 0;            // three chained launches per step: decoder 1, decoder 2, encoder
    for (int i = 0; i < s.nphase; ++i) s.plan[i].phase = (stamp_seq % 3) * DVS_STACK_PHASES + i;
    ++stamp_seq;
#endif
    size_t lds = 0;
    for (int i = 0; i < s.nphase; ++i) {
        const int k = s.ph[i].kind;
        const size_t b = k == DVS_PH_FFN ? ffnb_lds_bytes() : k == DVS_PH_ATTN ? attnb_lds_floats() * 4 : projb_lds_bytes(k - DVS_PH_PROJ1 + 1);
        lds = b > lds ? b : lds;
    }
#define DVS_BSTACK_LAUNCH(TG, NWV)                                                             \
    do {                                                                                       \
        DVS_SET_LDS((k_bwd_stack<TG, NWV>), lds);                                              \
        DVS_LAUNCH_AS("k_bwd_stack<" #TG ">", (k_bwd_stack<TG, NWV>), dim3(grid), dim3(64 * NWV), lds, st, s); \
    } while (0)
    if (nw == 4) {
        if (tag == 0) DVS_BSTACK_LAUNCH(0, 4);
        else DVS_BSTACK_LAUNCH(1, 4);
    } else {
        if (tag == 0) DVS_BSTACK_LAUNCH(0, 8);
        else DVS_BSTACK_LAUNCH(1, 8);
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
    if (q == 0) {                                  // wave 0: the 64 columns' totals, and (optionally) their sum of squares
        const f4 tot = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (in) *(f4*)(a.grads + i4) = tot;
        if (a.sqpart) {
            // entries beyond P inside the last float4 are padding of the flat layout: written as computed, NOT counted
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += (in && i4 + e < a.P) ? tot[e] * tot[e] : 0.f;
            ss = dvs_sum_wave(ss);                 // fixed order: bitwise reproducible
            if (c == 0) a.sqpart[blockIdx.x] = ss;
        }
    }
}

void dvs_launch_reduce_slabs(const ReduceArgs& a, dvs_stream_t st) {
    const int64_t n4 = (a.P + 3) / 4;
    DVS_LAUNCH(k_reduce_slabs, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, a);
}

#ifdef DVS_STAMPS
extern "C" int dvs_debug_read_stamps_bwd(void* out, size_t bytes, int clear) {
    if (bytes > sizeof(dvs_stamps_bwd)) bytes = sizeof(dvs_stamps_bwd);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dvs_stamps_bwd), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dvs_stamps_bwd)) != hipSuccess || hipMemset(p, 0, sizeof(dvs_stamps_bwd)) != hipSuccess) return 2;
    }
    return 0;
}
#endif
