// Staging plans: what a phase of a chained launch (k_fwd_stack / k_bwd_stack) keeps in LDS besides its per-wave
// scratch — weight images (16-byte segments copied verbatim from the per-step images, dvs_wimg.h) and small vectors
// (biases, LayerNorm parameters; optionally permuted by dvs_pi) — described as data, so that the PREVIOUS phase can fetch
// it while its own tail runs:
//
//     phase p:   [DAG loop] -> older wave group: issue(plan of p+1): global loads into registers, nothing waits for them
//                -> workgroup barrier (every wave is done with p's images)
//                -> younger group: epilogue of p (gradient partials through LDS above DVS_BWD_EPI_FLOOR, dvs_bwd_phases.h)
//                   older group:   commit(plan of p+1): registers -> LDS below that floor    -> workgroup barrier
//     phase p+1: [DAG loop] ...
//
// Round 1 staged every phase's images behind the closing barrier of the phase before: 5-9 k cycles per phase (8-15 % of a
// backward phase) in which no wave of the workgroup computes.  With the loads issued ahead of that barrier
// their latency hides behind the barrier wait (the older wave group idles 8-20 k cycles there) and the epilogue.  The
// registers are free at that point — the DAG loop's state is dead — and live only inside the phase function's tail, so
// nothing is carried round the phase loop (round 1's attempt carried them and the compiler spilled every one).
#pragma once
#include "dvs_device.h"

constexpr int DVS_PLAN_VECS = 6;        // <= 8: two units per wave of a 4-wave tail (dvs_prefetch_issue)
struct DvsStagePlan {
    const f4* src;                     // global source of the image block (16-byte aligned): ONE contiguous block per phase —
                                       // dvs_wimg.h lays the per-step images out so that every phase kind finds its own
    int dst16;                         // LDS destination, in 16-byte units from the start of dynamic LDS
    int n16;                           // length in 16-byte units, a multiple of 64 (whole 1 KB wave chunks)
    // small vectors (biases, LayerNorm parameters), cut into units of <= 64 floats: unit j = elements vbase[j] .. vbase[j] + 63
    // of the vector at vsrc[j] (vlen[j] elements in all, 0: unused unit) -> LDS floats vdst[j] + vbase[j] ..
    const float* vsrc[DVS_PLAN_VECS];
    unsigned short vdst[DVS_PLAN_VECS];      // (16-bit: LDS float indices stay below 40 960; the 9-phase tables must fit the
    unsigned short vbase[DVS_PLAN_VECS];     // 4 KB kernel-argument block)
    unsigned short vlen[DVS_PLAN_VECS];
    int vperm;                         // bit j: unit j is read through dvs_pi (attention slot order)
    int nvec;
    int zero_int;                      // LDS int index of two group-barrier counters to clear, or -1
    int phase;                         // index of the phase inside its chained launch (diagnostic stamps, tools/phase_stamps.py)
    // COLD inputs of the phase's DAG loop: tile buffers last written a whole pass ago (saved forward activations, the decoder
    // memory) — as opposed to the tiles the same wave wrote one phase earlier.  The PREVIOUS phase's tail touches the first
    // round's tiles of these (dvs_touch_first) so that every wave of the chip does not open the phase with the same HBM burst.
    const float* cold[2];              // unused entries repeat a valid pointer (unconditional loads)
    int cold_tiles;                    // tiles in each buffer (B * NT); 0: nothing to touch
};

// Diagnostic build only (make stamps -> libdvs_hip_stamps.so, never shipped or loaded by the package): lane 0 of every
// wave records s_memtime at fixed points of every phase; tools/phase_stamps.py turns them into a per-phase time budget.
#ifdef DVS_STAMPS
constexpr int DVS_STAMP_IDS = 8, DVS_STAMP_PHASES = 32, DVS_STAMP_WAVES = 8, DVS_STAMP_WGS = 256;
#define DVS_STAMP_DECL(name) __device__ unsigned long long name[DVS_STAMP_WGS * DVS_STAMP_WAVES * DVS_STAMP_PHASES * DVS_STAMP_IDS]
#define DVS_STAMP(buf, pp, id)                                                                                              \
    do {                                                                                                                    \
        if ((dvs_tid() & 63) == 0 && dvs_bid() < DVS_STAMP_WGS && (dvs_tid() >> 6) < DVS_STAMP_WAVES)                        \
            buf[(((size_t)dvs_bid() * DVS_STAMP_WAVES + (dvs_tid() >> 6)) * DVS_STAMP_PHASES + ((pp)->phase & 31)) *          \
                    DVS_STAMP_IDS + (id)] = __builtin_amdgcn_s_memtime();                                                    \
    } while (0)
#else
#define DVS_STAMP(buf, pp, id) ((void)0)
#endif

// Plans are built on the HOST by the launchers (the LDS layout functions are __host__ __device__) and travel in the kernel
// argument block: device code reads their fields where it needs them (scalar loads) instead of deriving ~40 uniform values
// per phase in registers — computed on the device they cost the chained kernels 171 SGPR spills.
#ifndef DVS_EMU
#define DVS_HD __host__ __device__
#else
#define DVS_HD
#endif
#define DVS_FAKE_LDS ((char*)(uintptr_t)(1u << 20))      // host-side stand-in for the dynamic-LDS base: only differences are used
DVS_HD inline void dvs_plan_clear(DvsStagePlan& p) {
    p.src = nullptr;
    p.dst16 = 0;
    p.n16 = 0;
    for (int i = 0; i < DVS_PLAN_VECS; ++i) {
        p.vsrc[i] = nullptr;
        p.vdst[i] = 0;
        p.vbase[i] = 0;
        p.vlen[i] = 0;
    }
    p.phase = 0;
    p.nvec = 0;
    p.vperm = 0;
    p.zero_int = -1;
    p.cold[0] = p.cold[1] = nullptr;
    p.cold_tiles = 0;
}
DVS_HD inline void dvs_plan_cold(DvsStagePlan& p, const float* a, const float* b, int tiles) {
    p.cold[0] = a ? a : b;
    p.cold[1] = b ? b : a;
    p.cold_tiles = (a || b) ? tiles : 0;
}
DVS_HD inline void dvs_plan_seg(DvsStagePlan& p, const char* smem, const void* lds_dst, const void* src, int n_bf16) {
    p.src = (const f4*)src;
    p.dst16 = (int)(((const char*)lds_dst - smem) >> 4);
    p.n16 = n_bf16 >> 3;               // every image is a multiple of 64 rows x 144 bytes = 9 wave chunks
}
DVS_HD inline void dvs_plan_vec(DvsStagePlan& p, const char* smem, const float* lds_dst, const float* src, int n,
                                bool perm = false) {
    for (int base = 0; base < n; base += 64) {
        if (p.nvec >= DVS_PLAN_VECS) {     // host-side builder only: a plan that does not fit is a programming error of the
            p.nvec = DVS_PLAN_VECS + 1;    // launcher; it is reported (dvs_plan_ok -> code 20), never written past the arrays
            return;
        }
        p.vsrc[p.nvec] = src;
        p.vdst[p.nvec] = (unsigned short)(((const char*)lds_dst - smem) >> 2);
        p.vbase[p.nvec] = (unsigned short)base;
        p.vlen[p.nvec] = (unsigned short)n;
        if (perm) p.vperm |= 1 << p.nvec;
        ++p.nvec;
    }
}

// Registers of one thread's share of a plan: NCH x 16 bytes of image data and one float per small vector.  In the chained
// kernels only the OLDER wave group (waves 0-3, DVS_PF_THREADS threads) fetches: it leaves its DAG loop 5-17 k cycles ahead
// of the younger group (age-priority arbitration between the two waves of a SIMD; tools/phase_stamps.py) and idles at the
// closing barrier anyway, while the younger group's tail is the workgroup's critical path — issuing its share there cost
// that path 2.3-3.5 k cycles per phase.  256 threads: the backward's largest plan (attention: 73 KB) needs 18 chunks, the
// forward's (attention x6 images: 108 KB) 27.  Per-phase launches stage with the whole workgroup (9 / 14 chunks at 512).
constexpr int DVS_PF_THREADS = 256;
constexpr int DVS_PF_BWD = 9, DVS_PF_FWD = 14, DVS_PF_BWD_TAIL = 18, DVS_PF_FWD_TAIL = 27;
template <int NCH>
struct DvsPrefetch {
    f4 v[NCH];
    float s[2];                        // vector units wave and wave + #waves (#waves >= 4, DVS_PLAN_VECS <= 8)
};

// The device functions below take the plan through a pointer type PP: a plain pointer (per-phase kernels: the plan is a
// kernel argument of its own) or a pointer into the CONSTANT address space (chained kernels: k_bwd_stack / k_fwd_stack read
// their plan table straight from the kernel-argument segment, __builtin_amdgcn_kernarg_segment_ptr).  Indexing the table
// through a reference to the by-value argument struct made hipcc copy the whole 3.7 KB argument block to scratch memory and
// route EVERY argument access through it (554 scratch loads in k_bwd_stack).
#ifndef DVS_EMU
typedef const __attribute__((address_space(4))) DvsStagePlan* DvsPlanK;       // plan inside the kernel-argument segment
#else
typedef const DvsStagePlan* DvsPlanK;
#endif

// Every load below is UNCONDITIONAL (indices are clamped, unused entries point at valid memory: dvs_plan_seal): a load
// under a runtime condition makes hipcc branch around it and wait for it (vmcnt(0)) before the next one — 9-14 dependent
// L2 round trips per phase instead of one batch in flight (measured: the first version of this file, with `if (k < total)
// v = *src`, made the chained kernels 10-20 % SLOWER than staging behind the barrier).  Only the LDS stores are predicated.
DVS_HD inline bool dvs_plan_ok(const DvsStagePlan& p) { return p.nvec <= DVS_PLAN_VECS; }
void dvs_note_hip_error(const char* what, int hip_error, const char* hip_message);       // dvs_api.hip: the call returns code 20
DVS_HD inline void dvs_plan_seal(DvsStagePlan& p) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if (!dvs_plan_ok(p)) {          // dvs_plan_vec refused a unit: surface it as a failed launch instead of staging a partial plan
        dvs_note_hip_error("staging plan", -1, "more than DVS_PLAN_VECS vector units (launcher bug)");
        p.nvec = DVS_PLAN_VECS;
    }
#endif
    for (int i = 0; i < DVS_PLAN_VECS; ++i)
        if (i >= p.nvec || !p.vsrc[i] || p.vlen[i] <= 0) {
            p.vsrc[i] = (const float*)p.src;
            p.vbase[i] = 0;
            p.vlen[i] = 0;
        }
}

// The tail of a phase runs on ONE wave per SIMD (the older group), so every instruction costs its full issue latency
// (~5 cycles) and the INSTRUCTION COUNT of issue + commit is what the critical path pays: the first versions spent 17-20
// instructions per 16-byte slot on per-lane chunk arithmetic (two segments, rotation, range checks) and took 4.3-8.5 k
// cycles per phase for the commit alone — as much warm as cold, i.e. neither load latency nor instruction fetch
// (profiles/r02_phase_stamps.txt).  Now: one segment, work handed out in whole 1 KB WAVE chunks (chunk c -> wave c mod
// #waves; slot u of a wave is chunk u * #waves + wave), so that liveness, rotation and addresses are scalar arithmetic and a
// slot costs ~4 instructions in the issue and ~3 in the commit; the small vectors are handed out one unit per wave.
//  * fields are read as VALUES: `(s1 ? p->dst16[1] : p->dst16[0])` is an lvalue, clang then selects the ADDRESS per lane and
//    emits a VECTOR load from the kernel-argument segment per slot, each followed by s_waitcnt vmcnt(0) (18-27 serialised
//    L2 round trips: the 5.7-8.5 k version).
//  * ROT (first phase of a launch, per-phase kernels): every workgroup of the launch fetches the SAME images at the same
//    time, each starts at a different chunk so that they do not all queue on the same L2 lines.  Tails are already
//    de-synchronised by their DAG loops and skip it.
__device__ __forceinline__ int dvs_uniform(int v) {
#ifndef DVS_EMU
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
struct DvsPlanHead {
    const char* src;
    int dst, nwc, rot, wave, nw, lane16;
};
template <bool ROT, class PP>
__device__ __forceinline__ DvsPlanHead dvs_plan_head(PP p, int tid, int step) {
    DvsPlanHead h;
    h.src = (const char*)p->src;
    h.dst = p->dst16 * 16;
    h.nwc = p->n16 >> 6;
    h.rot = ROT ? (int)(((unsigned)dvs_bid() * 2654435761u) % (unsigned)(h.nwc > 0 ? h.nwc : 1)) : 0;
    h.wave = dvs_uniform(tid >> 6);
    h.nw = step >> 6;
    h.lane16 = (tid & 63) * 16;
    return h;
}
// slot u of this wave -> chunk (live or not) and its byte offset inside the block
template <bool ROT>
__device__ __forceinline__ bool dvs_plan_slot(const DvsPlanHead& h, int u, int& byte) {
    int c = u * h.nw + h.wave;
    const bool live = c < h.nwc;
    if (ROT) {
        c += h.rot;
        c = c >= h.nwc ? c - h.nwc : c;
    }
    byte = ((live ? c : 0) << 10) + h.lane16;
    return live;
}

// tid / step: index of this thread among the `step` threads (a multiple of 64) that share the plan — the whole workgroup, or
// its first DVS_PF_THREADS threads in a chained kernel's tail; callers keep the other threads out
template <bool ROT, int NCH, class PP>
__device__ __forceinline__ void dvs_prefetch_issue(DvsPrefetch<NCH>& pf, PP p, int tid, int step) {
    const DvsPlanHead h = dvs_plan_head<ROT>(p, tid, step);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        int byte;
        dvs_plan_slot<ROT>(h, u, byte);
        pf.v[u] = *(const f4*)(h.src + (unsigned)byte);
    }
    // vector units: wave w fetches units w and w + #waves (uniform index into the plan: scalar loads), every load unconditional
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = h.wave + q * h.nw, jc = j < DVS_PLAN_VECS ? j : 0;
        const int idx = p->vbase[jc] + (tid & 63);
        const int ic = (j < DVS_PLAN_VECS && idx < p->vlen[jc]) ? idx : 0;
        pf.s[q] = p->vsrc[jc][((p->vperm >> jc) & 1) ? dvs_pi(ic) : ic];
    }
}
template <bool ROT, int NCH, class PP>
__device__ __forceinline__ void dvs_prefetch_commit(const DvsPrefetch<NCH>& pf, PP p, char* smem, int tid, int step) {
    const DvsPlanHead h = dvs_plan_head<ROT>(p, tid, step);
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        int byte;
        if (dvs_plan_slot<ROT>(h, u, byte)) *(f4*)(smem + h.dst + byte) = pf.v[u];
    }
    float* ldsf = (float*)smem;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = h.wave + q * h.nw, jc = j < DVS_PLAN_VECS ? j : 0;
        const int idx = p->vbase[jc] + (tid & 63);
        if (j < DVS_PLAN_VECS && idx < p->vlen[jc]) ldsf[p->vdst[jc] + idx] = pf.s[q];
    }
    if (p->zero_int >= 0 && tid < 2) ((int*)smem)[p->zero_int + tid] = 0;
}
// Touch the tiles this wave will load first in the NEXT phase (round 0 of its DAG loop: tile bid * nw + wave) from the phase's
// cold buffers: one dword per 64-byte piece of the 4 KB tile = ONE load instruction per tile, whose only purpose is to start the
// line fills while the tail of the current phase runs (barriers, gradient flush, image commit: 5-8 k cycles).  Without it all
// 2 048 waves of the chip open a backward phase with the same ~16 MB burst of cold reads and both waves of every SIMD sit in it
// (the first DAG round of the backward chains costs ~0.1 ms per step more than a later round at B = 4096).
// The values are kept alive until dvs_touch_done at the end of the phase function — a register each, never used.
struct DvsTouch {
    float v[2];
};
template <class PP>
__device__ __forceinline__ DvsTouch dvs_touch_first(PP next, bool has_next, int nw, int wave, int lane) {
    DvsTouch t;
    t.v[0] = t.v[1] = 0.f;
    if (has_next && next->cold_tiles > 0) {
        int tile = dvs_bid() * nw + wave;
        tile = tile < next->cold_tiles ? tile : 0;
        const size_t off = (size_t)tile * 1024 + (size_t)lane * 16;
        t.v[0] = next->cold[0][off];
        t.v[1] = next->cold[1][off];
    }
    return t;
}
__device__ __forceinline__ void dvs_touch_done(const DvsTouch& t) {
#ifndef DVS_EMU
    asm volatile("" ::"v"(t.v[0]), "v"(t.v[1]));
#else
    (void)t;
#endif
}

// Stage a plan right away (first phase of a launch, per-phase launches): all loads in flight, then the stores.  The plan
// must fit NCH chunks per thread at this workgroup size (a 1024-thread workgroup needs half as many).
template <int NCH, class PP>
__device__ __forceinline__ void dvs_stage_now(PP p, char* smem) {
    DvsPrefetch<NCH> pf;
    dvs_prefetch_issue<true>(pf, p, dvs_tid(), (int)blockDim.x);
    dvs_prefetch_commit<true>(pf, p, smem, dvs_tid(), (int)blockDim.x);
}
