// Wide path (17..48 tokens per DAG), backward kernels: attention core, loss head, embedding.
// See dvs_wide.h for the execution model.  Persistent workgroups of 4 waves (grid = number of gradient slabs); wave w
// owns tile w of the workgroup's current DAG and accumulates weight gradients in MFMA accumulators exactly like the
// one-tile kernels; the workgroup's waves are added in fixed order through LDS and written to the workgroup's slab.
#include "dvs_wide.h"
#include "dvs_wimg.h"

// ---------------------------------------------------------------------------------------------------------
// Attention-core backward, wide: d(pre) -> d q, d k, d v projections (frag tiles, natural feature order) and dWo, dbo.
// Nothing but the sublayer input was saved: q, k, v and the probabilities are recomputed.
//   stage 1  : x -> q,k,v (bf16x3 from the per-step images: waves 0..2NT-1 = (tile, half of the in-projection rows)) and
//              dO^T = Wo^T dropout-mask(d pre) (the remaining waves, by output tile), parked in LDS   | barrier
//   core     : wave h = head h, fp32 MFMA over the NON-EMPTY (query tile, key tile) pairs of the DAG's mask, both score
//              orientations (attnwb_core): row statistics, O (for dWo), dq; then dk, dv, which overwrite the head's own
//              columns of the K, V buffers                                                            | barrier
//   wave w   : stores dq, dk, dv tiles; dWo += dy(N)^T O(N) in MFMA accumulators                  | barrier
// Round 2's core walked ancestor / descendant bit-rows with one lane per (token, head) on the VALU (35 k of the kernel's 48 k
// cycles per DAG at n = 37, matrix pipe 5 % busy, 68 % of the wave cycles waiting: the walk is a chain of dependent LDS reads).
// ---------------------------------------------------------------------------------------------------------
constexpr int ATTNWB_TLD = 20;                   // floats per row of a transposing scratch tile (16 + 4: 16-byte rows, 2-way banks at most)
constexpr int ATTNWB_TSCR = 2 * 16 * ATTNWB_TLD; // per wave: dS and P' tiles of the current pair (attnwb_core)
// Two sets of {Q, K, V, dO} ([48][DVS_LD] fp32 each): the core works on one while the waves the tail leaves idle fill the other
// with the workgroup's next DAG.  No in-projection images: q, k, v are read back from the forward (AttnArgs::qkv).
struct AttnWBLds {
    dvs_bf16 *WoTh, *WoTl;       // bf16x3 pair of Wo^T (dvs_wimg.h: WoutT)
    float *outb, *Q, *K, *V, *DO, *tscr;      // Q / K / V / dO of the CURRENT set; they end the core as dq / dk / dv / O (attnwb_core)
    uint64_t* al;                // [48] ancestor bit-rows (whom token i attends) of the current set
};
constexpr size_t ATTNWB_SET = 4 * (size_t)DVS_WSCR;      // floats of one buffer set
__device__ __forceinline__ AttnWBLds attnwb_lds(char* smem, int set) {
    AttnWBLds l;
    l.WoTh = (dvs_bf16*)smem;
    l.WoTl = l.WoTh + 64 * DVS_LDB;
    l.outb = (float*)(l.WoTl + 64 * DVS_LDB);
    l.Q = l.outb + 64 + set * ATTNWB_SET;
    l.K = l.Q + DVS_WSCR;
    l.V = l.K + DVS_WSCR;
    l.DO = l.V + DVS_WSCR;
    l.tscr = l.outb + 64 + 2 * ATTNWB_SET;       // [8 waves][2][16][ATTNWB_TLD]
    l.al = (uint64_t*)(l.tscr + 8 * ATTNWB_TSCR) + set * DVS_WTOK;   // (offset is a multiple of 8 bytes)
    return l;
}
constexpr size_t ATTNWB_FLOATS = 2 * DVS_IMG64 / 2 + 64 + 2 * ATTNWB_SET + 8 * ATTNWB_TSCR + 2 * 2 * DVS_WTOK;
static_assert(ATTNWB_FLOATS * 4 <= 160 * 1024, "k_attn_bwd_w LDS");


#ifdef DVS_STAMPS
// per-stage s_memtime totals of k_attn_bwd_w (tools/wide_stamps.py): [workgroup][wave][stage], summed over DAGs and launches
__device__ unsigned long long dvs_stamps_wb[256 * 8 * 8];
#define WBSTAMP(k)                                                                                          \
    do {                                                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) dvs_stamps_wb[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] += now_ - wst_; \
        wst_ = now_;                                                                                        \
    } while (0)
extern "C" int dvs_debug_read_stamps_wb(void* out, size_t bytes, int clear) {
    if (bytes > sizeof(dvs_stamps_wb)) bytes = sizeof(dvs_stamps_wb);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dvs_stamps_wb), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dvs_stamps_wb)) != hipSuccess || hipMemset(p, 0, sizeof(dvs_stamps_wb)) != hipSuccess) return 2;
    }
    return 0;
}
// ... and of the two head kernels: ids 0-3 k_embed_bwd_w (tile stage, barrier, scatters, barrier), 4-7 k_loss_bwd_w (heads + U / V,
// pass 1 incl. barrier, pass 2 incl. barrier, edge matrices + d h + store incl. barrier)
__device__ unsigned long long dvs_stamps_wc[256 * 8 * 8];
#define WCSTAMP(k)                                                                                          \
    do {                                                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) dvs_stamps_wc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (k)] += now_ - wst_; \
        wst_ = now_;                                                                                        \
    } while (0)
extern "C" int dvs_debug_read_stamps_wc(void* out, size_t bytes, int clear) {
    if (bytes > sizeof(dvs_stamps_wc)) bytes = sizeof(dvs_stamps_wc);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dvs_stamps_wc), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dvs_stamps_wc)) != hipSuccess || hipMemset(p, 0, sizeof(dvs_stamps_wc)) != hipSuccess) return 2;
    }
    return 0;
}
#else
#define WBSTAMP(k) ((void)0)
#define WCSTAMP(k) ((void)0)
#endif

// ---- MFMA core of the wide attention backward ----------------------------------------------------------------------------
// Wave h owns head h: it touches ONLY the head's 8 feature columns c0 = 8h .. of the parked [48][DVS_LD] buffers, so the eight
// waves need no synchronisation among themselves, and every result overwrites its own operand in place: dq the rows of Q, O the
// rows of dO (per query tile, once the tile's fragments are in registers), dk / dv the K and V buffers (after the last query
// tile).  Every MFMA operand of the head is loaded from LDS, as two kinds of fragments per 16-token tile t:
//   row[s]   = X[16t + r][c0 + 4s + g]          s = 0, 1     contraction over the head's 8 features (two K = 4 steps)
//   col[reg] = X[16t + 4g + reg][c0 + (r & 7)]  reg = 0..3   contraction over the tile's 16 tokens (four K = 4 steps); A-operand
//                                                            rows r >= 8 duplicate r - 8 and their result rows are ignored
// ONE pass over the query tiles, in the T orientation  S^T = K Q^T  (reg <-> key j = 16jt + 4g + reg, lane r <-> query i): row
// statistics in-lane + 2 swaps, and a D-layout register `reg` IS the B operand of contraction step `reg` (k = g <-> token
// 4g + reg) for dq and O.  dk and dv contract over the QUERIES, i.e. need dS and P' with lane r <-> key j: the two 16 x 16 tiles
// of a pair go through a per-wave LDS scratch (one 16-byte write per lane and tile, four 4-byte transposed reads) — round 3's
// first version recomputed scores, exponentials and the dropout draws of every pair in a second pass with the operands
// swapped (8.7 k of the core's 26 k cycles per DAG, `profiles/r03_wide_stamps_bwd.txt`).  Tile pairs whose 16 x 16 block of the
// mask is empty — all pairs above the diagonal for DAGs in topological vertex order, more for sparse ones — are skipped:
// `pairs` bit 3 it + jt.  Per pair 20 MFMAs.
// fragments are fetched where they are used (cheap: 4-byte LDS reads; the kernel is register-bound, not LDS-bound)
__device__ __forceinline__ void attnwb_row(float (&f)[2], const float* X, int t, int c0, const Lane& L) {
    const float* p = X + (16 * t + L.r) * DVS_LD + c0 + L.g;
    f[0] = p[0];
    f[1] = p[4];
}
__device__ __forceinline__ f4 attnwb_col(const float* X, int t, int c0, const Lane& L) {
    const float* p = X + (16 * t + 4 * L.g) * DVS_LD + c0 + (L.r & 7);
    return f4{p[0], p[DVS_LD], p[2 * DVS_LD], p[3 * DVS_LD]};
}
// keep-factors {0, scale} of the probabilities' dropout for 4 elements: T orientation (query i = lane's, keys j0 .. j0 + 3,
// j0 a multiple of 4: two draws), element index (h NTOK + i) NTOK + j as in k_attn_fwd_w
__device__ __forceinline__ f4 attnwb_mask_T(uint32_t key, int h, int i, int j0, int NTOK, const DvsDrop& D) {
    if (!D.on) return f4_splat(1.f);
    const uint32_t p0 = (uint32_t)((h * NTOK + i) * NTOK + j0) >> 1;
    const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
    f4 m;
    m[0] = ((h0 & 0xFFFFu) >= D.thr16) ? D.scale : 0.f;
    m[1] = ((h0 >> 16) >= D.thr16) ? D.scale : 0.f;
    m[2] = ((h1 & 0xFFFFu) >= D.thr16) ? D.scale : 0.f;
    m[3] = ((h1 >> 16) >= D.thr16) ? D.scale : 0.f;
    return m;
}
// a T-orientation tile (lane (r, g) holds (row r, columns 4g ..)) through the wave's scratch: out[reg] = element (row 4g + reg, column r)
__device__ __forceinline__ void attnwb_put(float* scr, const f4& v, const Lane& L) { *(f4*)(scr + L.r * ATTNWB_TLD + 4 * L.g) = v; }
__device__ __forceinline__ f4 attnwb_get(const float* scr, const Lane& L) {
    const float* p = scr + (4 * L.g) * ATTNWB_TLD + L.r;
    return f4{p[0], p[ATTNWB_TLD], p[2 * ATTNWB_TLD], p[3 * ATTNWB_TLD]};
}
__device__ __forceinline__ void attnwb_core(const AttnWBLds& l, int h, int N, int NT, uint32_t kprob, const DvsDrop& D, float scale,
                                            const Lane& L, unsigned long long& wst_) {
    const int c0 = 8 * h, NTOK = 16 * NT;
    float* const tscr = l.tscr + L.wave * ATTNWB_TSCR;
    // non-empty tile pairs, from the DAG's 48 ancestor rows: lane i < 48 holds row i
    uint32_t pairs = 0;
    {
        const uint64_t row = L.lane < DVS_WTOK ? l.al[L.lane] : 0ull;
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            const unsigned long long b = __ballot(((row >> (16 * jt)) & 0xFFFFull) != 0ull);
#pragma unroll
            for (int it = 0; it < DVS_WNT; ++it)
                if ((b >> (16 * it)) & 0xFFFFull) pairs |= 1u << (3 * it + jt);
        }
    }
    f4 dk[DVS_WNT], dv[DVS_WNT];
#pragma unroll
    for (int jt = 0; jt < DVS_WNT; ++jt) dk[jt] = dv[jt] = f4_zero();
    // (walking the query tiles in opposite directions on the two waves of a SIMD, so that one's softmax arithmetic meets the
    // other's MFMA chains: measured +1.5 % — the older wave runs ahead anyway)
    for (int it = 0; it < NT; ++it) {
        const int i = 16 * it + L.r;
        const uint64_t row = l.al[i];
        float fq[2], fg[2];
        attnwb_row(fq, l.Q, it, c0, L);
        attnwb_row(fg, l.DO, it, c0, L);
        const f4 qc = attnwb_col(l.Q, it, c0, L), gc = attnwb_col(l.DO, it, c0, L);     // for dk, dv: this tile's rows are overwritten below
        f4 sT[DVS_WNT], dpT[DVS_WNT];
        float m = -3.0e38f;
        // every key tile's row fragments in one batch of LDS reads (rows of tiles the DAG does not have are zeros): one wait for
        // the tile's score products instead of one per pair — the core is a chain of LDS and MFMA latencies on two waves per SIMD
        float fk[DVS_WNT][2], fv[DVS_WNT][2];
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            attnwb_row(fk[jt], l.K, jt, c0, L);
            attnwb_row(fv[jt], l.V, jt, c0, L);
        }
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            sT[jt] = dpT[jt] = f4_zero();
            if (!((pairs >> (3 * it + jt)) & 1u)) continue;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sT[jt] = dvs_mfma(fk[jt][s], fq[s], sT[jt]);
                dpT[jt] = dvs_mfma(fv[jt][s], fg[s], dpT[jt]);
            }
            const uint32_t ok4 = (uint32_t)(row >> (16 * jt + 4 * L.g)) & 0xFu;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) m = ((ok4 >> reg) & 1u) ? fmaxf(m, sT[jt][reg]) : m;
        }
        m = dvs_max_g(m);
        float den = 0.f;
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            if (!((pairs >> (3 * it + jt)) & 1u)) continue;
            const uint32_t ok4 = (uint32_t)(row >> (16 * jt + 4 * L.g)) & 0xFu;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                sT[jt][reg] = ((ok4 >> reg) & 1u) ? __expf(sT[jt][reg] - m) : 0.f;
                den += sT[jt][reg];
            }
        }
        den = dvs_sum_g(den);
        const float rden = den > 0.f ? 1.0f / den : 0.f;              // padding queries have no keys: all-zero rows
        float delta = 0.f;
        f4 pmk[DVS_WNT];
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            pmk[jt] = f4_zero();
            if (!((pairs >> (3 * it + jt)) & 1u)) continue;
            const f4 mk = attnwb_mask_T(kprob, h, i, 16 * jt + 4 * L.g, NTOK, D);
            sT[jt] *= rden;                                           // P
            dpT[jt] *= mk;                                            // d P' -> d P
            pmk[jt] = sT[jt] * mk;                                    // P'
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) delta += sT[jt][reg] * dpT[jt][reg];
        }
        delta = dvs_sum_g(delta);
        f4 dq = f4_zero(), o = f4_zero();
#pragma unroll
        for (int jt = 0; jt < DVS_WNT; ++jt) {
            if (!((pairs >> (3 * it + jt)) & 1u)) continue;
            f4 ds;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ds[reg] = sT[jt][reg] * (dpT[jt][reg] - delta);
            attnwb_put(tscr, ds, L);
            attnwb_put(tscr + 16 * ATTNWB_TLD, pmk[jt], L);
            dvs_wave_sync();
            // all four operands of the pair in one batch of reads, then four independent MFMA chains interleaved
            const f4 kc = attnwb_col(l.K, jt, c0, L), vc = attnwb_col(l.V, jt, c0, L);
            const f4 dsS = attnwb_get(tscr, L), pS = attnwb_get(tscr + 16 * ATTNWB_TLD, L);       // (query 4g + reg, key r)
            dvs_wave_sync();
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                dq = dvs_mfma(kc[reg], ds[reg], dq);
                o = dvs_mfma(vc[reg], pmk[jt][reg], o);
                dk[jt] = dvs_mfma(qc[reg], dsS[reg], dk[jt]);
                dv[jt] = dvs_mfma(gc[reg], pS[reg], dv[jt]);
            }
        }
        // D[row f = 4g + reg][col i = r]: the head's 8 features live in the lane groups g = 0, 1.  In place: this query tile's
        // fragments of Q and dO are in registers (in-order LDS queue of the wave: the reads above are ahead of these writes)
        dvs_wave_sync();
        if (L.g < 2) {
            *(f4*)(l.Q + i * DVS_LD + c0 + 4 * L.g) = dq * scale;
            *(f4*)(l.DO + i * DVS_LD + c0 + 4 * L.g) = o;
        }
    }
    WBSTAMP(3);
    dvs_wave_sync();
#pragma unroll
    for (int jt = 0; jt < DVS_WNT; ++jt) {
        if (jt < NT && L.g < 2) {
            const int j = 16 * jt + L.r;
            *(f4*)(l.K + j * DVS_LD + c0 + 4 * L.g) = dk[jt];
            *(f4*)(l.V + j * DVS_LD + c0 + 4 * L.g) = dv[jt];
        }
    }
}

// Fill one buffer set with DAG `dag`: q (scaled), k, v as the forward parked them (AttnArgs::qkv: one 16-byte load per lane and
// output tile, unit u = 12 tile + o12, o12: 0-3 q, 4-7 k, 8-11 v), dO^T = Wo^T dropout-mask(d pre) (bf16x3: a gradient product,
// as in the one-tile backward), the DAG's ancestor rows.  `nw` waves take part (wave index wi): every wave issues the loads of
// its share of the q / k / v units first, the first NT of them then compute the dO tile of one token tile each (a tile costs a
// load, the dropout draws and the bf16 split before its first product) — the loads land meanwhile —, then the units are parked.
constexpr int ATTNWB_MAXU = 8;                   // q / k / v units per wave and batch
__device__ __forceinline__ void attnwb_fill(const AttnWBLds& l, const AttnBwdArgs& a, int dag, int wi, int nw, const DvsDrop& D,
                                            const Lane& L, unsigned long long& wst_) {
    const int N = a.dims.N, NT = a.dims.NT, nunits = 12 * NT;
    if (wi == nw - 1 && L.lane < DVS_WTOK) {
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + dag;
        l.al[L.lane] = L.lane < N ? rec->allowed[L.lane] : 0ull;
    }
    const f4* src = (const f4*)(a.qkv + (size_t)dag * NT * 12 * 256) + L.lane;
    for (int u0 = wi; u0 < nunits; u0 += nw * ATTNWB_MAXU) {
        f4 v[ATTNWB_MAXU];
#pragma unroll
        for (int k = 0; k < ATTNWB_MAXU; ++k) {
            const int u = u0 + k * nw;
            v[k] = src[(size_t)(u < nunits ? u : u0) * 64];
        }
        WBSTAMP(1);
        if (u0 == wi && wi < NT) {               // (first batch only) this wave's token tile of dO
            const int tw = wi;
            const uint32_t gdag = a.dims.dag_offset + dag;
            f4 dyt[4];
            dvs_load_grad(dyt, a.gpre, (size_t)dag * NT + tw, dvs_rows_of(N, tw), L);
            dvs_dropout_tile(dyt, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L, 16 * tw);
            const SplitT ds = dvs_split_T(dyt);
            f4 o4[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
            dvs_matb_T<4>(o4, ds, l.WoTh, l.WoTl, 0, L);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) dvs_park_col(l.DO, 16 * tw, ot, o4[ot], L);
        }
        WBSTAMP(2);
#pragma unroll
        for (int k = 0; k < ATTNWB_MAXU; ++k) {
            const int u = u0 + k * nw;
            if (u < nunits) {
                const int t = u / 12, o12 = u - 12 * t;
                dvs_park_col(l.Q + (o12 >> 2) * DVS_WSCR, 16 * t, o12 & 3, v[k], L);       // Q, K, V are consecutive
            }
        }
    }
}

// 8 waves, two DAGs in flight per workgroup.  Per DAG:
//   core    : wave h = head h on the current buffer set (attnwb_core)                                               | barrier
//   tail 1  : waves 0..NT-1 own the tiles: store dq, dk, dv, park d y and O as bf16 pairs;
//             waves NT..7 fill the OTHER buffer set with the workgroup's next DAG (attnwb_fill)                    | barrier
//   tail 2  : waves 0..3: dWo += dy^T O cooperatively                                                              | barrier
// Nothing but LDS traffic and MFMAs sits between two cores of a workgroup besides the tile owners' stores: the loads of the next
// DAG have the whole tail to land.  Round 3's first versions recomputed q, k, v from the sublayer input inside a stage of its
// own (5-7 k cycles per DAG on six waves, dO on the other two 10 k, then the core, then a tail that left five waves idle:
// 41.9 k cycles per DAG, `profiles/r03_wide_stamps_bwd.txt`); with the in-projection images gone (55 KB) the second buffer set
// fits LDS.
__global__ __launch_bounds__(512) void k_attn_bwd_w(AttnBwdArgs a) {
    DVS_DYN_LDS(smem);
    unsigned long long wst_ = 0;
#ifdef DVS_STAMPS
    wst_ = __builtin_amdgcn_s_memtime();
#endif
    const AttnWBLds l0 = attnwb_lds(smem, 0);
    dvs_copy_image(l0.WoTh, (const dvs_bf16*)a.wimg + DvsAttnImg::WoutT, (int)(2 * DVS_IMG64));     // WoutT pair
    dvs_stage_vector(l0.outb, a.out_b, 64);
    for (int i = threadIdx.x; i < (int)(2 * ATTNWB_SET); i += blockDim.x) l0.Q[i] = 0.f;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N, NT = a.dims.NT, B = a.dims.B;
    const int tok0 = 16 * L.wave, Nl = dvs_rows_of(N, L.wave);
    const bool has_tile = L.wave < NT;
    const float scale = 0.35355339059327373f;
    // d out_proj.weight / bias, cooperatively (dvs_backward.h): wave w < 4 accumulates rows 16w .. of dWo over the DAG's tiles —
    // 16 accumulator registers per wave instead of 64 (all eight waves would carry them through the core)
    f4 aWo[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()}, abo = f4_zero();
    int cur = 0;
    WBSTAMP(0);
    // the first pass (dag < 0) only fills the other buffer set with the workgroup's first DAG: one copy of every stage's code
    for (int dag = (int)blockIdx.x - (int)gridDim.x; dag < B; dag += gridDim.x) {
        const bool real = dag >= 0;
        const size_t tile = (size_t)(real ? dag : 0) * NT + L.wave;
        const uint32_t gdag = a.dims.dag_offset + dag;
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        const AttnWBLds l = attnwb_lds(smem, cur);
        if (real) {
            // the tile owners touch their d pre tile of the tail ahead of the core (one 4-byte load per lane: every line of the tile)
            // ... and the other waves every line of what they load for the next DAG in the tail (attnwb_fill: this wave's q / k / v
            // units, 8 lines of 128 bytes each — lane = (unit, line) —, and its d pre tile): a cold load costs 4-7 k cycles here,
            // more than the tail lasts; behind the touches the tail's loads hit the cache
            float tch = 0.f, tch2 = 0.f;
            if (has_tile) {
                tch = a.gpre[tile * 1024 + (size_t)L.lane * 16];
            } else if (dag + (int)gridDim.x < B) {
                const int nd = dag + (int)gridDim.x, wi = L.wave - NT, nw = 8 - NT, nunits = 12 * NT;
                int u = wi + (L.lane >> 3) * nw;
                u = u < nunits ? u : wi;
                tch = a.qkv[((size_t)nd * NT * 12 + u) * 256 + (L.lane & 7) * 32];
                if (wi < NT) tch2 = a.gpre[((size_t)nd * NT + wi) * 1024 + (size_t)L.lane * 16];
            }
            // ---- core: wave h = head h, on the matrix pipe (attnwb_core above) ------------------------------------------
            attnwb_core(l, L.wave, N, NT, kprob, D, scale, L, wst_);
#ifndef DVS_EMU
            asm volatile("" ::"v"(tch), "v"(tch2));
#endif
        }
        WBSTAMP(4);
        __syncthreads();
        WBSTAMP(5);
        if (!real) {
        } else if (has_tile) {
            const bool valid = L.r < Nl;
            f4 dy[4];                 // this tile's masked d pre again, for dWo / dbo
            dvs_load_grad(dy, a.gpre, tile, Nl, L);
            dvs_dropout_tile(dy, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L, tok0);
            f4 g[4];
            dvs_lds_T(g, l.Q, tok0, L);                  // d q (in place, attnwb_core)
#pragma unroll
            for (int t = 0; t < 4; ++t) g[t] = valid ? g[t] : f4_zero();
            dvs_store_tile(a.gq, tile, g, L);
            dvs_lds_T(g, l.K, tok0, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) g[t] = valid ? g[t] : f4_zero();
            dvs_store_tile(a.gk, tile, g, L);
            dvs_lds_T(g, l.V, tok0, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) g[t] = valid ? g[t] : f4_zero();
            dvs_store_tile(a.gv, tile, g, L);
            // dWo += dy^T O, dbo += column sums of dy: both tiles parked as bf16 [hi | lo] pairs for the cooperative product
            // below — d y over this tile's rows of Q (free now), O converted in place (a pair is exactly the 16 fp32 rows it replaces)
            f4 o[4];
            dvs_lds_T(o, l.DO, tok0, L);                 // O (in place)
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = valid ? o[t] : f4_zero();
            dvs_wave_sync();
            dvs_park_bf((dvs_bf16*)(l.Q + tok0 * DVS_LD), dy, L);
            dvs_park_bf((dvs_bf16*)(l.DO + tok0 * DVS_LD), o, L);
        }
        if (!has_tile && dag + (int)gridDim.x < B) attnwb_fill(attnwb_lds(smem, cur ^ 1), a, dag + gridDim.x, L.wave - NT, 8 - NT, D, L, wst_);
        WBSTAMP(6);
        __syncthreads();
        if (real && L.wave < 4) dvsw_coop_dw(aWo, abo, (const dvs_bf16*)l.Q, (const dvs_bf16*)l.DO, NT, L);
        __syncthreads();
        WBSTAMP(7);
        cur ^= 1;
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    float* red = (float*)smem;                        // [4 waves][64]
    if (L.wave < 4) {
        red[L.wave * 64 + L.lane] = 0.f;
        dvs_wave_sync();
        if (L.r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[L.wave * 64 + 16 * L.wave + 4 * L.g + reg] = abo[reg];
        }
        dvs_coop_flush<4>(nullptr, slab + a.o_out_w, aWo, L);      // rows 16w .. of dWo straight from the accumulators
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[w * 64 + threadIdx.x];
        slab[a.o_out_b + threadIdx.x] = s;
    }
}

void dvs_launch_attn_bwd_w(const AttnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = ATTNWB_FLOATS * 4;
    DVS_SET_LDS(k_attn_bwd_w, lds);
    DVS_LAUNCH(k_attn_bwd_w, dim3(grid), dim3(512), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Loss head backward, wide (autograd of pace.py:1880-1972 + the last decoder LayerNorm; k_loss_bwd restated).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_bwd_w(LossArgs a) {
    DVS_DYN_LDS(smem);
    const LossWLds l = lossw_lds(smem);
    lossw_stage(l, a);
    lossw_zero_parks(l);
    __syncthreads();
    const Lane L = dvs_lane();
    const int N = a.dims.N, C = a.dims.C, NT = a.dims.NT;
    const int tok0 = 16 * L.wave, Nl = dvs_rows_of(N, L.wave);
    const bool has_tile = L.wave < NT;
    const int tok = tok0 + L.r;
    constexpr int DLD = DVS_WTOK + 1;
    float* scr = l.scr + L.wave * DVS_SCR;
    const float b2 = l.b2[0];
    const float gr = a.gcoef[0];
    // the two 64 x 64 edge matrices cooperatively (dvsw_coop_dw): wave w accumulates rows 16w .. over the DAG's tiles — 16
    // accumulator registers per matrix instead of 64 (round 2: 512 registers + 246 spilled, 636 bytes of scratch per lane);
    // d edge0.bias rides along as the column sums of dV
    f4 dWn1[2][4], dWn2[3][2], aWa[4], aWb[4], abU = f4_zero(), abV = f4_zero(), dbn1[2], dbn2[3], dw2[4], dgam[4], dbet[4];
    float db2 = 0.f;
    dvs_bf16* const pU = (dvs_bf16*)l.pU;           // parked dU / dV tiles of the DAG, [hi | lo] pairs (4 blocks each: tile 3 zero)
    dvs_bf16* const pV = (dvs_bf16*)l.pV;
#pragma unroll
    for (int i = 0; i < 4; ++i) dw2[i] = dgam[i] = dbet[i] = aWa[i] = aWb[i] = f4_zero();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dbn1[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dWn1[i][j] = f4_zero();
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        dbn2[i] = f4_zero();
        dWn2[i][0] = dWn2[i][1] = f4_zero();
    }
    unsigned long long wst_ = 0;
#ifdef DVS_STAMPS
    wst_ = __builtin_amdgcn_s_memtime();
#endif
    for (int dag = blockIdx.x; dag < a.dims.B; dag += gridDim.x) {
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + dag;
        const size_t tile = (size_t)dag * NT + L.wave;
        f4 h[4], xhat[4], dh[4], w2v[4], dU[4], dV[4];
        float rstd = 1.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) w2v[t] = dvs_vecT(l.w2, t, L);      // (every wave: all four share the pair walks)
        if (has_tile) {
            f4 hN[4], U[4], V[4];
            dvs_load_x<true>(h, xhat, rstd, a.xin, a.ln, l.lg, l.lb, tile, Nl, L);
            dvs_t2n<4>(hN, h, scr, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) dh[t] = f4_zero();
            // ---- node head --------------------------------------------------------------------------------------
            {
                f4 t1p[2], t1[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) t1p[t] = dvs_vecT(l.bn1, t, L);
                dvs_mat_T<2, 4>(t1p, h, l.Wn1, DVS_LD, 0, L);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) t1[t][kk] = fmaxf(t1p[t][kk], 0.f);
                f4 lgt[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) lgt[ct] = *(const f4*)(l.bn2 + 16 * ct + 4 * L.g);
                dvs_mat_T<3, 2>(lgt, t1, l.Wn2, LOSSW_LDN2, 0, L);
                float mx = -3.0e38f;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) mx = (16 * ct + 4 * L.g + reg < C) ? fmaxf(mx, lgt[ct][reg]) : mx;
                mx = dvs_max_g(mx);
                f4 ex[3];
                float se = 0.f;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        ex[ct][reg] = (16 * ct + 4 * L.g + reg < C) ? __expf(lgt[ct][reg] - mx) : 0.f;
                        se += ex[ct][reg];
                    }
                se = dvs_sum_g(se);
                const float rse = dvs_rcp(se);
                const int target = rec->label[tok + 1 < DVS_WTOK ? tok + 1 : 0];
                const bool vt = tok < N - 1;
                f4 dlg[3];
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int c = 16 * ct + 4 * L.g + reg;
                        dlg[ct][reg] = (vt && c < C) ? gr * (ex[ct][reg] * rse - (c == target ? 1.f : 0.f)) : 0.f;
                    }
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) dbn2[ct] += dlg[ct];
                f4 dlgN[3], t1N[2];
                dvs_t2n<3>(dlgN, dlg, scr, L);
                dvs_t2n<2>(t1N, t1, scr, L);
                dvs_outer_acc<3, 2>(dWn2, dlgN, t1N);
                f4 dt1[2] = {f4_zero(), f4_zero()};
                dvs_mat_Tt<2, 3>(dt1, dlg, l.Wn2, LOSSW_LDN2, 0, L);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) dt1[t][kk] = t1p[t][kk] > 0.f ? dt1[t][kk] : 0.f;
                    dbn1[t] += dt1[t];
                }
                f4 dt1N[2];
                dvs_t2n<2>(dt1N, dt1, scr, L);
                dvs_outer_acc<2, 4>(dWn1, dt1N, hN);
                dvs_mat_Tt<4, 2>(dh, dt1, l.Wn1, DVS_LD, 0, L);
            }
            // ---- edge head: U, V of this tile; both parked for the other tiles ------------------------------------
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                U[t] = f4_zero();
                V[t] = dvs_vecT(l.be1, t, L);
                dU[t] = dV[t] = f4_zero();
            }
            dvs_mat_T<4, 4>(U, h, l.Wa, DVS_LD, 0, L);
            dvs_mat_T<4, 4>(V, h, l.Wb, DVS_LD, 0, L);
            dvs_park_T(l.V + tok0 * DVS_LD, V, L);
            dvs_park_T(l.U + tok0 * DVS_LD, U, L);
        }
        WCSTAMP(4);
        __syncthreads();
        // ---- the two pair walks, shared evenly by the FOUR waves --------------------------------------------------------------
        // A walk step is (token tile t, partner index): pass 1 — lane r = token i = 16 t + r walks j < i: d logit(i, j), dU_i, dw2,
        // db2 —, pass 2 — lane r = token j walks i > j: dV_j.  Tile t has min(N - 2, 16 t + 15) steps in pass 1 and
        // N - 1 - max(16 t + 1, 1) in pass 2: with one tile per wave the walks lasted as long as their longest tile (38 of 84 and
        // 38 of 66 steps at N = 40, the wave without a tile idle: 24 k + 21 k of the kernel's 69 k cycles per DAG,
        // `profiles/r03_wide_stamps_heads.txt`).  Now the steps of all tiles, in tile order, are cut into four equal ranges; a wave
        // takes U_t (pass 2: V_t) of the tiles its range touches from LDS, and its partial sums go to the tile's owner through a
        // slot (the per-wave transpose tiles and the parked dU / dV blocks are free during the walks), added in wave order.
        const int slots_per_wave = NT == 1 ? 1 : 2;
        auto slot_of = [&](int id) -> float* {
            return id < 4 ? l.scr + id * DVS_SCR : (id < 4 + NT ? l.pU + (id - 4) * DVS_SCR : l.pV + (id - 4 - NT) * DVS_SCR);
        };
        // steps of tile t in a pass and the first partner index
        auto steps1 = [&](int t) { const int e = N - 2 < 16 * t + 15 ? N - 2 : 16 * t + 15; return t < NT && e > 0 ? e : 0; };
        auto first2 = [&](int t) { return 16 * t + 1 > 1 ? 16 * t + 1 : 1; };
        auto steps2 = [&](int t) { const int c = N - 1 - first2(t); return t < NT && c > 0 ? c : 0; };
        // this wave's share [lo, hi) of tile t's steps when `total` steps in tile order are cut into 4 ranges of q
        auto share = [&](int w, int q, int off, int cnt, int& lo, int& hi) {
            const int g0 = w * q, g1 = g0 + q;
            lo = (g0 > off ? g0 : off) - off;
            hi = (g1 < off + cnt ? g1 : off + cnt) - off;
        };
        // owner side: sum of the partial tiles of tile `t`, in wave order
        auto gather = [&](int t, int q, bool pass2, f4 (&sum)[4]) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) sum[tt] = f4_zero();
            for (int w = 0; w < 4; ++w) {
                int off = 0, part = 0;
                for (int u = 0; u < DVS_WNT; ++u) {
                    const int cnt = pass2 ? steps2(u) : steps1(u);
                    int lo, hi;
                    share(w, q, off, cnt, lo, hi);
                    if (lo < hi) {
                        if (u == t) {
                            const f4* sl = (const f4*)slot_of(w * slots_per_wave + part) + L.lane;
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) sum[tt] += sl[tt * 64];
                        }
                        ++part;
                    }
                    off += cnt;
                }
            }
        };
        const int total1 = steps1(0) + steps1(1) + steps1(2), q1 = (total1 + 3) >> 2;
        const int total2 = steps2(0) + steps2(1) + steps2(2), q2 = (total2 + 3) >> 2;
        {
            // pass 1
            int off = 0, part = 0;
            for (int t = 0; t < DVS_WNT; ++t) {
                const int cnt = steps1(t);
                int lo, hi;
                share(L.wave, q1, off, cnt, lo, hi);
                off += cnt;
                if (lo >= hi) continue;
                const int ptok = 16 * t + L.r;
                f4 Ut[4];
                dvs_lds_T(Ut, l.U, 16 * t, L);
                const uint64_t par = rec->parents[ptok + 1 < DVS_WTOK ? ptok + 1 : 0];
                // written for instruction count, as k_loss_bwd's passes: f4 arithmetic, w2 factored out (dU = w2 * sU), the ReLU
                // derivative as a clamped multiply
                f4 sU[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
                for (int j = lo; j < hi; ++j) {
                    f4 pre[4], ev = f4_zero();
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const f4 x = Ut[tt] + *(const f4*)(l.V + j * DVS_LD + 16 * tt + 4 * L.g);
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) pre[tt][kk] = fmaxf(x[kk], 0.f);
                        ev += w2v[tt] * pre[tt];
                    }
                    const float logit = dvs_sum_g((ev[0] + ev[1]) + (ev[2] + ev[3])) + b2;
                    const bool pv = (ptok > j) && (ptok <= N - 2);
                    const float truth = (float)((par >> (j + 1)) & 1ull);
                    const float sg = dvs_rcp(1.0f + __expf(-logit));
                    const float dl = pv ? gr * (sg - truth) : 0.f;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        dw2[tt] += pre[tt] * dl;
                        f4 st;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) st[kk] = fminf(fmaxf(pre[tt][kk] * 1.0e30f, 0.f), 1.f);
                        sU[tt] += st * dl;
                    }
                    if (L.g == 0) {
                        db2 += dl;
                        l.dlm[ptok * DLD + j] = dl;
                    }
                }
                f4* sl = (f4*)slot_of(L.wave * slots_per_wave + part) + L.lane;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) sl[tt * 64] = sU[tt];
                ++part;
            }
        }
        WCSTAMP(5);
        __syncthreads();
        if (has_tile) {
            gather(L.wave, q1, false, dU);
#pragma unroll
            for (int t = 0; t < 4; ++t) dU[t] = w2v[t] * dU[t];
        }
        __syncthreads();                 // the slots are free again
        {
            // pass 2
            int off = 0, part = 0;
            for (int t = 0; t < DVS_WNT; ++t) {
                const int cnt = steps2(t);
                int lo, hi;
                share(L.wave, q2, off, cnt, lo, hi);
                off += cnt;
                if (lo >= hi) continue;
                const int ptok = 16 * t + L.r, i0 = first2(t);
                f4 Vt[4];
                dvs_lds_T(Vt, l.V, 16 * t, L);
                f4 sV[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
                for (int i = i0 + lo; i < i0 + hi; ++i) {
                    const float dl = (ptok < i) ? l.dlm[i * DLD + ptok] : 0.f;
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const f4 x = *(const f4*)(l.U + i * DVS_LD + 16 * tt + 4 * L.g) + Vt[tt];
                        f4 st;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) st[kk] = fminf(fmaxf(x[kk] * 1.0e30f, 0.f), 1.f);
                        sV[tt] += st * dl;
                    }
                }
                f4* sl = (f4*)slot_of(L.wave * slots_per_wave + part) + L.lane;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) sl[tt * 64] = sV[tt];
                ++part;
            }
        }
        WCSTAMP(6);
        __syncthreads();
        if (has_tile) gather(L.wave, q2, true, dV);
        __syncthreads();                 // every owner has its sums: the slots become the parked tiles
        if (has_tile) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dV[t] = w2v[t] * dV[t];
            dvs_park_bf(pU + L.wave * 2 * DVS_SCR, dU, L);
            dvs_park_bf(pV + L.wave * 2 * DVS_SCR, dV, L);
            dvs_park_bf((dvs_bf16*)scr, h, L);                        // the wave's transpose tile is free by now
        }
        __syncthreads();
        dvsw_coop_dw(aWa, abU, pU, (const dvs_bf16*)l.scr, NT, L);   // all four waves (wave 3 has no tile but owns rows 48..63)
        dvsw_coop_dw(aWb, abV, pV, (const dvs_bf16*)l.scr, NT, L);
        if (has_tile) {
            dvs_mat_Tt<4, 4>(dh, dU, l.Wa, DVS_LD, 0, L);
            dvs_mat_Tt<4, 4>(dh, dV, l.Wb, DVS_LD, 0, L);
            dvs_ln_bwd(dh, xhat, rstd, l.lg, dgam, dbet, L);
            dvs_store_tile(a.gout, tile, dh, L);
        }
        __syncthreads();
        WCSTAMP(7);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    {
        // the edge matrices: rows 16w .. straight from wave w's accumulators (add_edge.0.weight is [64][128] = [Wa | Wb])
        dvs_coop_flush<4>(nullptr, slab + a.o_edge0_w, aWa, L, false, false, 128);
        dvs_coop_flush<4>(nullptr, slab + a.o_edge0_w + 64, aWb, L, false, false, 128);
        if (L.r == 0) {          // d edge0.bias: wave w holds features 16w + 4g + reg (every column r the same)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) slab[a.o_edge0_b + 16 * L.wave + 4 * L.g + reg] = abV[reg];
        }
        float* rN1 = (float*)smem;                      // 4 * 2048
        float* rN2 = rN1 + 4 * 2048;                    // 4 * 1536
        float* rv1 = rN2 + 4 * 1536;                    // dbn1: 4 * 32
        float* rv2 = rv1 + 4 * 32;                      // dbn2: 4 * 48
        float* rv3 = rv2 + 4 * 48;                      // dbe1, dw2, dgam, dbet: 4 * 64 each
        float* rb2 = rv3 + 4 * DVS_RED_VEC;             // 16
        float* es = rb2 + 16 + L.wave * DVS_SCR;
        dvs_stage_dw<2, 4>(rN1, dWn1, L);
        dvs_stage_dw<3, 2>(rN2, dWn2, L);
        dvs_stage_vec<2>(rv1, dbn1, es, L);
        dvs_stage_vec<3>(rv2, dbn2, es, L);
        dvs_stage_vec<4>(rv3 + DVS_RED_VEC, dw2, es, L);
        dvs_stage_vec<4>(rv3 + 2 * DVS_RED_VEC, dgam, es, L);
        dvs_stage_vec<4>(rv3 + 3 * DVS_RED_VEC, dbet, es, L);
        const float sb2 = dvs_sum_wave(L.g == 0 ? db2 : 0.f);
        if (L.lane == 0) rb2[L.wave] = sb2;
        __syncthreads();
        dvs_flush_dw<2, 4>(rN1, slab + a.o_node0_w, L);
        dvs_flush_dw<3, 2>(rN2, slab + a.o_node2_w, L, C, 32);
        dvs_flush_vec<2>(rv1, slab + a.o_node0_b, L);
        dvs_flush_vec<3>(rv2, slab + a.o_node2_b, L, C);
        dvs_flush_vec<4>(rv3 + DVS_RED_VEC, slab + a.o_edge2_w, L);
        dvs_flush_vec<4>(rv3 + 2 * DVS_RED_VEC, slab + a.o_ln_g, L);
        dvs_flush_vec<4>(rv3 + 3 * DVS_RED_VEC, slab + a.o_ln_b, L);
        if (threadIdx.x == 0) {
            float s = rb2[0];
            for (int w = 1; w < L.nwaves; ++w) s += rb2[w];
            slab[a.o_edge2_b] = s;
        }
    }
}

void dvs_launch_loss_bwd_w(const LossArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = dvs_lossw_lds_floats() * 4;       // (the epilogue's staging areas are smaller than the DAG loop's layout)
    DVS_SET_LDS(k_loss_bwd_w, lds);
    DVS_LAUNCH(k_loss_bwd_w, dim3(grid), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Embedding backward, wide (autograd of pace.py:201-221, 1181-1184).  The forward's row gathers become row scatters;
// here they are OWNER-COMPUTES: thread (row p, feature f) of W1 keeps its gradient element in a register and, per DAG,
// pulls the rows of the parked d e1 tile that the DAG maps to p (tokens at position p; tokens with a parent at
// position p) — no atomics, fixed summation order.  Same for the label table (class c <- tokens labelled c).
// Up to two gradient sources per DAG (encoder- and decoder-side embeddings share weights and the scatter pattern, so
// their d e1 are added before parking).
// ---------------------------------------------------------------------------------------------------------
constexpr int EMBW_DLE_LD = 36;
constexpr int EMBW_W1_PER_THREAD = 2 * DVS_WTOK * 64 / 256;      // 24
constexpr int EMBW_LAB_PER_THREAD = 32 * DVS_WTOK / 256;         // 6
__global__ __launch_bounds__(256) void k_embed_bwd_w(EmbedArgs a, const float* gout2, int site2) {
    DVS_DYN_LDS(smem);
    const EmbWLds l = embw_lds(smem);
    float* DE1 = (float*)smem + EMBW_FLOATS;                     // [48][LD]
    float* DLE = DE1 + DVS_WSCR;                                 // [48][36]
    float* scr0 = DLE + DVS_WTOK * EMBW_DLE_LD;                  // 4 per-wave transpose tiles
    uint64_t* posmask = (uint64_t*)(scr0 + 4 * DVS_SCR);         // [48] tokens at position p
    uint64_t* parmask = posmask + DVS_WTOK;                      // [48] tokens with a parent at position p
    uint64_t* labmask = parmask + DVS_WTOK;                      // [48] tokens labelled c
    uint64_t* parents = labmask + DVS_WTOK;                      // [48]
    embw_stage(l, a);
    for (int i = threadIdx.x; i < DVS_WSCR + DVS_WTOK * EMBW_DLE_LD; i += blockDim.x) DE1[i] = 0.f;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N, C = a.dims.C, NT = a.dims.NT;
    const int tok0 = 16 * L.wave, Nl = dvs_rows_of(N, L.wave);
    const bool has_tile = L.wave < NT;
    float* scr = scr0 + L.wave * DVS_SCR;
    // d W1 ([2 N][64]: rows p < N <- tokens at position p, rows N + p <- tokens with parents at position p, once per parent) and
    // the label table's gradient ([C classes][32]) as products with 0 / count selector matrices on the matrix pipe, as the
    // one-tile kernel does: wave w owns feature tile w of d W1 (6 row tiles of 16 positions) and label tiles w, w + 4 of the six
    // (3 class tiles x 2 feature tiles).  A operand element [row p][token i] = popcount(mask_p & sel_i): sel_i = bit i for a
    // position row, the parent row of token i for a parent row.  (Round 3's first version was owner-computes: thread (row, feature)
    // walked the set bits of its rows' masks, a chain of dependent LDS reads: 16 k of the kernel's 36 k cycles per DAG.)
    f4 aW1[6], alab[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) aW1[k] = f4_zero();
    alab[0] = alab[1] = f4_zero();
    f4 dW2[4][2], dlabb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) dW2[i][0] = dW2[i][1] = f4_zero();
    dlabb[0] = dlabb[1] = f4_zero();
    unsigned long long wst_ = 0;
#ifdef DVS_STAMPS
    wst_ = __builtin_amdgcn_s_memtime();
#endif
    for (int dag = blockIdx.x; dag < a.dims.B; dag += gridDim.x) {
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + dag;
        const size_t tile = (size_t)dag * NT + L.wave;
        const uint32_t gdag = a.dims.dag_offset + dag;
        // the DAG's masks, one ballot each over lane = token (wave w: p = w, w + 4, ..): round 3's first version had thread p walk
        // the record's bytes in global memory, 2 N dependent loads per DAG ahead of everything else
        {
            const int ti = L.lane < N ? L.lane : 0;
            const int mypos = L.lane < N ? (int)rec->pos[ti] : -1, mylab = L.lane < N ? (int)rec->label[ti] : -1;
            const uint64_t mypar = L.lane < N ? rec->parents[ti] : 0ull;
            if (L.wave == 0 && L.lane < DVS_WTOK) parents[L.lane] = mypar;
            for (int p = L.wave; p < DVS_WTOK; p += 4) {
                const uint64_t pm = __ballot(mypos == p), lm = __ballot(mylab == p);
                if (L.lane == 0) {
                    posmask[p] = pm;
                    labmask[p] = lm;
                }
            }
        }
        if (has_tile) {
            const bool valid = L.r < Nl;
            const int label = rec->label[valid ? tok0 + L.r : 0];
            f4 e1[4];
            embw_hidden(e1, l.W1, rec, l.posl + 16 * L.wave, N, tok0, Nl, L);
            f4 de1s[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()}, dles[2] = {f4_zero(), f4_zero()};
            for (int src = 0; src < 2; ++src) {
                const float* gsrc = src == 0 ? a.gout : gout2;
                if (!gsrc) continue;
                const int site = src == 0 ? a.site : site2;
                f4 gx[4];
                dvs_load_grad(gx, gsrc, tile, Nl, L);
                f4 dle[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int f = 16 * t + 4 * L.g + kk;
                        const float pre = l.labw[f * EMBW_LABLD + label] + l.labb[f];
                        dle[t][kk] = (valid && pre > 0.f) ? gx[t][kk] : 0.f;
                    }
                dlabb[0] += dle[0];
                dlabb[1] += dle[1];
                dles[0] += dle[0];
                dles[1] += dle[1];
                const uint32_t k1 = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site, gdag);
                f4 m1[4] = {f4_splat(1.f), f4_splat(1.f), f4_splat(1.f), f4_splat(1.f)};      // k_embed_bwd: one draw of the mask
                dvs_dropout_tile(m1, k1, D, L, tok0);
                f4 e1d[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) e1d[t] = e1[t] * m1[t];
                f4 de2[2] = {gx[2], gx[3]};
                dvs_dropout_tile<2>(de2, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site + 1, gdag), D, L, tok0);
                f4 e1dN[4], de2N[2];
                dvs_t2n<4>(e1dN, e1d, scr, L);
                dvs_t2n<2>(de2N, de2, scr, L);
                dvs_outer_acc<4, 2>(dW2, e1dN, de2N);
                f4 de1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
                dvs_mat_T<4, 2>(de1, de2, l.W2, EMB_LDW2, 0, L);
#pragma unroll
                for (int t = 0; t < 4; ++t) de1[t] *= m1[t];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) de1s[t][kk] += e1[t][kk] > 0.f ? de1[t][kk] : 0.f;
            }
            dvs_park_T(DE1 + tok0 * DVS_LD, de1s, L);
#pragma unroll
            for (int t = 0; t < 2; ++t) *(f4*)(DLE + (tok0 + L.r) * EMBW_DLE_LD + 16 * t + 4 * L.g) = dles[t];
        }
        WCSTAMP(0);
        __syncthreads();
        WCSTAMP(1);
        // ---- scatters as selector-matrix products --------------------------------------------------------------------
        {
            uint64_t mp[6];              // this lane's row p = 16 pt + r of d W1: the tokens at its position (0: row beyond 2 N)
            bool par_row[6];
#pragma unroll
            for (int pt = 0; pt < 6; ++pt) {
                const int prow = 16 * pt + L.r;
                par_row[pt] = prow >= N;
                mp[pt] = prow < N ? posmask[prow] : (prow < 2 * N ? posmask[prow - N] : 0ull);
            }
            const int lt0 = L.wave, lt1 = L.wave + 4;                  // label tiles: class tile lt % 3, feature tile lt / 3
            const uint64_t ml0 = labmask[16 * (lt0 % 3) + L.r], ml1 = lt1 < 6 ? labmask[16 * (lt1 % 3) + L.r] : 0ull;
            const int npt = (2 * N + 15) >> 4;
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int i = 16 * kt + 4 * kk + L.g;             // token of this contraction step
                    const float b = DE1[i * DVS_LD + 16 * L.wave + L.r];
                    const uint64_t bit = 1ull << i, pari = parents[i];
#pragma unroll
                    for (int pt = 0; pt < 6; ++pt) {
                        if (pt < npt) {
                            const uint64_t x = mp[pt] & (par_row[pt] ? pari : bit);
                            aW1[pt] = dvs_mfma((float)__popcll(x), b, aW1[pt]);
                        }
                    }
                    const float bl0 = DLE[i * EMBW_DLE_LD + 16 * (lt0 / 3) + L.r];
                    alab[0] = dvs_mfma((ml0 & bit) ? 1.f : 0.f, bl0, alab[0]);
                    if (lt1 < 6) {
                        const float bl1 = DLE[i * EMBW_DLE_LD + 16 * (lt1 / 3) + L.r];
                        alab[1] = dvs_mfma((ml1 & bit) ? 1.f : 0.f, bl1, alab[1]);
                    }
                }
            }
        }
        WCSTAMP(2);
        __syncthreads();
        WCSTAMP(3);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    // D layout: register reg of lane (r, g) is element [row 4g + reg][column r] of the tile
#pragma unroll
    for (int pt = 0; pt < 6; ++pt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int prow = 16 * pt + 4 * L.g + reg;
            if (prow < 2 * N) slab[a.oW1 + (size_t)prow * 64 + 16 * L.wave + L.r] = aW1[pt][reg];
        }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int lt = L.wave + 4 * k;
        if (lt < 6) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int c = 16 * (lt % 3) + 4 * L.g + reg, f = 16 * (lt / 3) + L.r;
                if (c < C) slab[a.olab_w + (size_t)f * C + c] = alab[k][reg];
            }
        }
    }
    float* rW2 = (float*)smem;                 // 4 * 2048
    float* rv = rW2 + 4 * 2048;                // 4 * 32
    dvs_stage_dw<4, 2>(rW2, dW2, L);
    dvs_stage_vec<2>(rv, dlabb, rv + 4 * 32 + L.wave * DVS_SCR, L);
    __syncthreads();
    dvs_flush_dw<4, 2>(rW2, slab + a.oW2, L);
    dvs_flush_vec<2>(rv, slab + a.olab_b, L);
}

void dvs_launch_embed_bwd_w(const EmbedArgs& a, const float* gout2, int site2, int grid, dvs_stream_t st) {
    size_t lds = (EMBW_FLOATS + DVS_WSCR + DVS_WTOK * EMBW_DLE_LD + 4 * DVS_SCR + 2 * 4 * DVS_WTOK) * 4;
    const size_t red = (4 * 2048 + 4 * 32 + 4 * (size_t)DVS_SCR) * 4;
    if (lds < red) lds = red;
    DVS_SET_LDS(k_embed_bwd_w, lds);
    DVS_LAUNCH(k_embed_bwd_w, dim3(grid), dim3(256), lds, st, a, gout2, site2);
}
