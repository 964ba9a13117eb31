// Per-step weight images (the LDS images of dvs_bf16.h, built ONCE per step in global memory).
//
// Every sublayer kernel used to convert its weights on entry: fp32 -> bf16 parts, column permutation, element-wise
// 2-byte LDS stores — 10-15 us of every launch, with only 16 DAGs per workgroup to amortise it (a batch of 8 DAGs ran
// k_attn_fwd in 24 us against 34 us for 4096).  Parameters change once per step, so k_prepare_images writes, for every
// attention and FFN sublayer, the images exactly as the kernels want them in LDS; staging is then a 16-byte-per-lane
// copy (dvs_copy_image).  Blocks live in the caller's workspace (DvsWorkspace::wimg), bf16 units below.
//
//   attention block : Win  x6 [3][192][LDB]  rows in slot order         (k_attn_fwd; parts 0,1 = the x3 pair of k_attn_bwd)
//                     Wout x6 [3][ 64][LDB]  columns in slot order      (k_attn_fwd)
//                     WoutT x3 [2][64][LDB]  image rows in slot order   (k_attn_bwd: dO^T = Wo^T dy^T)
//                     WinB  x3 [2][192][LDB] = parts 0, 1 of Win again   (k_attn_bwd: q, k, v recompute)
//                     WinT  x3 [3 proj][2][64][LDB]                     (k_proj_bwd: dX^T = W_p^T dY_p^T)
//   FFN block       : W2T x3 [2][64][LDB], W1T x3 [2][64][LDB]          (k_ffn_bwd)
//                     W1 x6 [3][64][LDB], W2 x6 [3][64][LDB]            (k_ffn_fwd; W1 also k_ffn_bwd's hidden recompute)
//   loss block      : Wa, Wb x6 [3][64][LDB] each (the two 64-column halves of add_edge.0.weight: k_loss_fwd and
//                     k_loss_bwd's recompute — the sign of Wa h_i + Wb h_j + b is a ReLU mask, so both use the same
//                     bf16x6 sequence), WaT, WbT x3 [2][64][LDB] (k_loss_bwd: d h)
#pragma once
#include "dvs_bf16.h"

constexpr size_t DVS_IMG64 = 64 * (size_t)DVS_LDB;            // one 64-row image
// Every phase stages ONE contiguous block (dvs_stage.h): [Win, Wout] the attention forward, [WoutT, WinB] the attention-core
// backward (WinB: a second copy of parts hi, mid of Win — 55 KB more per attention and step for k_prepare_images, against a
// two-segment copy in every backward tail), WinT pairs the projection backward; [W2T, W1T, W1] the FFN backward, [W1, W2] the
// FFN forward.
struct DvsAttnImg {
    static constexpr size_t Win = 0;
    static constexpr size_t Wout = 3 * 3 * DVS_IMG64;
    static constexpr size_t WoutT = Wout + 3 * DVS_IMG64;
    static constexpr size_t WinB = WoutT + 2 * DVS_IMG64;
    static constexpr size_t WinT = WinB + 2 * 3 * DVS_IMG64;
    static constexpr size_t SIZE = WinT + 3 * 2 * DVS_IMG64;
};
struct DvsFfnImg {
    static constexpr size_t W2T = 0;
    static constexpr size_t W1T = 2 * DVS_IMG64;
    static constexpr size_t W1 = 4 * DVS_IMG64;
    static constexpr size_t W2 = 7 * DVS_IMG64;
    static constexpr size_t SIZE = 10 * DVS_IMG64;
};
struct DvsLossImg {
    static constexpr size_t Wa = 0;
    static constexpr size_t Wb = 3 * DVS_IMG64;
    static constexpr size_t WaT = 6 * DVS_IMG64;
    static constexpr size_t WbT = 8 * DVS_IMG64;
    // fp32 "head block" behind the images, exactly as k_loss_fwd / k_loss_bwd keep it in LDS (dvs_loss.h), so that the whole loss
    // block is ONE verbatim copy (dvs_stage.h) instead of ten staging loops with a memory round trip each:
    //   Wn1 [32][DVS_LD] add_node.0.weight | Wn2 [16][36] add_node.2.weight (rows >= C zero) | bn1 [32] | bn2 [16] | be1 [64]
    //   add_edge.0.bias | w2 [64] add_edge.2.weight | b2 [16] (first: add_edge.2.bias) | lg [64], lb [64] last decoder LayerNorm
    static constexpr size_t Head = 10 * DVS_IMG64;                 // bf16 units, like the rest
    static constexpr int HEAD_FLOATS = 32 * 68 + 16 * 36 + 32 + 16 + 64 + 64 + 16 + 64 + 64;      // 3072 = 12 wave chunks
    static constexpr size_t SIZE = Head + 2 * (size_t)HEAD_FLOATS;
};
static_assert(DvsLossImg::HEAD_FLOATS % 256 == 0 && DVS_LD == 68, "whole 1 KB chunks (dvs_stage.h)");
// Embedding block (k_embed_fwd / k_embed_bwd, one-tile path), fp32, in the kernels' LDS layout:
//   W1 [32][DVS_LD] (rows >= 2 N zero) | W2 [64][36] | lab_w [32][16] (classes >= C zero) | lab_b [32] | pad to whole 1 KB chunks
struct DvsEmbImg {
    static constexpr int FLOATS = 5120;                            // 2176 + 2304 + 512 + 32 = 5024, padded: 20 wave chunks
};
struct DvsLossHeadArgs {
    const float *node0_w, *node0_b, *node2_w, *node2_b, *edge0_b, *edge2_w, *edge2_b, *ln_g, *ln_b;
    float* dst;                  // null: no head block (wide path)
    int C;
    const float *W1, *W2, *lab_w, *lab_b;                          // embedding block (dst_emb, same launch)
    float* dst_emb;
    int N;
};
// blocks of one step: encoder layer i -> attention block 3i ... see dvs_api.hip (img_enc_attn etc.); the loss block is last
constexpr int DVS_N_ATTN_BLOCKS = 9, DVS_N_FFN_BLOCKS = 6;
constexpr size_t DVS_WIMG_LOSS = DVS_N_ATTN_BLOCKS * DvsAttnImg::SIZE + DVS_N_FFN_BLOCKS * DvsFfnImg::SIZE;
constexpr size_t DVS_WIMG_EMB = DVS_WIMG_LOSS + DvsLossImg::SIZE;      // embedding block (fp32, DvsEmbImg)
constexpr size_t DVS_WIMG_BF16 = DVS_WIMG_EMB + 2 * (size_t)DvsEmbImg::FLOATS;

// Latent block (fc1 / fc2 / fc3; k_latent_fwd, k_latent_bwd): fp32 images whose contraction / row index is the FRAG-ORDER
// position k' of the DAG's activation tiles (k' -> token 16 tile + r', feature 16 t + 4 g' + kk; zero where the token is beyond
// N), in both orientations, so that an MFMA operand is ONE 16-byte load per lane and a wave's loads are contiguous rows.  The
// kernels used to read the parameters in place: 64 scattered 16-byte (or 4-byte) pieces per wave load, every 128-byte line
// fetched by four different waves of a workgroup — 250 MB of L2 reads per launch for 0.37 MB of weights.
struct DvsLatImg {
    // K = 1024 * NT.  A: [64 o][LD] rows 0-31 fc1, 32-63 fc2;  AT: [K][64];  W3: [K][32] fc3 rows;  W3T: [32][LD];  B3: [K] fc3 bias.
    // LD = K + 64: with a power-of-two row pitch (4 KB at NT = 1) the 16 rows a wave load touches sit in ONE L2 channel and
    // one L1 set — measured on the chained latent phase: 256 KB of weights per workgroup took 12 k cycles (21 B/clk).
    static constexpr size_t LD(int NT) { return (size_t)1024 * NT + 64; }
    static constexpr size_t A(int) { return 0; }
    static constexpr size_t AT(int NT) { return 64 * LD(NT); }
    static constexpr size_t W3(int NT) { return AT(NT) + (size_t)64 * 1024 * NT; }
    static constexpr size_t W3T(int NT) { return W3(NT) + (size_t)32 * 1024 * NT; }
    static constexpr size_t B3(int NT) { return W3T(NT) + 32 * LD(NT); }
    static constexpr size_t floats(int NT) { return B3(NT) + (size_t)1024 * NT; }
};
struct DvsLatImgArgs {
    const float *fc1_w, *fc2_w, *fc3_w, *fc3_b;      // parameter order: [32][N*64], [32][N*64], [N*64][32], [N*64]
    float* img;
    int N, NT;
};

struct DvsImgJob {
    int64_t src;                 // float offset of the matrix in the flat parameter buffer
    int64_t dst;                 // bf16 offset of the image (first part) in the image buffer
    int32_t rows;                // source rows (64 columns)
    int32_t flags;               // bit 0: transposed x3 (else x6 rows); bit 1: rperm; bit 2: cperm; bit 3: source rows are 128 floats apart; bit 4: x6 rows, parts 0 and 1 only
};
constexpr int DVS_MAX_IMG_JOBS = 96;
struct DvsImgJobs {
    DvsImgJob job[DVS_MAX_IMG_JOBS];
    int count;
};
void dvs_launch_prepare_images(const DvsImgJobs& jobs, const float* params, dvs_bf16* wimg, const DvsLatImgArgs& lat,
                               const DvsLossHeadArgs& head, dvs_stream_t st);

// LDS <- global image copy, 16 bytes per lane (n = bf16 count, a multiple of 8; both 16-byte aligned).  Loads are issued
// in batches of 8 per thread before the first store: a load -> store loop pays one full L2 round trip per iteration
// (measured: 15.4 k cycles for 64 KB on 512 threads, 20 % of k_ffn_bwd).
__device__ __forceinline__ void dvs_copy_image(dvs_bf16* dst, const dvs_bf16* __restrict__ src, int n) {
    const f4* s = (const f4*)src;
    f4* d = (f4*)dst;
    const int n16 = n >> 3, step = blockDim.x;
    // every workgroup of the launch copies the SAME image at the same time: start each one at a different offset so that
    // they do not all queue on the same L2 lines
    const int rot = (int)(((unsigned)dvs_bid() * 2654435761u) % (unsigned)n16);
    int i = dvs_tid();
    for (; i + 7 * step < n16; i += 8 * step) {
        f4 v[8];
        int j[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            j[u] = i + u * step + rot;
            j[u] = j[u] >= n16 ? j[u] - n16 : j[u];
            v[u] = s[j[u]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) d[j[u]] = v[u];
    }
    // the tail (up to 7 chunks per thread) as one predicated batch too: a load -> store tail loop pays a full L2 round trip
    // per iteration — 5-6 of them for the 108 KB attention-forward block on 512 threads
    if (i < n16) {
        f4 v[7];
        int j[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int k = i + u * step;
            j[u] = k + rot;
            j[u] = j[u] >= n16 ? j[u] - n16 : j[u];
            if (k < n16) v[u] = s[j[u]];
        }
#pragma unroll
        for (int u = 0; u < 7; ++u)
            if (i + u * step < n16) d[j[u]] = v[u];
    }
}
