// Host/device shared plain structs: flat parameter layout, workspace layout, kernel argument blocks.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/dvs.h"

constexpr int DVS_LAYERS = 3;
constexpr int DVS_HEADS = 8;
constexpr int DVS_LATENT = 32;
constexpr int DVS_FCH = 32;   // fc_hidden
constexpr int DVS_EMB = 32;   // vertices_embedding_size
constexpr int DVS_MAXTOK = 16;   // tokens of one tile (narrow path: whole DAG)
constexpr int DVS_WTOK = 48;     // widest DAG of the tiled ("wide") path: 3 tiles of 16 tokens (alarm n = 37 -> N = 40)
constexpr int DVS_FC_PARTS = 16; // batch parts of the fc1/fc2/fc3 weight-gradient GEMMs
constexpr int DVS_NSLOTS = 17;   // saved activation slots: 0 enc-embed, 1..6 enc sublayers, 7 dec-embed, 8..16 dec sublayers

struct DvsAttnP { int64_t in_w, in_b, out_w, out_b; };
struct DvsFfnP { int64_t l1_w, l1_b, l2_w, l2_b; };
struct DvsNormP { int64_t w, b; };

struct DvsLayout {   // float offsets into the flat parameter buffer (state-dict registration order)
    int64_t W1, W2, lab_w, lab_b;
    struct { DvsAttnP sa; DvsFfnP ff; DvsNormP n1, n2; } enc[DVS_LAYERS];
    int64_t fc1_w, fc1_b, fc2_w, fc2_b;
    struct { DvsAttnP sa, ca; DvsFfnP ff; DvsNormP n1, n2, n3; } dec[DVS_LAYERS];
    int64_t node0_w, node0_b, node2_w, node2_b, edge0_w, edge0_b, edge2_w, edge2_b, fc3_w, fc3_b;
    int64_t total;
};

DvsLayout dvs_make_layout(int N, int C, dvs_param_entry* table /* may be null */, int cap, int* count);

struct DvsDropH {   // host mirror of DvsDrop (dvs_device.h)
    uint32_t thr16;
    float scale;
    int on;
};

struct DvsDims {
    int B, N, C, training;
    int NT;                     // 16-token tiles per DAG: 1 (N <= 16, one wave owns a DAG) or ceil(N/16) (wide path)
    DvsDropH drop;
    uint32_t seed_lo, seed_hi, dag_offset;
    float beta, eps_scale;
    int debug;                  // DVS_DEBUG_SKIP timing experiments (0 in production)
};

// Workspace layout (float offsets unless noted); every region is 256-byte aligned.
struct DvsWorkspace {
    size_t act[DVS_NSLOTS];     // [B*NT][1024] frag-order pre-LayerNorm sums (slot 0/7: embedding outputs)
    size_t stats[DVS_NSLOTS];   // [B][32]: mean[16], rstd[16] of the slot's LayerNorm
    size_t enc_out;             // [B][1024] LayerNorm'ed encoder output (input of fc1/fc2)
    size_t mu, logvar, z, epsv; // [B][32]
    size_t mem;                 // [B][1024] fc3 output = decoder memory
    size_t dag_loss;            // [B][2]: per-DAG node+edge NLL, per-DAG KL
    size_t gA, gB;              // [B][1024] gradient ping/pong (d pre)
    size_t gq, gk, gv;          // [B][1024] attention projection gradients
    size_t gmem;                // [B][1024] d memory (summed over the 3 decoder layers)
    size_t genc;                // [B][1024] d enc_out
    size_t gz;                  // [B][64]: d mu | d logvar
    size_t slabs;               // [nslab][P] per-workgroup partial parameter gradients
    size_t fcpart;              // [DVS_FC_PARTS][P] partial fc1/fc2/fc3 gradients
    size_t wimg;                // per-step weight images (dvs_wimg.h), bf16, written by the forward entry points
    size_t limg;                // per-step latent weight images (dvs_wimg.h: DvsLatImg), fp32, contraction index in frag order
    size_t qkv[9];              // wide path only: q, k, v of the 9 attention sublayers as the forward parked them, [B NT][12 output tiles][256]
    size_t total_floats;
    int nslab;
};

DvsWorkspace dvs_make_workspace(int B, int NT, int64_t P, int nslab, bool wide);
