// Declarations shared by k_decode.hip and the C-ABI layer (dvs_api.hip): batched generation.
#pragma once
#include "dvs_wide.h"

constexpr int DEC_OUT = 1;      // graph_label_output (pace.py:1154); the label of padding tokens (1543)

struct DvsDecodeState {          // one per DAG (see include/dvs.h: dvs_decode_state)
    uint64_t parents[DVS_WTOK];  // bit j: edge j -> i
    uint8_t label[DVS_WTOK];
    int32_t nv;                  // vertices grown so far
    int32_t finished;            // sampled `output`: stopped growing
};
static_assert(sizeof(DvsDecodeState) == DVS_DECODE_STATE_BYTES, "dvs.h: DVS_DECODE_STATE_BYTES");

struct DecodeArgs {
    DvsDims dims;
    int wide;                    // record type
    int idx;                     // step: the vertex being added (2 .. N-1)
    void* rec;
    DvsDecodeState* state;
    const float* xin;            // last decoder sublayer's pre-LayerNorm sum (frag tiles)
    DvsLN ln;                    // decoder.layers.2.norm3
    const float *node0_w, *node0_b, *node2_w, *node2_b, *edge0_w, *edge0_b, *edge2_w, *edge2_b;
    const float* uniforms;       // optional [B][N][N]
};

void dvs_launch_decode_init(const DecodeArgs& a, dvs_stream_t st);
void dvs_launch_decode_step(const DecodeArgs& a, int grid, dvs_stream_t st);
void dvs_launch_decode_memory(const DvsDims& d, const float* z, const float* w, const float* b, float* mem, dvs_stream_t st);
