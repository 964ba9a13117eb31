// LDS layout of the loss-head kernels (k_loss_fwd, k_loss_bwd; one-tile path) = the per-step loss block of dvs_wimg.h, copied
// verbatim by one staging plan (dvs_stage.h), then the per-wave scratch tiles.
#pragma once
#include "dvs_kernels.h"
#include "dvs_wimg.h"
#include "dvs_stage.h"

constexpr int LOSS_LDN2 = 36;
struct LossLds {
    dvs_bf16 *Wa, *Wb;           // bf16x6 triples of the two halves of add_edge.0.weight (U, V; k_loss_bwd recomputes them alike)
    dvs_bf16 *WaT, *WbT;         // bf16x3 pairs of the transposes (k_loss_bwd: d h; the forward carries them along unused)
    float *Wn1, *Wn2, *bn1, *bn2, *be1, *w2, *b2, *lg, *lb, *scr;
};
DVS_HD inline LossLds loss_lds(char* smem) {
    LossLds l;
    l.Wa = (dvs_bf16*)smem;
    l.Wb = l.Wa + 3 * DVS_IMG64;
    l.WaT = l.Wb + 3 * DVS_IMG64;
    l.WbT = l.WaT + 2 * DVS_IMG64;
    l.Wn1 = (float*)(l.WbT + 2 * DVS_IMG64);          // head block (DvsLossImg::Head), same order
    l.Wn2 = l.Wn1 + 32 * DVS_LD;
    l.bn1 = l.Wn2 + 16 * LOSS_LDN2;
    l.bn2 = l.bn1 + 32;
    l.be1 = l.bn2 + 16;
    l.w2 = l.be1 + 64;
    l.b2 = l.w2 + 64;
    l.lg = l.b2 + 16;
    l.lb = l.lg + 64;
    l.scr = l.lb + 64;
    return l;
}
inline size_t loss_lds_bytes(int nwaves, int tiles_per_wave) {
    return DvsLossImg::SIZE * sizeof(dvs_bf16) + (size_t)nwaves * tiles_per_wave * DVS_SCR * sizeof(float);
}
inline void loss_plan(DvsStagePlan& p, const LossArgs& a) {
    dvs_plan_clear(p);
    const LossLds l = loss_lds(DVS_FAKE_LDS);
    dvs_plan_seg(p, DVS_FAKE_LDS, l.Wa, (const dvs_bf16*)a.wimg + DvsLossImg::Wa, (int)DvsLossImg::SIZE);
    dvs_plan_seal(p);
}
constexpr int LOSS_CHUNKS = (int)(DvsLossImg::SIZE * sizeof(dvs_bf16) / 1024);      // 102 wave chunks
