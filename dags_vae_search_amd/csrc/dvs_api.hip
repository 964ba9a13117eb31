// C ABI (include/dvs.h): parameter/workspace layout and the launch sequences of the PACE-VAE step.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dvs_kernels.h"
#include "dvs_backward.h"
#include "dvs_wide.h"
#include "dvs_wimg.h"

static thread_local char g_err[256] = "";

// ---- optional per-kernel timing (HIP events on the launch stream) -------------------------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace {
struct ProfRec {
    const char* name;
#ifndef DVS_EMU
    hipEvent_t a, b;
#endif
};
// The one piece of process-global mutable state of the library: begin/end pairs of concurrent callers may interleave
// (the record then pairs the wrong events), but the container itself is never corrupted.
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
thread_local int t_prof_slot = -1;      // index of this thread's open record
}  // namespace

void dvs_prof_begin(const char* name, dvs_stream_t st) {
    t_prof_slot = -1;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name;
#ifndef DVS_EMU
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) {
        (void)hipEventDestroy(r.a);
        return;
    }
    (void)hipEventRecord(r.a, st);
#endif
    t_prof_slot = (int)g_prof.size();
    g_prof.push_back(r);
}
void dvs_prof_end(dvs_stream_t st) {
    if (t_prof_slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
#ifndef DVS_EMU
    if (t_prof_slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[t_prof_slot].b, st);
#endif
    t_prof_slot = -1;
}
extern "C" void dvs_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = on != 0;
}
// Waits for the recorded events, then writes up to `cap` rows (name, launches, total milliseconds); returns the
// number of distinct kernels and clears the record.
extern "C" int dvs_profile_collect(char* names, int name_stride, int* counts, float* total_ms, int cap) {
    std::map<std::string, std::pair<int, float>> agg;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (auto& r : g_prof) {
        float ms = 0.f;
#ifndef DVS_EMU
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
#endif
        auto& e = agg[r.name];
        e.first += 1;
        e.second += ms;
    }
    g_prof.clear();
    int i = 0;
    for (auto& kv : agg) {
        if (i >= cap) break;
        snprintf(names + (size_t)i * name_stride, name_stride, "%s", kv.first.c_str());
        counts[i] = kv.second.first;
        total_ms[i] = kv.second.second;
        ++i;
    }
    return (int)agg.size();
}

static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

// ---- HIP runtime failures inside an entry point (DVS_LAUNCH / DVS_SET_LDS, dvs_kernels.h) ---------------------------
static thread_local int t_hip_err = 0;
static thread_local char t_hip_msg[200] = "";
void dvs_note_hip_error(const char* what, int hip_error, const char* hip_message) {
    if (t_hip_err != 0) return;             // keep the first failure of the call
    t_hip_err = hip_error ? hip_error : -1;
    snprintf(t_hip_msg, sizeof(t_hip_msg), "%s: HIP error %d (%s)", what, hip_error, hip_message ? hip_message : "?");
}
static void call_begin() {
    t_hip_err = 0;
    (void)hipGetLastError();                // errors that are not ours must not be attributed to our launches
}
static int call_end(const char* fn) {
    if (t_hip_err == 0) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", fn, t_hip_msg);
    t_hip_err = 0;
    return 20;
}
#ifndef DVS_EMU
#define DVS_HIP_CALL(expr)                                                             \
    do {                                                                               \
        const hipError_t dvs_ce_ = (expr);                                             \
        if (dvs_ce_ != hipSuccess) dvs_note_hip_error(#expr, (int)dvs_ce_, hipGetErrorString(dvs_ce_)); \
    } while (0)
#endif

extern "C" int dvs_version(void) { return DVS_VERSION; }
extern "C" const char* dvs_last_error(void) { return g_err; }

extern "C" int dvs_device_cus(void) {
#ifdef DVS_EMU
    return 2;
#else
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
#endif
}

// ---- flat parameter layout ---------------------------------------------------------------------------------
namespace {
struct LayoutBuilder {
    int64_t off = 0;
    dvs_param_entry* table;
    int cap, count = 0;
    int64_t add(const char* name, int rows, int cols) {
        const int64_t o = off;
        if (table && count < cap) {
            dvs_param_entry& e = table[count];
            memset(&e, 0, sizeof(e));
            snprintf(e.name, sizeof(e.name), "%s", name);
            e.offset = o;
            e.rows = rows;
            e.cols = cols;
        }
        ++count;
        const int64_t n = (int64_t)rows * (cols ? cols : 1);
        off += (n + 3) & ~(int64_t)3;
        return o;
    }
};
}  // namespace

DvsLayout dvs_make_layout(int N, int C, dvs_param_entry* table, int cap, int* count) {
    LayoutBuilder b;
    b.table = table;
    b.cap = cap;
    DvsLayout l;
    char nm[96];
    auto attn = [&](const char* prefix, DvsAttnP& p) {
        snprintf(nm, sizeof(nm), "%s.in_proj_weight", prefix);  p.in_w = b.add(nm, 192, 64);
        snprintf(nm, sizeof(nm), "%s.in_proj_bias", prefix);    p.in_b = b.add(nm, 192, 0);
        snprintf(nm, sizeof(nm), "%s.out_proj.weight", prefix); p.out_w = b.add(nm, 64, 64);
        snprintf(nm, sizeof(nm), "%s.out_proj.bias", prefix);   p.out_b = b.add(nm, 64, 0);
    };
    auto ffn = [&](const char* prefix, DvsFfnP& p) {
        snprintf(nm, sizeof(nm), "%s.linear1.weight", prefix); p.l1_w = b.add(nm, 64, 64);
        snprintf(nm, sizeof(nm), "%s.linear1.bias", prefix);   p.l1_b = b.add(nm, 64, 0);
        snprintf(nm, sizeof(nm), "%s.linear2.weight", prefix); p.l2_w = b.add(nm, 64, 64);
        snprintf(nm, sizeof(nm), "%s.linear2.bias", prefix);   p.l2_b = b.add(nm, 64, 0);
    };
    auto norm = [&](const char* prefix, int k, DvsNormP& p) {
        snprintf(nm, sizeof(nm), "%s.norm%d.weight", prefix, k); p.w = b.add(nm, 64, 0);
        snprintf(nm, sizeof(nm), "%s.norm%d.bias", prefix, k);   p.b = b.add(nm, 64, 0);
    };
    l.W1 = b.add("vertex_position_embed.W1", 2 * N, 64);
    l.W2 = b.add("vertex_position_embed.W2", 64, 32);
    l.lab_w = b.add("vertex_label_embed.0.weight", 32, C);
    l.lab_b = b.add("vertex_label_embed.0.bias", 32, 0);
    char pre[64], pre2[80];
    for (int i = 0; i < DVS_LAYERS; ++i) {
        snprintf(pre, sizeof(pre), "encoder.layers.%d", i);
        snprintf(pre2, sizeof(pre2), "%s.self_attn", pre);
        attn(pre2, l.enc[i].sa);
        ffn(pre, l.enc[i].ff);
        norm(pre, 1, l.enc[i].n1);
        norm(pre, 2, l.enc[i].n2);
    }
    l.fc1_w = b.add("fc1.weight", 32, N * 64);
    l.fc1_b = b.add("fc1.bias", 32, 0);
    l.fc2_w = b.add("fc2.weight", 32, N * 64);
    l.fc2_b = b.add("fc2.bias", 32, 0);
    for (int i = 0; i < DVS_LAYERS; ++i) {
        snprintf(pre, sizeof(pre), "decoder.layers.%d", i);
        snprintf(pre2, sizeof(pre2), "%s.self_attn", pre);
        attn(pre2, l.dec[i].sa);
        snprintf(pre2, sizeof(pre2), "%s.multihead_attn", pre);
        attn(pre2, l.dec[i].ca);
        ffn(pre, l.dec[i].ff);
        norm(pre, 1, l.dec[i].n1);
        norm(pre, 2, l.dec[i].n2);
        norm(pre, 3, l.dec[i].n3);
    }
    l.node0_w = b.add("add_node.0.weight", 32, 64);
    l.node0_b = b.add("add_node.0.bias", 32, 0);
    l.node2_w = b.add("add_node.2.weight", C, 32);
    l.node2_b = b.add("add_node.2.bias", C, 0);
    l.edge0_w = b.add("add_edge.0.weight", 64, 128);
    l.edge0_b = b.add("add_edge.0.bias", 64, 0);
    l.edge2_w = b.add("add_edge.2.weight", 1, 64);
    l.edge2_b = b.add("add_edge.2.bias", 1, 0);
    l.fc3_w = b.add("fc3.weight", N * 64, 32);
    l.fc3_b = b.add("fc3.bias", N * 64, 0);
    l.total = b.off;
    if (count) *count = b.count;
    return l;
}

DvsWorkspace dvs_make_workspace(int B, int NT, int64_t P, int nslab, bool wide) {
    DvsWorkspace w;
    size_t off = 0;
    auto take = [&](size_t n) {
        const size_t o = off;
        off += (n + 63) & ~(size_t)63;
        return o;
    };
    const size_t tile = (size_t)B * NT * 1024;
    for (int s = 0; s < DVS_NSLOTS; ++s) {
        w.act[s] = take(tile);
        w.stats[s] = take((size_t)B * NT * 32);
    }
    w.enc_out = take(tile);
    w.mu = take((size_t)B * 32);
    w.logvar = take((size_t)B * 32);
    w.z = take((size_t)B * 32);
    w.epsv = take((size_t)B * 32);
    w.mem = take(tile);
    w.dag_loss = take((size_t)B * 2);
    w.gA = take(tile);
    w.gB = take(tile);
    w.gq = take(tile);
    w.gk = take(tile);
    w.gv = take(tile);
    w.gmem = take(tile);
    w.genc = take(tile);
    w.gz = take((size_t)B * 64);
    w.nslab = nslab;
    w.slabs = take((size_t)nslab * (size_t)P);
    w.fcpart = take((size_t)DVS_FC_PARTS * (size_t)P);
    w.wimg = take((DVS_WIMG_BF16 + 1) / 2);
    w.limg = take(DvsLatImg::floats(NT));
    for (int blk = 0; blk < 9; ++blk) w.qkv[blk] = take(wide ? (size_t)B * NT * 12 * 256 : 0);
    w.total_floats = off;
    return w;
}

// ---- shape checks / derived dims ------------------------------------------------------------------------------
static int check_shape(const dvs_shape* s) {
    if (!s) return fail(1, "dvs: null shape");
    if (s->batch <= 0) return fail(2, "dvs: batch must be > 0");
    if (s->n_tokens < 4 || s->n_tokens > DVS_WTOK)
        return fail(3, "dvs: n_tokens (= max_num_vertices + 3) must be in [4, 48] in this build");
    if (s->n_classes < 4 || s->n_classes > DVS_WTOK)
        return fail(4, "dvs: n_classes (= vertex_label_cardinality + 3) must be in [4, 48] in this build");
    if (!(s->dropout >= 0.f && s->dropout < 1.f)) return fail(5, "dvs: dropout must be in [0, 1)");
    return 0;
}

static bool is_wide(const dvs_shape* s);
static int64_t param_floats(const dvs_shape* s) { return dvs_make_layout(s->n_tokens, s->n_classes, nullptr, 0, nullptr).total; }
int dvs_num_slabs();

// Caller-owned buffers against what the shape needs (include/dvs.h: code 14); nothing has been enqueued yet.
// Pass a negative / zero "have" for a buffer the entry point does not take.
static int check_buffers(const dvs_shape* s, const char* fn, bool has_records, size_t records_bytes, bool has_params,
                         int64_t n_params, bool has_ws, size_t workspace_bytes) {
    char msg[240];
    if (has_records) {
        const size_t need = (size_t)s->batch * ((s->n_tokens > DVS_MAXTOK || s->n_classes > 16) ? sizeof(DvsRecordW) : sizeof(DvsRecord));
        if (records_bytes < need) {
            snprintf(msg, sizeof(msg), "%s: records_bytes %zu < batch * dvs_record_bytes = %zu", fn, records_bytes, need);
            return fail(14, msg);
        }
    }
    if (has_params) {
        const int64_t need = param_floats(s);
        if (n_params < need) {
            snprintf(msg, sizeof(msg), "%s: n_params %lld < dvs_param_count = %lld", fn, (long long)n_params, (long long)need);
            return fail(14, msg);
        }
    }
    if (has_ws) {
        const size_t need = dvs_make_workspace(s->batch, (s->n_tokens + 15) / 16, param_floats(s), dvs_num_slabs(), is_wide(s)).total_floats * sizeof(float);
        if (workspace_bytes < need) {
            snprintf(msg, sizeof(msg), "%s: workspace_bytes %zu < dvs_workspace_bytes = %zu", fn, workspace_bytes, need);
            return fail(14, msg);
        }
    }
    return 0;
}

// One-tile path: a wave owns a whole DAG (N, C <= 16).  Wide path: NT tiles of 16 tokens per DAG, cross-token kernels
// of dvs_wide.h (also taken when only the class count exceeds one tile).
static bool is_wide(const dvs_shape* s) { return s->n_tokens > DVS_MAXTOK || s->n_classes > 16; }
static int tiles_of(const dvs_shape* s) { return (s->n_tokens + 15) / 16; }

static DvsDims make_dims(const dvs_shape* s) {
    DvsDims d;
    d.NT = tiles_of(s);
    d.B = s->batch;
    d.N = s->n_tokens;
    d.C = s->n_classes;
    d.training = s->training ? 1 : 0;
    d.drop.thr16 = (uint32_t)lrintf(s->dropout * 65536.0f);
    d.drop.on = (d.training && d.drop.thr16 > 0) ? 1 : 0;
    d.drop.scale = 1.0f / (1.0f - (float)d.drop.thr16 / 65536.0f);
    d.seed_lo = (uint32_t)(s->seed & 0xFFFFFFFFull);
    d.seed_hi = (uint32_t)(s->seed >> 32);
    d.dag_offset = s->dag_offset;
    d.beta = s->beta;
    d.eps_scale = s->eps_scale;
    d.debug = 0;
    return d;
}

static int grid_for(int units, int per_wg = 8) {   // persistent forward kernels: `per_wg` units (tiles / DAGs) per pass
    const int cus = dvs_device_cus();
    const int want = (units + per_wg - 1) / per_wg;
    const int cap = cus > 0 ? cus : 256;
    return want < cap ? want : cap;
}

int dvs_num_slabs() {   // backward kernels run on at most this many workgroups (one gradient slab each); sizes the workspace
    const int cus = dvs_device_cus();
    return cus > 0 ? cus : 256;
}

// Waves per workgroup of the one-tile stack kernels (k_fwd_stack / k_bwd_stack and their per-phase twins).  8: two waves per
// SIMD, 8 DAGs per workgroup and pass — the mapping every kernel was tuned for; it fills the chip from 8 x #CU DAGs up.  Below
// 4 x #CU DAGs (1 024 on an MI355X: a 4 096 batch cut over 4 or 8 GPUs) half of the CUs or more would idle while the others
// run two waves per SIMD, so the NARROW mapping takes over: 4 waves (one cooperative weight-gradient group), 4 DAGs per
// workgroup, twice the workgroups, one wave per SIMD.  DVS_WAVES_PER_WG=4|8 forces either (A/B runs, tests).
static int waves_per_wg(const DvsDims& d, bool wide) {
    const char* env = getenv("DVS_WAVES_PER_WG");         // read per call: the tests switch it between calls
    const int force = env ? atoi(env) : 0;
    if (wide) return 8;
    if (force == 4 || force == 8) return force;
    return d.B <= 4 * dvs_num_slabs() ? 4 : 8;
}
// Workgroups of the backward kernels = gradient slabs that are written (and reduced) this step: all of them from 4 x #CU
// DAGs up; fewer for smaller batches, so that k_reduce_slabs does not stream slabs of zeros.
static int active_slabs(const DvsDims& d, bool wide) {
    const int all = dvs_num_slabs();
    if (wide) return all;
    const int want = (d.B + 3) / 4;          // the narrowest backward kernels own 4 DAGs per workgroup and pass
    return want < all ? want : all;
}

extern "C" int64_t dvs_param_count(const dvs_shape* s) {
    if (check_shape(s)) return -1;
    return dvs_make_layout(s->n_tokens, s->n_classes, nullptr, 0, nullptr).total;
}

extern "C" int dvs_param_table(const dvs_shape* s, dvs_param_entry* out, int cap) {
    if (check_shape(s)) return -1;
    int count = 0;
    dvs_make_layout(s->n_tokens, s->n_classes, out, cap, &count);
    return count;
}

extern "C" size_t dvs_workspace_bytes(const dvs_shape* s) {
    if (check_shape(s)) return 0;
    const int64_t P = dvs_make_layout(s->n_tokens, s->n_classes, nullptr, 0, nullptr).total;
    return dvs_make_workspace(s->batch, tiles_of(s), P, dvs_num_slabs(), is_wide(s)).total_floats * sizeof(float);
}

extern "C" size_t dvs_record_bytes(const dvs_shape* s) {
    if (check_shape(s)) return 0;
    return is_wide(s) ? sizeof(DvsRecordW) : sizeof(DvsRecord);
}

extern "C" int dvs_pack_features(const dvs_shape* s, const float* label_onehot, const float* pos_onehot,
                                 const float* adjacency, const uint8_t* target_masks, void* records, size_t records_bytes,
                                 int32_t* status, void* stream) {
    if (int e = check_shape(s)) return e;
    if (!label_onehot || !pos_onehot || !adjacency || !target_masks || !records || !status)
        return fail(10, "dvs_pack_features: null pointer");
    if (int e = check_buffers(s, "dvs_pack_features", true, records_bytes, false, 0, false, 0)) return e;
    call_begin();
    PackArgs a;
    a.B = s->batch;
    a.N = s->n_tokens;
    a.C = s->n_classes;
    a.lab1h = label_onehot;
    a.pos1h = pos_onehot;
    a.adj = adjacency;
    a.tmask = target_masks;
    a.rec = (DvsRecord*)records;
    a.status = status;
    if (is_wide(s)) dvs_launch_pack_w(a, (dvs_stream_t)stream);
    else dvs_launch_pack(a, (dvs_stream_t)stream);
    return call_end("dvs_pack_features");
}

extern "C" int dvs_build_records(const dvs_shape* s, const uint8_t* labels, const void* preds, void* records,
                                 size_t records_bytes, int32_t* status, void* stream) {
    if (int e = check_shape(s)) return e;
    if (!labels || !preds || !records || !status) return fail(10, "dvs_build_records: null pointer");
    if (int e = check_buffers(s, "dvs_build_records", true, records_bytes, false, 0, false, 0)) return e;
    call_begin();
    if (is_wide(s)) {
        BuildWArgs a;
        a.B = s->batch;
        a.N = s->n_tokens;
        a.C = s->n_classes;
        a.labels = labels;
        a.preds = (const uint64_t*)preds;
        a.rec = (DvsRecordW*)records;
        a.status = status;
        dvs_launch_build_records_w(a, (dvs_stream_t)stream);
        return call_end("dvs_build_records");
    }
    BuildArgs a;
    a.B = s->batch;
    a.N = s->n_tokens;
    a.C = s->n_classes;
    a.labels = labels;
    a.preds = (const uint16_t*)preds;
    a.rec = (DvsRecord*)records;
    a.status = status;
    dvs_launch_build_records(a, (dvs_stream_t)stream);
    return call_end("dvs_build_records");
}

// slot numbering of saved activations
static inline int slot_enc(int layer, int sub) { return 1 + 2 * layer + sub; }        // sub 0 attn, 1 ffn
static inline int slot_dec(int layer, int sub) { return 8 + 3 * layer + sub; }        // sub 0 self, 1 cross, 2 ffn
// dropout sites (34 per step): 0,1 enc-embed; 2,3 dec-embed; 4+4l+{0..3} encoder; 16+6l+{0..5} decoder
static inline int site_enc(int layer, int k) { return 4 + 4 * layer + k; }
static inline int site_dec(int layer, int k) { return 16 + 6 * layer + k; }

// ---- per-step weight images (dvs_wimg.h) ------------------------------------------------------------------------------
static inline size_t img_attn(int block) { return (size_t)block * DvsAttnImg::SIZE; }                       // 0..8
static inline size_t img_ffn(int block) { return DVS_N_ATTN_BLOCKS * DvsAttnImg::SIZE + (size_t)block * DvsFfnImg::SIZE; }   // 0..5
static inline int blk_enc_attn(int layer) { return layer; }
static inline int blk_dec_self(int layer) { return 3 + 2 * layer; }
static inline int blk_dec_cross(int layer) { return 4 + 2 * layer; }
static inline int blk_enc_ffn(int layer) { return layer; }
static inline int blk_dec_ffn(int layer) { return 3 + layer; }

static void prepare_images(const DvsLayout& L, int N, int C, bool wide, const float* params, float* ws, const DvsWorkspace& W,
                           dvs_stream_t st) {
    DvsImgJobs J;
    J.count = 0;
    auto add = [&](int64_t src, size_t dst, int rows, int flags) {
        DvsImgJob& j = J.job[J.count++];
        j.src = src;
        j.dst = (int64_t)dst;
        j.rows = rows;
        j.flags = flags;
    };
    auto attn = [&](const DvsAttnP& p, int block) {
        const size_t b = img_attn(block);
        // the one-tile kernels keep q, k, v in head-aligned slot order (dvs_pi: in-projection rows / out-projection columns
        // permuted); the wide kernels (k_wide_fwd.hip) use parameter order
        add(p.in_w, b + DvsAttnImg::Win, 192, wide ? 0 : 2);             // x6
        add(p.out_w, b + DvsAttnImg::Wout, 64, wide ? 0 : 4);            // x6
        add(p.out_w, b + DvsAttnImg::WoutT, 64, 1 | (wide ? 0 : 4));     // x3 transposed
        add(p.in_w, b + DvsAttnImg::WinB, 192, 16 | (wide ? 0 : 2));     // parts hi, mid again, behind WoutT
        for (int q = 0; q < 3; ++q)                                    // W_q^T, W_k^T, W_v^T for k_proj_bwd
            add(p.in_w + 4096 * q, b + DvsAttnImg::WinT + (size_t)q * 2 * DVS_IMG64, 64, 1 | (wide ? 0 : 2));
    };
    auto ffn = [&](const DvsFfnP& p, int block) {
        const size_t b = img_ffn(block);
        add(p.l1_w, b + DvsFfnImg::W1, 64, 0);
        add(p.l2_w, b + DvsFfnImg::W2, 64, 0);
        add(p.l2_w, b + DvsFfnImg::W2T, 64, 1);
        add(p.l1_w, b + DvsFfnImg::W1T, 64, 1);
    };
    for (int i = 0; i < DVS_LAYERS; ++i) {
        attn(L.enc[i].sa, blk_enc_attn(i));
        ffn(L.enc[i].ff, blk_enc_ffn(i));
        attn(L.dec[i].sa, blk_dec_self(i));
        attn(L.dec[i].ca, blk_dec_cross(i));
        ffn(L.dec[i].ff, blk_dec_ffn(i));
    }
    if (!wide) {                 // loss head (one-tile kernels): the two halves of add_edge.0.weight [64][128]
        add(L.edge0_w, DVS_WIMG_LOSS + DvsLossImg::Wa, 64, 8);
        add(L.edge0_w + 64, DVS_WIMG_LOSS + DvsLossImg::Wb, 64, 8);
        add(L.edge0_w, DVS_WIMG_LOSS + DvsLossImg::WaT, 64, 8 | 1);
        add(L.edge0_w + 64, DVS_WIMG_LOSS + DvsLossImg::WbT, 64, 8 | 1);
    }
    DvsLatImgArgs lat;
    lat.fc1_w = params + L.fc1_w;
    lat.fc2_w = params + L.fc2_w;
    lat.fc3_w = params + L.fc3_w;
    lat.fc3_b = params + L.fc3_b;
    lat.img = ws + W.limg;
    lat.N = N;
    lat.NT = (N + 15) / 16;
    DvsLossHeadArgs head;
    memset(&head, 0, sizeof(head));
    if (!wide) {
        head.node0_w = params + L.node0_w;
        head.node0_b = params + L.node0_b;
        head.node2_w = params + L.node2_w;
        head.node2_b = params + L.node2_b;
        head.edge0_b = params + L.edge0_b;
        head.edge2_w = params + L.edge2_w;
        head.edge2_b = params + L.edge2_b;
        head.ln_g = params + L.dec[DVS_LAYERS - 1].n3.w;
        head.ln_b = params + L.dec[DVS_LAYERS - 1].n3.b;
        head.dst = (float*)((dvs_bf16*)(ws + W.wimg) + DVS_WIMG_LOSS + DvsLossImg::Head);
        head.C = C;
        head.W1 = params + L.W1;
        head.W2 = params + L.W2;
        head.lab_w = params + L.lab_w;
        head.lab_b = params + L.lab_b;
        head.dst_emb = (float*)((dvs_bf16*)(ws + W.wimg) + DVS_WIMG_EMB);
        head.N = N;
    }
    dvs_launch_prepare_images(J, params, (dvs_bf16*)(ws + W.wimg), lat, head, st);
}
static inline const void* wimg_attn(const float* ws, const DvsWorkspace& W, int block) {
    return (const dvs_bf16*)(ws + W.wimg) + img_attn(block);
}
static inline const void* wimg_ffn(const float* ws, const DvsWorkspace& W, int block) {
    return (const dvs_bf16*)(ws + W.wimg) + img_ffn(block);
}

// launch grids of the forward kernels
struct FwdGrids {
    bool wide;
    int nw;         // waves per workgroup of the one-tile stack kernels (waves_per_wg)
    int chain;      // k_fwd_stack / k_attn_fwd / narrow k_ffn_fwd, k_embed_fwd, k_loss_fwd: nw tiles per workgroup and pass
    int tiles16;    // 16-wave tile-parallel kernels (k_ffn_fwd, one-tile k_embed_fwd)
    int attn;       // one-tile k_attn_fwd (DVS_ATTN_FWD_THREADS / 64 waves per workgroup)
    int tiles8;     // 8-wave tile-parallel kernels (one-tile k_attn_fwd / k_embed_fwd / k_loss_fwd)
    int tiles4;     // 4-wave tile-parallel kernels (k_embed_fwd_w)
    int dags;       // workgroup-per-DAG kernels of the wide path
    int dags2;      // ... those that fit two workgroups per CU
};
static FwdGrids fwd_grids(const DvsDims& d, bool wide) {
    FwdGrids g;
    g.wide = wide;
    g.nw = waves_per_wg(d, wide);
    g.chain = grid_for(d.B * d.NT, g.nw);
    g.tiles16 = grid_for(d.B * d.NT, 16);
    g.attn = grid_for(d.B * d.NT, dvs_attn_fwd_waves());
    g.tiles8 = grid_for(d.B * d.NT, 8);
    g.tiles4 = grid_for(d.B * d.NT, 4);
    g.dags = grid_for(d.B, 1);
    g.dags2 = d.B < 2 * g.dags ? d.B : 2 * g.dags;      // kernels that fit two workgroups per CU (k_loss_fwd_w)
    return g;
}
static void launch_embed_fwd(const EmbedArgs& e, const FwdGrids& g, dvs_stream_t st) {
    if (g.wide) dvs_launch_embed_fwd_w(e, g.tiles4, st);
    else if (g.nw == 4) dvs_launch_embed_fwd(e, g.chain, 4, st);
    else dvs_launch_embed_fwd(e, g.tiles16, 16, st);
}
static void launch_attn_fwd(const AttnArgs& a, const FwdGrids& g, dvs_stream_t st) {
    if (g.wide) dvs_launch_attn_fwd_w(a, g.dags, st);
    else if (g.nw == 4) dvs_launch_attn_fwd(a, g.chain, 4, st);
    else dvs_launch_attn_fwd(a, g.attn, 8, st);
}
// One-tile path: the sublayers of the encoder / decoder are chained into one launch each (k_fwd_stack); the wide path and
// DVS_SPLIT_STACK=1 (per-phase profiling) launch every sublayer on its own.
struct FwdChain {
    FwdStackArgs stack;
    const FwdGrids& g;
    dvs_stream_t st;
    int tag;
    bool chain;
    FwdChain(const FwdGrids& grids, int tag_, dvs_stream_t st_) : g(grids), st(st_), tag(tag_) {
        static const bool split_env = getenv("DVS_SPLIT_STACK") && atoi(getenv("DVS_SPLIT_STACK")) != 0;
        memset(&stack, 0, sizeof(stack));
        chain = !g.wide && !split_env;
    }
    FwdPhase& next(int kind) {
        if (stack.nphase == DVS_FWD_STACK_PHASES) flush();
        FwdPhase& ph = stack.ph[stack.nphase++];
        ph.kind = kind;
        return ph;
    }
    void attn(const AttnArgs& a) {
        if (chain) next(DVS_FPH_ATTN).u.a = a;
        else launch_attn_fwd(a, g, st);
    }
    void ffn(const FfnArgs& f) {
        if (chain) next(DVS_FPH_FFN).u.f = f;
        else if (g.nw == 4 && !g.wide) dvs_launch_ffn_fwd(f, g.chain, 4, st);
        else dvs_launch_ffn_fwd(f, g.tiles16, 16, st);
    }
    // the latent block as the last phase of the encoder chain (the workgroups of the chain own 16 DAGs each: one MFMA
    // group); false: not chained, the caller launches k_latent_fwd
    bool latent(const LatentArgs& l) {
        static const bool off = getenv("DVS_LATENT_KERNELS") && atoi(getenv("DVS_LATENT_KERNELS")) != 0;   // A/B: own launches
        // (narrow mapping: a workgroup owns 4 DAGs, a quarter of an MFMA column group: the latent block keeps its own launch)
        if (!chain || off || g.nw != 8 || stack.nphase == 0 || stack.nphase == DVS_FWD_STACK_PHASES) return false;
        next(DVS_FPH_LATENT).u.l = l;
        return true;
    }
    void flush() {
        if (stack.nphase > 0) dvs_launch_fwd_stack(stack, tag, g.chain, g.nw, st);
        stack.nphase = 0;
    }
};

// dec_embed: also write the decoder-side embedding (slot 7, dropout sites 2 / 3) from the same launch (one-tile path)
// lat: the latent block's arguments; it runs as the last phase of the encoder chain when there is one, as k_latent_fwd otherwise
static void encoder_forward(const DvsDims& d, const DvsLayout& L, const DvsWorkspace& W, const DvsRecord* rec,
                            const float* P, float* ws, const FwdGrids& grid, dvs_stream_t st, const LatentArgs& lat,
                            bool dec_embed = false, bool save_qkv = false) {
    EmbedArgs e;
    memset(&e, 0, sizeof(e));
    e.dims = d;
    e.rec = rec;
    e.W1 = P + L.W1;
    e.W2 = P + L.W2;
    e.lab_w = P + L.lab_w;
    e.lab_b = P + L.lab_b;
    e.embimg = (const float*)((const dvs_bf16*)(ws + W.wimg) + DVS_WIMG_EMB);
    e.out = ws + W.act[0];
    e.site = 0;
    if (dec_embed) {
        e.out2 = ws + W.act[7];
        e.site2 = 2;
    }
    launch_embed_fwd(e, grid, st);
    FwdChain chain(grid, 0, st);
    DvsLN ln = {nullptr, nullptr, nullptr};
    int prev = 0;
    for (int i = 0; i < DVS_LAYERS; ++i) {
        AttnArgs a;
        memset(&a, 0, sizeof(a));
        a.dims = d;
        a.rec = rec;
        a.xin = ws + W.act[prev];
        a.ln = ln;
        a.kv = nullptr;
        a.in_w = P + L.enc[i].sa.in_w;
        a.in_b = P + L.enc[i].sa.in_b;
        a.out_w = P + L.enc[i].sa.out_w;
        a.out_b = P + L.enc[i].sa.out_b;
        a.wimg = wimg_attn(ws, W, blk_enc_attn(i));
        a.qkv = (save_qkv && grid.wide) ? ws + W.qkv[blk_enc_attn(i)] : nullptr;
        const int sa = slot_enc(i, 0);
        a.out_pre = ws + W.act[sa];
        a.out_stats = ws + W.stats[sa];
        a.site_prob = site_enc(i, 0);
        a.site_post = site_enc(i, 1);
        chain.attn(a);
        FfnArgs f;
        memset(&f, 0, sizeof(f));
        f.dims = d;
        f.xin = ws + W.act[sa];
        f.ln = DvsLN{ws + W.stats[sa], P + L.enc[i].n1.w, P + L.enc[i].n1.b};
        f.l1_w = P + L.enc[i].ff.l1_w;
        f.l1_b = P + L.enc[i].ff.l1_b;
        f.l2_w = P + L.enc[i].ff.l2_w;
        f.l2_b = P + L.enc[i].ff.l2_b;
        f.wimg = wimg_ffn(ws, W, blk_enc_ffn(i));
        const int sf = slot_enc(i, 1);
        f.out_pre = ws + W.act[sf];
        f.out_stats = ws + W.stats[sf];
        f.site_hidden = site_enc(i, 2);
        f.site_post = site_enc(i, 3);
        if (i == DVS_LAYERS - 1) {
            f.out_norm = ws + W.enc_out;
            f.ng = P + L.enc[i].n2.w;
            f.nb = P + L.enc[i].n2.b;
        }
        chain.ffn(f);
        ln = DvsLN{ws + W.stats[sf], P + L.enc[i].n2.w, P + L.enc[i].n2.b};
        prev = sf;
    }
    const bool fused = chain.latent(lat);
    chain.flush();
    if (!fused) dvs_launch_latent_fwd(lat, st);
}

static LatentArgs latent_args(const DvsDims& d, const DvsLayout& L, const DvsWorkspace& W, const float* P, float* ws,
                              const float* eps, bool with_mem) {
    LatentArgs a;
    memset(&a, 0, sizeof(a));
    a.dims = d;
    a.xenc = ws + W.enc_out;
    a.fc1_w = P + L.fc1_w;
    a.fc1_b = P + L.fc1_b;
    a.fc2_w = P + L.fc2_w;
    a.fc2_b = P + L.fc2_b;
    a.fc3_w = P + L.fc3_w;
    a.fc3_b = P + L.fc3_b;
    a.limg = ws + W.limg;
    a.eps_in = eps;
    a.mu = ws + W.mu;
    a.logvar = ws + W.logvar;
    a.z = ws + W.z;
    a.epsv = ws + W.epsv;
    a.mem = with_mem ? ws + W.mem : nullptr;
    a.dag_loss = ws + W.dag_loss;
    return a;
}

LossArgs dvs_loss_args(const DvsDims& d, const DvsLayout& L, const DvsWorkspace& W, const DvsRecord* rec, const float* P,
                       float* ws) {
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.dims = d;
    a.rec = rec;
    const int last = slot_dec(DVS_LAYERS - 1, 2);
    a.xin = ws + W.act[last];
    a.ln = DvsLN{ws + W.stats[last], P + L.dec[DVS_LAYERS - 1].n3.w, P + L.dec[DVS_LAYERS - 1].n3.b};
    a.node0_w = P + L.node0_w;
    a.node0_b = P + L.node0_b;
    a.node2_w = P + L.node2_w;
    a.node2_b = P + L.node2_b;
    a.edge0_w = P + L.edge0_w;
    a.edge0_b = P + L.edge0_b;
    a.edge2_w = P + L.edge2_w;
    a.edge2_b = P + L.edge2_b;
    a.wimg = (const dvs_bf16*)(ws + W.wimg) + DVS_WIMG_LOSS;
    a.dag_loss = ws + W.dag_loss;
    return a;
}

// TransformerDecoder forward (pace.py:163-182) from the embedding in slot `dec_in`; memory = W.mem.
static void decoder_forward(const DvsDims& d, const DvsLayout& L, const DvsWorkspace& W, const DvsRecord* rec,
                            const float* params, float* ws, const FwdGrids& grid, int dec_in, dvs_stream_t st) {
    FwdChain chain(grid, 1, st);
    DvsLN ln = {nullptr, nullptr, nullptr};
    int prev = dec_in;
    for (int i = 0; i < DVS_LAYERS; ++i) {
        const auto& pl = L.dec[i];
        AttnArgs a;
        memset(&a, 0, sizeof(a));
        a.dims = d;
        a.rec = rec;
        a.xin = ws + W.act[prev];
        a.ln = ln;
        a.in_w = params + pl.sa.in_w;
        a.in_b = params + pl.sa.in_b;
        a.out_w = params + pl.sa.out_w;
        a.out_b = params + pl.sa.out_b;
        a.wimg = wimg_attn(ws, W, blk_dec_self(i));
        a.qkv = grid.wide ? ws + W.qkv[blk_dec_self(i)] : nullptr;
        const int s0 = slot_dec(i, 0);
        a.out_pre = ws + W.act[s0];
        a.out_stats = ws + W.stats[s0];
        a.site_prob = site_dec(i, 0);
        a.site_post = site_dec(i, 1);
        chain.attn(a);

        AttnArgs c;
        memset(&c, 0, sizeof(c));
        c.dims = d;
        c.rec = rec;
        c.xin = ws + W.act[s0];
        c.ln = DvsLN{ws + W.stats[s0], params + pl.n1.w, params + pl.n1.b};
        c.kv = ws + W.mem;
        c.in_w = params + pl.ca.in_w;
        c.in_b = params + pl.ca.in_b;
        c.out_w = params + pl.ca.out_w;
        c.out_b = params + pl.ca.out_b;
        c.wimg = wimg_attn(ws, W, blk_dec_cross(i));
        c.qkv = grid.wide ? ws + W.qkv[blk_dec_cross(i)] : nullptr;
        const int s1 = slot_dec(i, 1);
        c.out_pre = ws + W.act[s1];
        c.out_stats = ws + W.stats[s1];
        c.site_prob = site_dec(i, 2);
        c.site_post = site_dec(i, 3);
        chain.attn(c);

        FfnArgs f;
        memset(&f, 0, sizeof(f));
        f.dims = d;
        f.xin = ws + W.act[s1];
        f.ln = DvsLN{ws + W.stats[s1], params + pl.n2.w, params + pl.n2.b};
        f.l1_w = params + pl.ff.l1_w;
        f.l1_b = params + pl.ff.l1_b;
        f.l2_w = params + pl.ff.l2_w;
        f.l2_b = params + pl.ff.l2_b;
        f.wimg = wimg_ffn(ws, W, blk_dec_ffn(i));
        const int s2 = slot_dec(i, 2);
        f.out_pre = ws + W.act[s2];
        f.out_stats = ws + W.stats[s2];
        f.site_hidden = site_dec(i, 4);
        f.site_post = site_dec(i, 5);
        chain.ffn(f);
        ln = DvsLN{ws + W.stats[s2], params + pl.n3.w, params + pl.n3.b};
        prev = s2;
    }
    chain.flush();
}

extern "C" int dvs_loss_forward(const dvs_shape* s, const void* records, size_t records_bytes, const float* params,
                                int64_t n_params, void* workspace, size_t workspace_bytes, const float* eps,
                                const int32_t* status, float* losses, float* mu, float* logvar, void* stream) {
    return dvs_loss_forward_notify(s, records, records_bytes, params, n_params, workspace, workspace_bytes, eps,
                                   (int32_t*)status, losses, mu, logvar, nullptr, 0u, stream);     // no host_tail: status is only read
}

extern "C" int dvs_loss_forward_notify(const dvs_shape* s, const void* records, size_t records_bytes, const float* params,
                                       int64_t n_params, void* workspace, size_t workspace_bytes, const float* eps,
                                       int32_t* status, float* losses, float* mu, float* logvar, void* host_tail,
                                       uint32_t host_seq, void* stream) {
    if (int e = check_shape(s)) return e;
    if (!records || !params || !workspace || !losses) return fail(10, "dvs_loss_forward: null pointer");
    if (int e = check_buffers(s, "dvs_loss_forward", true, records_bytes, true, n_params, true, workspace_bytes)) return e;
    call_begin();
    const DvsDims d = make_dims(s);
    const DvsLayout L = dvs_make_layout(d.N, d.C, nullptr, 0, nullptr);
    const DvsWorkspace W = dvs_make_workspace(d.B, d.NT, L.total, dvs_num_slabs(), is_wide(s));
    float* ws = (float*)workspace;
    const DvsRecord* rec = (const DvsRecord*)records;
    dvs_stream_t st = (dvs_stream_t)stream;
    const FwdGrids grid = fwd_grids(d, is_wide(s));

    prepare_images(L, d.N, d.C, grid.wide, params, ws, W, st);
    const bool fused_dec_embed = d.drop.on;
    encoder_forward(d, L, W, rec, params, ws, grid, st, latent_args(d, L, W, params, ws, eps, true), fused_dec_embed, true);

    // decoder input embedding: identical to the encoder's in eval mode / dropout 0 (pace.py:2000-2012 recomputes it
    // only to redraw the dropout masks)
    int dec_in = fused_dec_embed ? 7 : 0;
    if (d.drop.on && !fused_dec_embed) {
        EmbedArgs e;
        memset(&e, 0, sizeof(e));
        e.dims = d;
        e.rec = rec;
        e.W1 = params + L.W1;
        e.W2 = params + L.W2;
        e.lab_w = params + L.lab_w;
        e.lab_b = params + L.lab_b;
        e.embimg = (const float*)((const dvs_bf16*)(ws + W.wimg) + DVS_WIMG_EMB);
        e.out = ws + W.act[7];
        e.site = 2;
        launch_embed_fwd(e, grid, st);
        dec_in = 7;
    }
    decoder_forward(d, L, W, rec, params, ws, grid, dec_in, st);
    if (grid.wide) dvs_launch_loss_fwd_w(dvs_loss_args(d, L, W, rec, params, ws), grid.dags2, st);
    else if (grid.nw == 4) dvs_launch_loss_fwd(dvs_loss_args(d, L, W, rec, params, ws), grid.chain, 4, st);
    else dvs_launch_loss_fwd(dvs_loss_args(d, L, W, rec, params, ws), grid.tiles8, 8, st);
    FinalizeArgs fa;
    fa.B = d.B;
    fa.beta = d.beta;
    fa.dag_loss = ws + W.dag_loss;
    fa.status = status;
    fa.losses = losses;
    fa.host_tail = (float*)host_tail;
    fa.host_seq = host_seq;
    dvs_launch_finalize(fa, st);
    const size_t nb = (size_t)d.B * 32 * sizeof(float);
#ifdef DVS_EMU
    if (mu) memcpy(mu, ws + W.mu, nb);
    if (logvar) memcpy(logvar, ws + W.logvar, nb);
#else
    if (mu) DVS_HIP_CALL(hipMemcpyAsync(mu, ws + W.mu, nb, hipMemcpyDeviceToDevice, st));
    if (logvar) DVS_HIP_CALL(hipMemcpyAsync(logvar, ws + W.logvar, nb, hipMemcpyDeviceToDevice, st));
#endif
    return call_end("dvs_loss_forward");
}

extern "C" int dvs_encode(const dvs_shape* s, const void* records, size_t records_bytes, const float* params,
                          int64_t n_params, void* workspace, size_t workspace_bytes, float* mu, float* logvar,
                          void* stream) {
    if (int e = check_shape(s)) return e;
    if (!records || !params || !workspace || !mu || !logvar) return fail(10, "dvs_encode: null pointer");
    if (int e = check_buffers(s, "dvs_encode", true, records_bytes, true, n_params, true, workspace_bytes)) return e;
    call_begin();
    const DvsDims d = make_dims(s);
    const DvsLayout L = dvs_make_layout(d.N, d.C, nullptr, 0, nullptr);
    const DvsWorkspace W = dvs_make_workspace(d.B, d.NT, L.total, dvs_num_slabs(), is_wide(s));
    float* ws = (float*)workspace;
    dvs_stream_t st = (dvs_stream_t)stream;
    prepare_images(L, d.N, d.C, is_wide(s), params, ws, W, st);
    LatentArgs la = latent_args(d, L, W, params, ws, nullptr, false);
    la.dims.training = 0;
    encoder_forward(d, L, W, (const DvsRecord*)records, params, ws, fwd_grids(d, is_wide(s)), st, la);
    const size_t nb = (size_t)d.B * 32 * sizeof(float);
#ifdef DVS_EMU
    memcpy(mu, ws + W.mu, nb);
    memcpy(logvar, ws + W.logvar, nb);
#else
    DVS_HIP_CALL(hipMemcpyAsync(mu, ws + W.mu, nb, hipMemcpyDeviceToDevice, st));
    DVS_HIP_CALL(hipMemcpyAsync(logvar, ws + W.logvar, nb, hipMemcpyDeviceToDevice, st));
#endif
    return call_end("dvs_encode");
}

// ---- generation (k_decode.hip) -------------------------------------------------------------------------------------
#include "dvs_decode.h"

extern "C" int dvs_decode(const dvs_shape* s, const float* params, int64_t n_params, void* workspace,
                          size_t workspace_bytes, void* records, size_t records_bytes, const float* z,
                          const float* uniforms, void* state_out, size_t state_bytes, void* stream) {
    if (int e = check_shape(s)) return e;
    if (!params || !workspace || !records || !z || !state_out) return fail(10, "dvs_decode: null pointer");
    if (s->training) return fail(13, "dvs_decode: generation runs in eval mode (shape.training must be 0)");
    if (int e = check_buffers(s, "dvs_decode", true, records_bytes, true, n_params, true, workspace_bytes)) return e;
    if (state_bytes < (size_t)s->batch * sizeof(dvs_decode_state))
        return fail(14, "dvs_decode: state_bytes < batch * sizeof(dvs_decode_state)");
    call_begin();
    const DvsDims d = make_dims(s);
    const DvsLayout L = dvs_make_layout(d.N, d.C, nullptr, 0, nullptr);
    const DvsWorkspace W = dvs_make_workspace(d.B, d.NT, L.total, dvs_num_slabs(), is_wide(s));
    float* ws = (float*)workspace;
    dvs_stream_t st = (dvs_stream_t)stream;
    const bool wide = is_wide(s);
    const FwdGrids grid = fwd_grids(d, wide);
    const DvsRecord* rec = (const DvsRecord*)records;

    prepare_images(L, d.N, d.C, wide, params, ws, W, st);
    dvs_launch_decode_memory(d, z, params + L.fc3_w, params + L.fc3_b, ws + W.mem, st);
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.dims = d;
    a.wide = wide ? 1 : 0;
    a.rec = records;
    a.state = (DvsDecodeState*)state_out;
    const int last = slot_dec(DVS_LAYERS - 1, 2);
    a.xin = ws + W.act[last];
    a.ln = DvsLN{ws + W.stats[last], params + L.dec[DVS_LAYERS - 1].n3.w, params + L.dec[DVS_LAYERS - 1].n3.b};
    a.node0_w = params + L.node0_w;
    a.node0_b = params + L.node0_b;
    a.node2_w = params + L.node2_w;
    a.node2_b = params + L.node2_b;
    a.edge0_w = params + L.edge0_w;
    a.edge0_b = params + L.edge0_b;
    a.edge2_w = params + L.edge2_w;
    a.edge2_b = params + L.edge2_b;
    a.uniforms = uniforms;
    dvs_launch_decode_init(a, st);
    for (int idx = 2; idx < d.N; ++idx) {
        EmbedArgs e;
        memset(&e, 0, sizeof(e));
        e.dims = d;
        e.rec = rec;
        e.W1 = params + L.W1;
        e.W2 = params + L.W2;
        e.lab_w = params + L.lab_w;
        e.lab_b = params + L.lab_b;
        e.embimg = (const float*)((const dvs_bf16*)(ws + W.wimg) + DVS_WIMG_EMB);
        e.out = ws + W.act[7];
        e.site = 2;
        launch_embed_fwd(e, grid, st);
        decoder_forward(d, L, W, rec, params, ws, grid, 7, st);
        a.idx = idx;
        dvs_launch_decode_step(a, grid_for(d.B, 4), st);
    }
    return call_end("dvs_decode");
}

extern "C" int dvs_bic_scores_impl(int B, int n, int S, const uint64_t* data, const uint8_t* card, const uint64_t* parents,
                                   double* local, double* out, int* status, void* stream);
extern "C" int dvs_bic_scores(int32_t batch, int32_t n_vars, int32_t n_samples, const uint64_t* data, const uint8_t* card,
                              const uint64_t* parents, double* scratch, double* out, int32_t* status, void* stream) {
    if (batch <= 0 || n_samples <= 0) return fail(2, "dvs_bic_scores: batch and n_samples must be > 0");
    if (n_vars < 1 || n_vars > DVS_WTOK) return fail(3, "dvs_bic_scores: n_vars must be in [1, 48]");
    if (!data || !card || !parents || !scratch || !out || !status) return fail(10, "dvs_bic_scores: null pointer");
    call_begin();
    if (int e = dvs_bic_scores_impl(batch, n_vars, n_samples, data, card, parents, scratch, out, status, stream)) return e;
    return call_end("dvs_bic_scores");
}

extern "C" int dvs_bic_parent_masks_impl(int B, int n, int wide, const uint8_t* labels, const void* preds, uint64_t* parents,
                                         int* status, void* stream);
extern "C" int dvs_bic_parent_masks(int32_t batch, int32_t n_vars, int32_t preds_are_u64, const uint8_t* labels,
                                    const void* preds, uint64_t* parents, int32_t* status, void* stream) {
    if (batch <= 0) return fail(2, "dvs_bic_parent_masks: batch must be > 0");
    if (n_vars < 1 || n_vars > DVS_WTOK) return fail(3, "dvs_bic_parent_masks: n_vars must be in [1, 48]");
    if (!preds_are_u64 && n_vars > 16) return fail(12, "dvs_bic_parent_masks: 16-bit predecessor rows hold at most 16 vertices");
    if (!labels || !preds || !parents || !status) return fail(10, "dvs_bic_parent_masks: null pointer");
    call_begin();
    if (int e = dvs_bic_parent_masks_impl(batch, n_vars, preds_are_u64 ? 1 : 0, labels, preds, parents, status, stream)) return e;
    return call_end("dvs_bic_parent_masks");
}

extern "C" int dvs_gp_predict_impl(int B, int M, int D, const float* x, const float* z, const double* alpha,
                                   double outputscale, double lengthscale, double constant, double* out, void* stream);
extern "C" int dvs_gp_predict(int32_t batch, int32_t n_inducing, int32_t dim, const float* x, const float* inducing,
                              const double* alpha, double outputscale, double lengthscale, double constant, double* out,
                              void* stream) {
    if (batch <= 0 || n_inducing <= 0 || dim <= 0) return fail(2, "dvs_gp_predict: sizes must be > 0");
    if (!(lengthscale > 0.0)) return fail(5, "dvs_gp_predict: lengthscale must be > 0");
    if (!x || !inducing || !alpha || !out) return fail(10, "dvs_gp_predict: null pointer");
    call_begin();
    if (int e = dvs_gp_predict_impl(batch, n_inducing, dim, x, inducing, alpha, outputscale, lengthscale, constant, out, stream)) return e;
    return call_end("dvs_gp_predict");
}

extern "C" int dvs_gp_kernel_impl(int na, int nb, int D, const float* xa, const float* xb, double outputscale, double lengthscale,
                                  double* K, void* stream);
extern "C" int dvs_gp_kernel_backward_impl(int na, int nb, int D, int symmetric, const float* xa, const float* xb,
                                           double outputscale, double lengthscale, const double* G, double* dxa, double* rows,
                                           void* stream);
static int gp_check(const char* fn, int na, int nb, int dim, double outputscale, double lengthscale) {
    char msg[160];
    if (na <= 0 || nb <= 0 || dim <= 0 || dim > 32) {
        snprintf(msg, sizeof(msg), "%s: sizes must be > 0 and dim <= 32", fn);
        return fail(2, msg);
    }
    if (!(lengthscale > 0.0) || !(outputscale > 0.0)) {
        snprintf(msg, sizeof(msg), "%s: lengthscale and outputscale must be > 0", fn);
        return fail(5, msg);
    }
    return 0;
}
extern "C" int dvs_gp_kernel(int32_t na, int32_t nb, int32_t dim, const float* xa, const float* xb, double outputscale,
                             double lengthscale, double* K, void* stream) {
    if (int e = gp_check("dvs_gp_kernel", na, nb, dim, outputscale, lengthscale)) return e;
    if (!xa || !xb || !K) return fail(10, "dvs_gp_kernel: null pointer");
    call_begin();
    if (int e = dvs_gp_kernel_impl(na, nb, dim, xa, xb, outputscale, lengthscale, K, stream)) return e;
    return call_end("dvs_gp_kernel");
}
extern "C" int dvs_gp_kernel_backward(int32_t na, int32_t nb, int32_t dim, int32_t symmetric, const float* xa, const float* xb,
                                      double outputscale, double lengthscale, const double* G, double* dxa, double* row_sums,
                                      void* stream) {
    if (int e = gp_check("dvs_gp_kernel_backward", na, nb, dim, outputscale, lengthscale)) return e;
    if (!xa || !xb || !G || !dxa || !row_sums) return fail(10, "dvs_gp_kernel_backward: null pointer");
    if (symmetric && na != nb) return fail(12, "dvs_gp_kernel_backward: symmetric needs na == nb");
    call_begin();
    if (int e = dvs_gp_kernel_backward_impl(na, nb, dim, symmetric, xa, xb, outputscale, lengthscale, G, dxa, row_sums, stream))
        return e;
    return call_end("dvs_gp_kernel_backward");
}

extern "C" int dvs_debug_activation(const dvs_shape* s, const void* workspace, int slot, float* out, void* stream) {
    if (int e = check_shape(s)) return e;
    const int64_t P = dvs_make_layout(s->n_tokens, s->n_classes, nullptr, 0, nullptr).total;
    const DvsWorkspace W = dvs_make_workspace(s->batch, tiles_of(s), P, dvs_num_slabs(), is_wide(s));
    const float* ws = (const float*)workspace;
    const float* src = nullptr;
    if (slot >= 0 && slot < DVS_NSLOTS) src = ws + W.act[slot];
    else if (slot == 100) src = ws + W.enc_out;
    else if (slot == 101) src = ws + W.mem;
    else if (slot == 102) src = ws + W.gA;
    else if (slot == 103) src = ws + W.gB;
    else if (slot == 104) src = ws + W.gmem;
    else if (slot == 105) src = ws + W.genc;
    else return fail(11, "dvs_debug_activation: bad slot");
    call_begin();
    dvs_launch_unfrag(src, out, s->batch * tiles_of(s), (dvs_stream_t)stream);
    return call_end("dvs_debug_activation");
}

// Error-path test hook (include/dvs.h): an empty kernel through the product's launch macro.
__global__ void k_debug_empty(int* sink) {
    DVS_DYN_LDS(smem);
    if (sink && threadIdx.x == 1u << 20) *sink = smem[0];
}
extern "C" int dvs_debug_launch(size_t dynamic_lds_bytes, void* stream) {
    call_begin();
    DVS_LAUNCH(k_debug_empty, dim3(1), dim3(64), dynamic_lds_bytes, (dvs_stream_t)stream, (int*)nullptr);
    return call_end("dvs_debug_launch");
}

#include "dvs_api_backward.inc"
