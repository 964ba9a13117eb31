/* dvs.h — C ABI of libdvs_hip.so: the MI355X-native PACE-VAE train-step hot path.
 *
 * The reference (rlog58/dags-vae-search) is 100 % Python and has no FFI of its own: its boundary for this
 * path is the Python class surface of PaceVaeV3 (src/encoders/pace.py:1139-2046) and train_batch
 * (experiments/03_synthetic_12/main.py:95-118).  This header is the C ABI that sits UNDER the drop-in
 * Python mirror (dags_vae_search_amd/pace.py, train.py); each entry point cites the reference code it
 * replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every pointer marked "device" is device
 * memory owned by the caller; the library never allocates or frees device memory and never synchronises
 * the host with the device.  All work is enqueued on the passed HIP stream (void* = hipStream_t).
 * Return value: 0 = ok, non-zero = error code; dvs_last_error() gives the thread-local message.  Codes: 1-5 bad shape,
 * 10 null pointer, 12/13 bad argument, 14 a caller buffer is smaller than the shape needs (records_bytes, n_params,
 * workspace_bytes, state_bytes are checked against dvs_record_bytes / dvs_param_count / dvs_workspace_bytes BEFORE anything
 * is enqueued), 20 the HIP runtime refused a kernel launch or an attribute/copy call of this entry point (message names
 * the kernel and carries hipGetErrorString; work enqueued before the failing launch stays enqueued, results are undefined).
 * Entry points are re-entrant: the only process-global state is the optional profiler record (dvs_profile_*), which is
 * mutex-guarded; error state is thread-local.
 *
 * Fixed architecture of this build (BASELINE.json configs; experiments/01_bn_asia/main.py:33-43):
 * vertices_embedding_size 32, num_heads 8, num_layers 3, ff_hidden_size 64, latent_layer_size 32,
 * fc_hidden 32.  n_tokens = n + 3 <= 48, n_classes = card + 3 <= 48.  Shapes with n_tokens <= 16 and n_classes <= 16
 * (asia, sachs, n = 12) take the one-tile path (a wavefront owns a DAG); larger ones (alarm-size, n = 37) the tiled
 * "wide" path (a workgroup owns a DAG, tiles of 16 tokens meet in LDS).  Same entry points for both.
 */
#ifndef DVS_H
#define DVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVS_VERSION 202
#define DVS_NUM_PARAMS 108
#define DVS_RECORD_BYTES 96          /* one-tile path */
#define DVS_RECORD_BYTES_WIDE 864    /* wide path; dvs_record_bytes(shape) returns the one that applies */
#define DVS_CLIP_SCRATCH_FLOATS 4096  /* 2 + partial sums of squares: 256 (dvs_clip_adam) or one per 256 parameters (dvs_loss_backward_sq) */
#define DVS_DECODE_STATE_BYTES 440   /* sizeof(dvs_decode_state) */

typedef struct dvs_shape {
    int32_t batch;        /* DAGs in this (rank-local) batch */
    int32_t n_tokens;     /* N = max_num_vertices + 3 (pace.py:1159), <= 48 */
    int32_t n_classes;    /* C = vertex_label_cardinality + 3 (pace.py:1160), <= 48 */
    int32_t training;     /* 1 = model.train(): dropout + reparameterisation noise; 0 = eval */
    float dropout;        /* p of every nn.Dropout / attention dropout (pace.py:1150) */
    float beta;           /* KL weight (pace.py:1977) */
    float eps_scale;      /* epsilon_scale of reparameterize (pace.py:1653), 0.01 */
    uint32_t dag_offset;  /* global index of this batch's first DAG (data-parallel shard offset) */
    uint64_t seed;        /* counter-based RNG seed; fold the step number in on the host */
} dvs_shape;

typedef struct dvs_param_entry {
    char name[64];        /* reference state-dict key, e.g. "encoder.layers.0.self_attn.in_proj_weight" */
    int64_t offset;       /* float offset inside the flat parameter / gradient buffer (16-byte aligned) */
    int32_t rows, cols;   /* cols == 0 for 1-D tensors */
} dvs_param_entry;

int dvs_version(void);
const char* dvs_last_error(void);
int dvs_device_cus(void);   /* compute units of the current device (grid sizing; informational) */

/* Flat parameter buffer: the 108 tensors of PaceVaeV3.state_dict() (pace.py:1176-1207) in registration
 * order, each 16-byte aligned.  dvs_param_table fills up to `cap` entries and returns the count. */
int64_t dvs_param_count(const dvs_shape* s);
int dvs_param_table(const dvs_shape* s, dvs_param_entry* out, int cap);

/* Scratch needed by forward+backward for s->batch DAGs (saved activations, gradient slabs). */
size_t dvs_workspace_bytes(const dvs_shape* s);

/* Bytes of one compact per-DAG record for this shape (96 or 864); the caller allocates batch * this. */
size_t dvs_record_bytes(const dvs_shape* s);

/* Replaces the `.to(device)` feature hand-over at pace.py:1981-1985 / 1616-1619: reads the reference-layout
 * dense features — vertex_label_features [B,N,C] f32, vertex_position_features [B,N,N] f32,
 * adjacency_matrices [B,N,N] f32, target_masks [8B,N,N] bool (1 byte each) — and writes one compact record
 * (dvs_record_bytes) per DAG.  status (device int32[1], zeroed by the caller) gets bit 0 set if a label/position
 * row is not one-hot, bit 1 if the 8 per-head masks of a DAG differ, bit 2 if a mask row forbids self. */
int dvs_pack_features(const dvs_shape* s, const float* label_onehot, const float* pos_onehot,
                      const float* adjacency, const uint8_t* target_masks, void* records, size_t records_bytes,
                      int32_t* status, void* stream);

/* Device-side feature front-end (replaces LabeledDag.from_dict_to_graph src/toolkit/labeled.py:132-154 +
 * from_labeled_graph_to_pace_graph pace.py:1250-1288 + generate_mask 1307-1343 + prepare_features 1345-1478 + pack):
 * builds the records straight from the row codec.  labels: device u8 [B][n] (n = n_tokens - 3, l{v} columns);
 * preds: device [B][n], bit u of preds[b][v] set <=> edge u -> v (the e{v} '0/1' string, u < v); element type u16 on
 * the one-tile path (dvs_record_bytes == 96), u64 on the wide path.  One thread per DAG does the PACE wrapping, the
 * FIFO-Kahn topological order (positions[v] = order[v], the reference's quirk), and the ancestor closure on bit rows.  status bit 0: a label is >= n_classes - 3; bit 3: an edge with u >= v. */
int dvs_build_records(const dvs_shape* s, const uint8_t* labels, const void* preds, void* records, size_t records_bytes,
                      int32_t* status, void* stream);

/* Buffer sizes: every compute entry point takes the byte size of the record buffer (>= batch * dvs_record_bytes), the
 * float count of the flat parameter (and gradient) buffer (>= dvs_param_count) and the byte size of the workspace
 * (>= dvs_workspace_bytes) and returns 14 without enqueueing anything when one is too small. */

/* PaceVaeV3.loss_direct forward (pace.py:1974-2035).  eps: optional device [B,32] noise already multiplied
 * by eps_scale (NULL = counter-based normal draws when training).  losses (device f32[DVS_LOSS_FLOATS = 5]):
 * {total, recon = -log-likelihood, kld, non-finite flag, invalid-features flag}.  status: optional device int32[1], the
 * validation word of the dvs_pack_features / dvs_build_records call that wrote `records`; losses[4] = 1 if it is
 * non-zero (0 when status is NULL), so that the flag can travel with the loss scalars (data-parallel all-reduce,
 * dvs_clip_adam's guard).  mu/logvar: optional device [B,32] outputs. */
#define DVS_LOSS_FLOATS 5
int dvs_loss_forward(const dvs_shape* s, const void* records, size_t records_bytes, const float* params, int64_t n_params,
                     void* workspace, size_t workspace_bytes, const float* eps, const int32_t* status, float* losses,
                     float* mu, float* logvar, void* stream);

/* dvs_loss_forward that also tells the HOST when the loss scalars are final, without an event or a copy on any stream (ABI 201).
 * host_tail: 16 bytes (16-byte aligned) of pinned, device-mapped host memory (hipHostMalloc / torch pin_memory).  The kernel
 * that reduces the per-DAG losses writes them with ONE 16-byte store: [0] total, [1] recon, [2] kld (f32), [3] a uint32 word
 * = (host_seq << 8) | (invalid-features flag << 7) | (non-finite flag << 6) | (validation word *status & 0x3F).  A host that
 * polls word [3] until its upper 24 bits equal the host_seq it passed (24 bits are kept), and reads [0..2] AFTER that, has final
 * values where the reference's `loss.item()` returns (experiments/03_synthetic_12/main.py:104), while backward and optimiser
 * are still queued.  With host_tail the validation word is RE-ARMED (*status = 0) once it has been read: status is written.
 * Coherence REQUIREMENT on host_tail (the library cannot check it): fine-grained / coherent pinned memory — what
 * hipHostMalloc gives by default (hipHostMallocCoherent) and what torch.pin_memory() allocates —, uncached on the device, so
 * that the single 16-byte store becomes one PCIe write the host sees whole; memory registered non-coherent
 * (hipHostMallocNonCoherent, hipExtHostRegisterCoarseGrained) is only guaranteed visible at the end of the kernel and must
 * not be used.  A careful host re-reads word [3] AFTER reading [0..2] and retries when it changed (the Python driver does);
 * a host that cannot rely on this passes host_tail = NULL and waits for an event instead (DVS_EARLY_READ=event). */
int dvs_loss_forward_notify(const dvs_shape* s, const void* records, size_t records_bytes, const float* params,
                            int64_t n_params, void* workspace, size_t workspace_bytes, const float* eps, int32_t* status,
                            float* losses, float* mu, float* logvar, void* host_tail, uint32_t host_seq, void* stream);

/* Backward of the same step (autograd of pace.py:1974-2035; experiments/03_synthetic_12/main.py:114).  Must follow
 * dvs_loss_forward on the same workspace and parameters: it reads the forward's saved activations and per-step weight images.
 * gcoef (device f32[2]): d(objective)/d(recon), d(objective)/d(kld).  grads: flat buffer, overwritten. */
int dvs_loss_backward(const dvs_shape* s, const void* records, size_t records_bytes, const float* params, int64_t n_params,
                      void* workspace, size_t workspace_bytes, const float* gcoef, float* grads, void* stream);

/* dvs_loss_backward that also leaves the partial sums of squares of `grads` — one per 256 gradient entries, written by the
 * kernel that sums the gradient slabs, in a fixed order — in clip_scratch[2 ..] (device f32[DVS_CLIP_SCRATCH_FLOATS], the
 * `scratch` of the dvs_clip_adam_from_partials call that follows; NULL: plain dvs_loss_backward).  (ABI 202.)  For the
 * single-process step only: clip_grad_norm_ (experiments/03_synthetic_12/main.py:115) needs the norm of the gradient the
 * optimiser sees, so a data-parallel step, whose gradient changes in the all-reduce, uses dvs_loss_backward + dvs_clip_adam. */
int dvs_loss_backward_sq(const dvs_shape* s, const void* records, size_t records_bytes, const float* params, int64_t n_params,
                         void* workspace, size_t workspace_bytes, const float* gcoef, float* grads, float* clip_scratch,
                         void* stream);

/* PaceVaeV3.encode_direct (pace.py:1613-1641): mu, logvar device [B,32]. */
int dvs_encode(const dvs_shape* s, const void* records, size_t records_bytes, const float* params, int64_t n_params,
               void* workspace, size_t workspace_bytes, float* mu, float* logvar, void* stream);

/* clip_grad_norm_(params, max_norm) + Adam.step (experiments/03_synthetic_12/main.py:115-116, lr 1e-4,
 * betas (0.9, 0.999), eps 1e-8, no weight decay) over flat buffers of n floats.  max_norm <= 0 disables
 * clipping.  scratch: device f32[DVS_CLIP_SCRATCH_FLOATS] ([0] = sum of squares, [1] = clip coefficient, rest =
 * partial sums); `step` is the 1-based Adam step.  guard: optional device f32[2] (normally &losses[3] of the step's
 * dvs_loss_forward, after the data-parallel all-reduce): when guard[0] != 0 (non-finite loss) or guard[1] != 0 (invalid
 * features) the whole update is skipped on the device — params, exp_avg, exp_avg_sq and grads stay as they are, as in the
 * reference, where loss_direct raises before backward / clip / step run (pace.py:97-98, main.py:111-116). */
int dvs_clip_adam(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float adam_eps, int64_t step, float max_norm, float* scratch,
                  const float* guard, void* stream);

/* The same update from the partial sums of squares dvs_loss_backward_sq left in scratch[2 ..] for exactly these n gradient
 * entries (one launch instead of two: no pass over the gradient for its norm).  (ABI 202.) */
int dvs_clip_adam_from_partials(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float lr,
                                float beta1, float beta2, float adam_eps, int64_t step, float max_norm, float* scratch,
                                const float* guard, void* stream);

/* One grown PACE graph of dvs_decode (vertex 0 = start, 1 = input, then the sampled vertices in order). */
typedef struct dvs_decode_state {
    uint64_t parents[48];   /* bit j of parents[i]: edge j -> i */
    uint8_t label[48];      /* PACE label of vertex i (user label + 3; 0 input, 1 output, 2 start) */
    int32_t nv;             /* number of vertices; == n_tokens unless the graph sampled `output` early */
    int32_t finished;       /* 1: the graph sampled `output` and stopped growing (pace.py:1738-1743) */
} dvs_decode_state;

/* PaceVaeV3.decode (pace.py:1666-1749), batched on the device.  z: device [B,32] latents;
 * records: device scratch of batch * dvs_record_bytes bytes; state_out: device dvs_decode_state[B].  The whole
 * n_tokens - 2 step autoregressive loop (records of the partial graphs -> embedding -> 3 decoder layers -> node-type /
 * edge sampling -> graph update) is enqueued on `stream`; nothing is read back in between.  uniforms: optional device
 * f32 [B, n_tokens, n_tokens]; step idx uses [b, idx, 0] for the node type (inverse CDF, as np.random.choice) and
 * [b, idx, 1 + vi] for edge candidate vi (edge iff u < sigmoid score, as torch.rand_like < score); NULL = counter-based
 * draws from s->seed.  s->training must be 0 (the reference decodes in eval mode). */
int dvs_decode(const dvs_shape* s, const float* params, int64_t n_params, void* workspace, size_t workspace_bytes,
               void* records, size_t records_bytes, const float* z, const float* uniforms, void* state_out,
               size_t state_bytes, void* stream);

/* BIC of B discrete Bayesian-network structures on one data set (replaces BNLearnWrapper.score,
 * src/problem/bn/bnlearn.py:27-61 = `Rscript bnlearn_score.R`: bnlearn::score(net, data, type = "bic")).
 * data: device u64 [n_samples][ceil(n_vars/16)], variable i's level code (0..15) in bits 4*(i%16).. of word i/16;
 * card: device u8 [n_vars] level counts; parents: device u64 [B][n_vars], bit u of parents[b][v] <=> edge u -> v in
 * DATASET variable indices (bnlearn.py:40-45 maps graph vertex v to variable labels[v]); scratch: device f64
 * [B][n_vars]; out: device f64 [B].  Tables of up to 36 864 (configuration, level) cells are counted densely in LDS;
 * larger parent sets go through an LDS sort of the samples, which needs n_samples <= 16 384 and <= 63 key bits —
 * otherwise bit 4 of status (device int32, zeroed by the caller) is set and that DAG's score is NaN.  n_vars <= 48. */
int dvs_bic_scores(int32_t batch, int32_t n_vars, int32_t n_samples, const uint64_t* data, const uint8_t* card,
                   const uint64_t* parents, double* scratch, double* out, int32_t* status, void* stream);

/* The relabelling step of BNLearnWrapper.score (src/problem/bn/bnlearn.py:34-45: graph vertex v stands for data-set variable
 * labels[v]) on the device, from the row codec of dvs_build_records: labels device u8 [B][n_vars], preds device [B][n_vars]
 * (u16, or u64 when preds_are_u64) -> parents device u64 [B][n_vars] in data-set variable indices, ready for dvs_bic_scores.
 * status bit 5 (device int32, zeroed by the caller): the labels of a DAG are not a permutation of 0..n_vars-1 (the reference
 * asserts, bnlearn.py:35); that DAG's masks are zero.  With dvs_encode this keeps the reference's predictor-data pipeline
 * (experiments/01_bn_asia/main.py:268-303: encode -> BIC -> (mu, target) rows) on the device. */
int dvs_bic_parent_masks(int32_t batch, int32_t n_vars, int32_t preds_are_u64, const uint8_t* labels, const void* preds,
                         uint64_t* parents, int32_t* status, void* stream);

/* Predictive mean of the reference's GP predictor (GPRegressionModel, src/predictors/gp.py:13-32:
 * ConstantMean + InducingPointKernel(ScaleKernel(RBFKernel())), evaluated as `model(test_x).mean`,
 * experiments/01_bn_asia/main.py:367-368):  out[b] = constant + outputscale * sum_m alpha[m] exp(-|x_b - z_m|^2 / (2 l^2)).
 * x: device f32 [batch][dim] (encoder means); inducing: device f32 [n_inducing][dim]; alpha: device f64 [n_inducing]
 * (solved once at fit time from the training set, see bic/predictor mirror); out: device f64 [batch].  fp64
 * accumulation.  Parity with gpytorch is unpinned (not installed; no reference predictions exist). */
int dvs_gp_predict(int32_t batch, int32_t n_inducing, int32_t dim, const float* x, const float* inducing,
                   const double* alpha, double outputscale, double lengthscale, double constant, double* out,
                   void* stream);

/* Hyper-parameter training of the same predictor (reference loop: src/predictors/gp.py:55-81 = experiments/01_bn_asia/
 * main.py:329-365, Adam lr 0.01 on -ExactMarginalLogLikelihood(likelihood, model) of the SGPR model): the two kernel-specific
 * steps of one iteration.  The M x M / M x n Cholesky factorisations between them are dense library calls of the host side
 * (dags_vae_search_amd/predictor.py), the parameter update is dvs_clip_adam over the flat [inducing points | 4 raw scalars].
 * dvs_gp_kernel: K [na][nb] (device f64) = outputscale * exp(-|xa_a - xb_b|^2 / (2 lengthscale^2)); xa [na][dim], xb [nb][dim]
 * device f32, dim <= 32.
 * dvs_gp_kernel_backward: G = d objective / d K [na][nb] (device f64) -> dxa [na][dim] (device f64, overwritten):
 * sum_b G'_ab K_ab (xb_b - xa_a) / l^2 with G' = G + G^T when `symmetric` (xa and xb are the same point set, K_uu), else
 * G' = G; row_sums [na][2] (device f64, overwritten): per-row partial sums of d/d lengthscale (sum_b G K d^2 / l^3) and
 * d/d outputscale (sum_b G K / o); the caller adds the rows.  Fixed summation order (bitwise reproducible). */
int dvs_gp_kernel(int32_t na, int32_t nb, int32_t dim, const float* xa, const float* xb, double outputscale,
                  double lengthscale, double* K, void* stream);
int dvs_gp_kernel_backward(int32_t na, int32_t nb, int32_t dim, int32_t symmetric, const float* xa, const float* xb,
                           double outputscale, double lengthscale, const double* G, double* dxa, double* row_sums,
                           void* stream);

/* Optional per-kernel timing for the benchmark's roofline leg: while enabled, every kernel launch is bracketed by
 * HIP events recorded on its own stream; dvs_profile_collect waits for them and returns, per kernel name, the
 * number of launches and their summed duration in milliseconds (rows of `name_stride` chars).  Process-global
 * debug state; leave disabled in production. */
void dvs_profile_enable(int on);
int dvs_profile_collect(char* names, int name_stride, int* counts, float* total_ms, int cap);

/* Test hook for the error path: launches an empty kernel with `dynamic_lds_bytes` of dynamic LDS through the same launch
 * macro as every product kernel.  A request above the 160 KB of a gfx950 CU must come back as code 20. */
int dvs_debug_launch(size_t dynamic_lds_bytes, void* stream);

/* Debug/test access: copy saved activation `slot` (natural [B, 16*ceil(n_tokens/16), 64] layout) out of the workspace. */
int dvs_debug_activation(const dvs_shape* s, const void* workspace, int slot, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
