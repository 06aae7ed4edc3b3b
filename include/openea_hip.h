/*
 * openea_hip.h -- C ABI of libopenea_hip.so: the MI355X (gfx950) implementation of OpenEA's
 * training / evaluation hot path.
 *
 * The reference (nju-websoft/OpenEA) has NO FFI of its own: the path runs as TensorFlow-1 stock
 * ops plus numpy/scipy calls.  Each entry point below therefore names the reference call site
 * (file:line relative to /root/reference/src/openea/) whose arithmetic it replaces; the Python
 * mirror of the reference's module API (openea_amd/modules/...) is the only caller.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.
 *   - every `*_dev` / unqualified data pointer is a HIP DEVICE pointer unless the name ends in
 *     `_host`; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - all functions return 0 on success, a negative OEA_E* code otherwise;
 *     oea_last_error() returns a thread-local message for the last failure.
 *   - tables are fp32 row-major [rows, ld] with ld >= dim and ld % 4 == 0; columns
 *     [dim, ld) must be zero and are kept zero.  ids are int32.  Triples are int32 [n,3]
 *     (head, relation, tail) interleaved.
 *   - calls are asynchronous on `stream`; the caller synchronises.  Not re-entrant on the
 *     same buffers.
 */
#ifndef OPENEA_HIP_H
#define OPENEA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OEA_OK 0
#define OEA_EINVAL (-1)   /* bad argument */
#define OEA_EHIP (-2)     /* HIP runtime error */
#define OEA_ENOMEM (-3)
#define OEA_EUNSUPPORTED (-4)

int oea_version(void);
const char *oea_last_error(void);
/* number of visible HIP devices, <0 on error (used by the host side to fail loudly) */
int oea_device_count(void);

/* Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline
 * figure; no reference counterpart).  Between begin and end, every sampled optimiser step carries 4
 * events: [m0 fwd_bwd kernel m1] [m2 apply kernel m3] (a GRAD-phase call the first pair, the
 * APPLY-phase call the second), each pair ATTACHED to its kernel's dispatch (hipExtLaunchKernelGGL
 * start / stop events = the dispatch's own begin / end timestamps, the figure rocprofv3's kernel
 * trace reports) rather than recorded as separate barrier packets around it.  oea_profile_end(4, ms, &n) returns ms[0] = total fwd_bwd
 * time, ms[1] = time between the two kernels (the exchange under data parallelism), ms[2] = total
 * apply time (milliseconds) over n steps.
 * stride: only every stride-th step records its marks (an event record costs a few microseconds of
 * enqueue time, comparable to the kernels themselves at the 15K shape). */
int oea_profile_begin(int32_t stride);
int oea_profile_end(int32_t group, double *out_ms_host, int32_t *n_calls_host);

/* ---------------------------------------------------------------------------------------
 * Embedding store -- replaces the tf.Variable tables of BasicModel._define_variables
 * (models/basic_model.py:73-78) and their .eval() round trips (basic_model.py:106-121,
 * 184-204).  Payload layout of load/save_host == the C-contiguous fp32 [rows, dim] payload
 * of ent_embeds.npy (modules/load/read.py:325-349).
 * ------------------------------------------------------------------------------------- */
typedef struct oea_store *oea_store_t;
int oea_store_create(int64_t rows, int32_t dim, oea_store_t *out);
int oea_store_destroy(oea_store_t s);
int64_t oea_store_rows(oea_store_t s);
int32_t oea_store_dim(oea_store_t s);
int32_t oea_store_ld(oea_store_t s);
float *oea_store_rows_ptr(oea_store_t s);                       /* device pointer [rows, ld] */
int oea_store_load_host(oea_store_t s, const float *src_host, void *stream);  /* [rows, dim] */
int oea_store_save_host(oea_store_t s, float *dst_host, void *stream);        /* [rows, dim] */

/* out[i, 0:dim] = table[ids[i]] (optionally row-L2-normalised: tf.nn.l2_normalize,
 * modules/base/initializers.py:26) -- tf.nn.embedding_lookup, basic_model.py:89-94,106-121.
 * out is [n, out_ld]; columns [dim, out_ld) are zero-filled. */
int oea_gather_rows(const float *table, int32_t dim, int32_t ld, const int32_t *ids, int64_t n,
                    int32_t normalize, float *out, int32_t out_ld, void *stream);
/* in place: table[r] = l2_normalize(table[r]) for all rows (sklearn normalize semantics when
 * sk != 0: zero rows stay zero; TF semantics otherwise: x*rsqrt(max(ss,1e-12))) */
int oea_normalize_rows(float *table, int64_t rows, int32_t dim, int32_t ld, int32_t sk, void *stream);
int oea_fill_f32(float *p, int64_t n, float value, void *stream);
/* device <-> host copies ordered on `stream`, complete on return (staging of oea_comm_init_callbacks hosts) */
int oea_copy_to_host(const void *dev, void *host, size_t bytes, void *stream);
int oea_copy_from_host(void *dev, const void *host, size_t bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Translational step -- replaces one session.run([triple_loss, triple_optimizer]) of
 * BasicModel.launch_triple_training_1epo (basic_model.py:222-232): gather
 * (basic_model.py:89-94) -> l2_normalize (initializers.py:26) -> loss (modules/base/losses.py:
 * 15-73, approaches/bootea.py:197) -> gradient -> duplicate rows summed -> optimizer
 * (modules/base/optimizers.py:4-20).
 * ------------------------------------------------------------------------------------- */
enum { OEA_LOSS_MARGIN = 0, OEA_LOSS_LIMITED = 1, OEA_LOSS_LOGISTIC = 2, OEA_LOSS_POSITIVE = 3,
       OEA_LOSS_ALIGN = 4 };
enum { OEA_OPT_SGD = 0, OEA_OPT_ADAGRAD = 1, OEA_OPT_ADAM = 2, OEA_OPT_ADADELTA = 3 };
enum { OEA_SCORE_TRANSE = 0, OEA_SCORE_TRANSH = 1, OEA_SCORE_TRANSD = 2 };

typedef struct oea_step_cfg {
    int32_t loss_kind;   /* OEA_LOSS_* */
    int32_t l1;          /* 1: args.loss_norm == 'L1'; 0: 'L2' (squared, no sqrt) */
    float margin;        /* margin-based: args.margin */
    float pos_margin;    /* limited: args.pos_margin */
    float neg_margin;    /* limited: args.neg_margin */
    float balance;       /* limited: args.neg_margin_balance */
    int32_t ent_l2_norm; /* args.ent_l2_norm */
    int32_t rel_l2_norm; /* args.rel_l2_norm */
    int32_t opt_kind;    /* OEA_OPT_* (args.optimizer) */
    float lr;            /* args.learning_rate */
    int32_t neg_group_k; /* 0: neg is an arbitrary list (reference order: python set order, batch.py:118).
                            k > 0: n_neg == n_pos*k and neg[p*k .. p*k+k) are the corruptions of pos p as
                            oea_sample_negatives writes them -> one workgroup-lane group scores a positive
                            with its negatives (fewer row reads / atomics; identical arithmetic per triple,
                            entries that are not corruptions of pos p are still scored correctly). */
    int32_t score_kind;  /* OEA_SCORE_TRANSE: s = |h + r - t|;  OEA_SCORE_TRANSH (approaches/bootea_transh.py:58-96):
                            h, t projected on the relation's hyperplane first, h' = h - (h.n) n with
                            n = l2_normalize(l2_normalize(normal[r])) -- also models/trans/transh.py:16-51 (margin
                            pairs, free negative lists).  OEA_SCORE_TRANSD (models/trans/transd.py:16-57):
                            h' = l2_normalize(h + (h.hp) rp) with the transfer vectors hp = ent[ent_transfer_base + h],
                            rp = rel[rel_transfer_base + r] stored in the same tables (below). */
    float *normal;       /* TransH: [n_rel, ld] normal_vector table (trained in place), else NULL */
    float *normal_acc;   /* TransH + Adagrad: its accumulator, else NULL */
    int32_t ent_transfer_base; /* TransD: `ent` has 2*base rows, [0, base) = ent_embeds, [base, 2 base) = ent_transfer */
    int32_t rel_transfer_base; /* TransD: `rel` has 2*base rows, [0, base) = rel_embeds, [base, 2 base) = rel_transfer;
                                  triple ids stay in [0, base); both halves share the l2_norm flag and the optimiser
                                  (transd.py:16-24), so scratch, exchange and apply see ordinary rows */
    /* OEA_OPT_ADAM / OEA_OPT_ADADELTA (optimizers.py:13-16; tf.train defaults): the update is DENSE -- the tables are
     * l2_normalize(variable), the gather gradient comes back through it as a dense tensor -- so every row moves every
     * step.  ent_acc / rel_acc then hold [2, rows, ld]: Adam (m, v) zeros; Adadelta (accum, accum_update) zeros. */
    float beta1;         /* Adam beta1 = 0.9;  Adadelta rho = 0.95 */
    float beta2;         /* Adam beta2 = 0.999 */
    float eps;           /* 1e-8 */
    int32_t opt_t;       /* Adam: 1-based step count of THIS optimiser instance */
} oea_step_cfg;

/* Workspace owned by the caller, sized by oea_step_workspace_bytes(); must be zero-initialised
 * once (hipMemset) before first use; the step leaves it zeroed again. */
size_t oea_step_workspace_bytes(int64_t n_ent, int64_t n_rel, int32_t ld);

/* One optimiser step.  pos/neg: int32 [n,3].  For OEA_LOSS_MARGIN n_neg must equal n_pos
 * (pairs aligned, losses.py:26); neg may be NULL when n_neg == 0 (positive / align losses).
 * ent_acc / rel_acc: Adagrad accumulators (same shape as the tables, initial value 0.1 =
 * tf.train.AdagradOptimizer default), ignored for SGD; Adam / Adadelta: [2, rows, ld] (see oea_step_cfg).
 * loss_accum: device double; the batch loss (sum over the batch, as in the reference) is
 * ADDED to it, so an epoch's loss is read back once (basic_model.py:231-233). */
int oea_triple_step(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                    int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                    const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                    double *loss_accum, void *stream);

/* The same step split at its one exchange point, for entity-table data parallelism
 * (no reference counterpart: the reference is single-device).  OEA_PHASE_GRAD only fills the
 * gradient scratch + touched flags (the first oea_step_exchange_floats() floats of the
 * workspace, one contiguous fp32 region that the host sums across ranks with ONE RCCL
 * all-reduce); OEA_PHASE_APPLY then runs the optimiser on every rank, so the replicas stay
 * bit-identical.  n_pos / n_neg must be repeated unchanged in the APPLY call. */
enum { OEA_PHASE_BOTH = 0, OEA_PHASE_GRAD = 1, OEA_PHASE_APPLY = 2 };
size_t oea_step_exchange_floats(int64_t n_ent, int64_t n_rel, int32_t ld);
/* Element type of the gradient scratch, the touched flags and every buffer of the partition protocol that carries them
 * (`send`, `own`, `rel_x` below): 4 = fp32 accumulated with hardware fp32 atomics (libopenea_hip.so); 8 = int64 FIXED POINT
 * with 32 fractional bits accumulated with 64-bit integer atomics (libopenea_hip_det.so, the same sources built with
 * -DOEA_DET_SCRATCH): integer sums do not depend on the order of the atomics, so a job gives the same bits run to run and
 * for any number of ranks (the collectives then sum int64: OEA_COMM_I64).  oea_step_exchange_floats counts ELEMENTS of this
 * type. */
int32_t oea_step_scratch_elem_bytes(void);
int oea_triple_step_phase(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc,
                          int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos,
                          const int32_t *neg, int64_t n_neg, const oea_step_cfg *cfg, void *workspace,
                          double *loss_accum, int32_t phase, void *stream);

/* work items of the scoring kernel of a step with n_pos positives and n_neg negatives under `cfg` (= the n_items oea_part_apply is
 * told, which decides how many loss partials it adds up): n_pos when the negatives ride with their positives, n_pos + n_neg else. */
int64_t oea_step_items(const oea_step_cfg *cfg, int64_t n_pos, int64_t n_neg);

/* Entity-id partitioning of the same step across `world` processes (one per GPU): owner of entity row id = id mod world
 * (BASELINE.json north_star; SURVEY 8e: ids are degree-ordered, contiguous ranges would put every hub on rank 0).  Every rank
 * keeps a full read copy of the entity table and the optimiser state of ITS rows only (acc_own [rows_per_rank, ld]).
 *   oea_triple_step_phase(..., OEA_PHASE_GRAD)        local gradients into the workspace scratch
 *   oea_part_pack      scratch -> `send` in owner-major order: world chunks of rpr*(ld+1) floats (rpr rows, then their
 *                      touched flags; rpr = oea_part_rows_per_rank), scratch cleared on the way; relation gradients + flags
 *                      -> rel_x [n_rel*(ld+1)]
 *   reduce-scatter of `send` over the ranks (chunk r to rank r: oea_comm_reduce_scatter_f32 / torch.distributed) -> `own`;
 *   all-reduce of rel_x (relations are few: replicated, every rank applies the same update)
 *   oea_part_apply     optimiser on the owned rows (in place in the natural-order table) + relation rows; the owned rows
 *                      after the update also go to `upd` [rpr, ld]; adds the step's loss (n_items = work items of the GRAD call:
 *                      n_pos for grouped negatives / margin pairs, else n_pos + n_neg)
 *   all-gather of `upd` -> `all` [world][rpr][ld]
 *   oea_part_unpack    the other ranks' rows into the local read copy
 * SGD / Adagrad, TransE score.  Result = the single-process step on the concatenated batch. */
int64_t oea_part_rows_per_rank(int64_t n_ent, int32_t world);
size_t oea_part_send_floats(int64_t n_ent, int32_t ld, int32_t world);
int oea_part_pack(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, int32_t world, void *send, void *rel_x,
                  void *stream);
int oea_part_apply(float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel, int32_t ld,
                   int32_t world, int32_t rank, void *own, void *rel_x, float *upd, const oea_step_cfg *cfg, void *workspace,
                   int64_t n_items, double *loss_accum, void *stream);
int oea_part_unpack(float *ent, int64_t n_ent, int32_t ld, int32_t world, int32_t rank, const float *all, void *stream);
/* TransH under the entity-id partition (approaches/bootea_transh.py:58-96): the normal-vector table is relation-sized and
 * replicated.  After the GRAD phase its gradient scratch (copy 0) and touched flags -- [n_rel, ld] and [n_rel] floats at the
 * byte offsets oea_step_normal_scratch reports inside the step workspace -- are summed over the ranks (two small all-reduces),
 * then every rank runs oea_step_apply_normals (the optimiser on the touched rows of cfg->normal, state cfg->normal_acc). */
int oea_step_normal_scratch(int64_t n_ent, int64_t n_rel, int32_t ld, int64_t *grad_offset_bytes, int64_t *touched_offset_bytes);
int oea_step_apply_normals(int64_t n_ent, int64_t n_rel, int32_t ld, const oea_step_cfg *cfg, void *workspace, void *stream);


/* Add externally computed gradients w.r.t. the (normalised) entity rows `ids` into the step's
 * gradient scratch: grad[ids[i]] += src[i].  Followed by oea_triple_step_phase(...,
 * n_pos = n_neg = 0, OEA_PHASE_APPLY) this runs the optimiser for losses that are not
 * translational -- MTransE's mapping loss alpha*(sum|e2 - e1 M|^2 + |M M^T - I|^2)
 * (modules/base/mapping.py:9-19, modules/base/losses.py:76-80), whose d x d GEMMs are plain
 * library GEMMs on the host side. */
int oea_step_scatter_ent_rows(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, const int32_t *ids,
                              int64_t n, const float *src, int32_t src_ld, void *stream);

/* ---------------------------------------------------------------------------------------
 * RotatE step in fp64 -- replaces session.run([triple_loss, triple_optimizer]) and
 * session.run([alignment_loss, alignment_optimizer]) of BootEA_RotatE (approaches/bootea_rotate.py:50-109,148-158,
 * 193-203): variables re_ent_embeds / im_ent_embeds / rel_embeds are tf.float64 (bootea_rotate.py:50-57),
 *   theta = l2n?(rel)[r] * phase_scale,  (a, b) = (h_re + i h_im) e^{i theta} - (t_re + i t_im),  dist = sum_d |(a, b)_d|,
 *   loss = sum softplus(dist+ - gamma) + sum softplus(gamma - dist-)        (= -sum log sigmoid(score), :59-81)
 * and TF's optimiser semantics (optimizers.py:4-20): Adam moves EVERY row every step.
 * ent: [2 n_ent, ld] doubles, rows [0, n_ent) = re_ent_embeds, [n_ent, 2 n_ent) = im_ent_embeds; rel: [n_rel, ld].
 * ent_state / rel_state: Adagrad accumulator [rows, ld] (initial 0.1), or Adam m then v [2, rows, ld] (zeros), NULL for SGD.
 * neg == NULL / n_neg == 0: the positive half alone (alignment loss).  neg_group_k as in oea_step_cfg.
 * phase: OEA_PHASE_*; the exchanged prefix of the workspace is oea_rotate_exchange_doubles() doubles.
 * workspace: oea_rotate_workspace_bytes(), zeroed once; the step leaves it zeroed. ------------------------------- */
typedef struct oea_rotate_cfg {
    double gamma;        /* args.gamma */
    double phase_scale;  /* pi / embedding_range, embedding_range = (gamma + 2.0) / dim  (bootea_rotate.py:29-33,90) */
    double lr;           /* args.learning_rate */
    double beta1, beta2, eps; /* Adam: 0.9, 0.999, 1e-8 (tf.train.AdamOptimizer defaults) */
    int64_t t;           /* Adam: 1-based step count of THIS optimiser instance */
    int32_t ent_l2_norm; /* args.ent_l2_norm (both entity tables) */
    int32_t rel_l2_norm; /* args.rel_l2_norm */
    int32_t opt_kind;    /* OEA_OPT_* */
    int32_t reserved;
} oea_rotate_cfg;
size_t oea_rotate_workspace_bytes(int64_t n_ent, int64_t n_rel, int32_t ld);
size_t oea_rotate_exchange_doubles(int64_t n_ent, int64_t n_rel, int32_t ld);
int oea_rotate_step(double *ent, double *ent_state, int64_t n_ent, double *rel, double *rel_state, int64_t n_rel,
                    int32_t dim, int32_t ld, const int32_t *pos, int64_t n_pos, const int32_t *neg, int64_t n_neg,
                    int32_t neg_group_k, const oea_rotate_cfg *cfg, void *workspace, double *loss_accum, int32_t phase,
                    void *stream);
/* out[i, 0:dim] = (float)( l2n?( part(re[ids[i]]) + part(im[ids[i]]) ) ), part = l2_normalize when part_norm: the
 * embeddings that evaluation, bootstrapping and the neighbour search of BootEA_RotatE read
 * (bootea_rotate.py:111-146,160-167: `re_ent_embeds + im_ent_embeds`, normalised again when sum_norm).
 * ids == NULL: rows 0..n-1.  out is fp32 [n, out_ld], pad columns zero -- the evaluation kernels are fp32. */
int oea_rotate_lookup(const double *ent, int64_t n_ent, int32_t dim, int32_t ld, const int32_t *ids, int64_t n,
                      int32_t part_norm, int32_t sum_norm, float *out, int32_t out_ld, void *stream);

/* MTransE's mapping step, fused (modules/base/mapping.py:9-19, losses.py:76-80, approaches/mtranse.py:84-96):
 *   loss = alpha * (sum_n ||e2_n - e1_n M||^2 + ||M M^T - I||_F^2),  e = l2_normalize(ent)[ids] (if ent_l2_norm).
 * M [dim, dim] row-major is updated in place (Adagrad with M_acc, or SGD); the gradients w.r.t. the NORMALISED
 * entity rows are added into ent_grad [n_ent, ld] / ent_touched -- the step workspace's scratch, see
 * oea_step_entity_scratch() -- so that oea_triple_step_phase(..., n_pos = 0, OEA_PHASE_APPLY) finishes the step.
 * work: oea_mapping_workspace_floats(n, ld, dim) floats.  loss_accum += the batch loss. */
size_t oea_mapping_workspace_floats(int64_t n_links, int32_t ld, int32_t dim);
int oea_mapping_step(const float *ent, int32_t ld, int32_t dim, int32_t ent_l2_norm, const int32_t *ids1,
                     const int32_t *ids2, int64_t n, float *M, float *M_acc, float alpha, float lr, int32_t opt_kind,
                     void *ent_grad, void *ent_touched, float *work, double *loss_accum, void *stream);
/* A whole mapping epoch (approaches/mtranse.py:84-96) enqueued by ONE call: `steps` x (oea_mapping_step on the step's n links +
 * the apply phase of the step engine with `cfg`).  batches: device int32 [steps, 2, n].  Single process only (a data-parallel
 * job exchanges the scratch between the two halves of every step). */
int oea_mapping_epoch(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel, int32_t dim,
                      int32_t ld, int32_t ent_l2_norm, const int32_t *batches, int32_t steps, int64_t n, float *M, float *M_acc,
                      float alpha, float lr, int32_t opt_kind, const oea_step_cfg *cfg, void *workspace, float *work,
                      double *mapping_loss_accum, double *step_loss_accum, void *stream);
/* addresses of the entity gradient scratch and its touched flags inside a step workspace */
int oea_step_entity_scratch(void *workspace, int64_t n_ent, int64_t n_rel, int32_t ld, void **ent_grad,
                            void **ent_touched);

/* ---------------------------------------------------------------------------------------
 * Negative sampling -- replaces generate_neg_triples_fast (modules/train/batch.py:89-119).
 * The membership set replaces the python set `all_triples_set`.
 * ------------------------------------------------------------------------------------- */
/* capacity: power of two >= 2*n;  table: uint64[capacity] on the device */
uint64_t oea_tripleset_capacity(int64_t n);
int oea_tripleset_build(const int32_t *triples, int64_t n, uint64_t *table, uint64_t capacity,
                        void *stream);
/* out[p*k + s] = s-th negative of positive p.  candidates = nbr[ent_pos[e]*nbr_k ...] when
 * nbr != NULL (truncated sampling, neighbor.get(e), batch.py:96-97) else entity_list.
 * RNG stream: Philox4x32-10 keyed by seed, counter (p + pos_offset, step, try, draw). */
int oea_sample_negatives(const int32_t *pos, int64_t n_pos, int32_t k, const uint64_t *table,
                         uint64_t capacity, const int32_t *entity_list, int32_t n_ent_list,
                         const int32_t *ent_pos, const int32_t *nbr, int32_t nbr_k, uint64_t seed,
                         uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                         int32_t *err_flag, void *stream);

/* The same kernel fed from a RECORD of the reference's own draws instead of Philox (tests: pins the sampler's algorithm -- rounds,
 * one side per round, distinct draws, true triples removed except in the last round, order of acceptance -- against
 * generate_neg_triples_fast itself, modules/train/batch.py:89-119, which draws with python's Mersenne Twister and cannot be matched
 * draw by draw otherwise).  replay int32 [n_pos, max_try, 1 + k]: per positive and round, np.random.binomial's value (1 = corrupt
 * the head) followed by the POSITIONS in the candidate list of that round's random.sample (as many as were still needed; the rest
 * -1).  err_flag 2: a recorded position does not fit the candidate list. */
int oea_sample_negatives_replay(const int32_t *pos, int64_t n_pos, int32_t k, const uint64_t *table, uint64_t capacity,
                                const int32_t *entity_list, int32_t n_ent_list, const int32_t *ent_pos, const int32_t *nbr,
                                int32_t nbr_k, int32_t max_try, const int32_t *replay, int32_t *out, int32_t *err_flag, void *stream);

/* One launch for a whole (pos_batch1 + pos_batch2) batch of generate_relation_triple_batch
 * (batch.py:36-45): positives [0, n_split) are sampled against side[0] (KG1's triple set,
 * entity list and neighbours), positives [n_split, n_pos) against side[1] (KG2).  Identical
 * draws to two oea_sample_negatives calls with pos_offset and pos_offset + n_split. */
typedef struct oea_sampler_side {
    const uint64_t *table;
    uint64_t capacity;
    const int32_t *entity_list;
    const int32_t *ent_pos;
    const int32_t *nbr;      /* NULL: uniform sampling over entity_list */
    int32_t n_ent_list;
    int32_t nbr_k;
    const uint32_t *filter;  /* NULL, or oea_tripleset_filter_build's bit array over the same triples (round 6) */
    uint64_t filter_bits;
} oea_sampler_side;
/* A "certainly absent" pre-test for the membership probes of the sampler (batch.py:108-110: `set(neg_triples) - pos_triples`): one bit per
 * hash bucket, oea_tripleset_filter_bits(capacity) = 8 x capacity bits (1 MB for the 400,000 triples of a 100K KG: resident in every XCD's
 * L2, where the 8 MB key table is not).  A clear bit answers the probe; a set bit (all present triples, ~5 % of the absent ones) goes on to
 * the key table: the sampler's output is unchanged. */
uint64_t oea_tripleset_filter_bits(uint64_t capacity);
int oea_tripleset_filter_build(const int32_t *triples, int64_t n, uint32_t *filter, uint64_t filter_bits, void *stream);
int oea_sample_negatives_pair(const int32_t *pos, int64_t n_pos, int64_t n_split, int32_t k,
                              const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                              uint32_t step, uint32_t pos_offset, int32_t max_try, int32_t *out,
                              int32_t *err_flag, void *stream);

/* Negatives of EVERY batch of an epoch in one launch: pos_all holds the batches back to back, batch s is
 * rows [offsets[s], offsets[s+1]) with its first splits[s] rows from KG1 (offsets_dev: int64 [steps+1],
 * splits_dev: int64 [steps], both on the DEVICE).  Draws are identical to `steps` calls of
 * oea_sample_negatives_pair with step = step_base + s (the sampler does not depend on the tables, so an
 * epoch's negatives can be drawn ahead of its optimiser steps). */
int oea_sample_negatives_epoch(const int32_t *pos_all, int64_t n_rows, const int64_t *offsets_dev,
                               const int64_t *splits_dev, int32_t steps, int32_t k, const oea_sampler_side *side0,
                               const oea_sampler_side *side1, uint64_t seed, uint32_t step_base, int32_t max_try,
                               int32_t *out_all, int32_t *err_flag, void *stream);

/* A whole epoch of BasicModel.launch_triple_training_1epo (basic_model.py:222-232) enqueued by
 * ONE call: for step in [0, steps): sample the negatives of batch `step`
 * (oea_sample_negatives_pair with Philox step = step_base + step) and run the fused optimiser
 * step.  pos_all holds the epoch's positive batches back to back; batch `step` is rows
 * [offsets_host[step], offsets_host[step+1]) of it and its first splits_host[step] rows belong
 * to KG1.  k == 0: positive-only losses (MTransE), no sampling.  When offsets_dev / splits_dev (device
 * copies of the two host arrays) are given, neg_buf must hold (total rows)*k triples and the whole
 * epoch is sampled by ONE launch up front (oea_sample_negatives_epoch); otherwise neg_buf holds
 * max_batch*k triples and every step samples its own batch.  side0 == side1 == NULL together with the device
 * layout: neg_buf ALREADY holds the epoch's negatives (drawn by the caller with oea_sample_negatives_epoch, typically
 * on a second stream while the previous epoch was still running).  Nothing is synchronised: the host returns after enqueueing ~3 kernels
 * per step, which removes the per-step host round trip of the reference's feed_dict loop. */
int oea_triple_epoch(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                     int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                     const int64_t *splits_host, int32_t steps, int32_t k, const oea_sampler_side *side0,
                     const oea_sampler_side *side1, uint64_t seed, uint32_t step_base, int32_t *neg_buf,
                     int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                     const int64_t *offsets_dev, const int64_t *splits_dev, void *stream);
/* The same for steps [step_begin, step_end) of the epoch only (a caller that stops or resumes inside an epoch -- e.g.
 * a benchmark timing K steps -- still enqueues them with one call).  step_base is the Philox step of the epoch's
 * step 0.  A range with step_begin == 0 draws the whole epoch's negatives into neg_buf in one launch when the device
 * layout is given; a later range of the same epoch either passes side0 == side1 == NULL (neg_buf still holds them) or
 * both sides (its steps are then drawn again batch by batch -- identical draws). */
int oea_triple_epoch_range(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                           int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                           const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                           const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                           uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                           void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                           void *stream);

/* An epoch's shuffle and batch layout in one call (basic_model.py:234-235: random.shuffle of both KGs' relation-triple lists;
 * modules/train/batch.py:17-22: batch s = KG1's slice s then KG2's slice s): triples int32 [n1 + n2, 3] = list 1 then list 2 (never
 * modified), slot int64 [n_slots] = the fixed map "position in the epoch's batch layout -> position in cat(list1, list2)"; a fresh
 * uniform permutation of each list (Philox4x32-10 keys of (index, epoch) under `seed`, one stable radix sort) is applied under the
 * map: dall[j] = triples[perm[slot[j]]].  No allocation, no host read (side stream). */
size_t oea_epoch_layout_bytes(int64_t n);
int oea_epoch_layout(const int32_t *triples, int64_t n1, int64_t n2, const int64_t *slot, int64_t n_slots, uint64_t seed, uint32_t epoch,
                     int32_t *dall, void *workspace, size_t ws_bytes, void *stream);

/* ---- the gathered-sum plan of an epoch (round 6; csrc/step_plan.h) ------------------------------------------------------------
 * Replaces, for the rows of the positives' own heads and tails, the scatter-add TF performs on the gradient of
 * tf.nn.embedding_lookup (IndexedSlices -> unsorted_segment_sum, models/basic_model.py:89-98, modules/base/optimizers.py:4-7) --
 * which the step kernels otherwise do with fp32 atomics that execute memory-side on gfx950 (~20 G requests/s: 24 of 54 us of the
 * scoring kernel at the EN-FR-100K shape).  The epoch's positives and the negatives drawn ahead for them say BEFORE the epoch runs
 * which entity row receives which positive's gradient rows: oea_step_plan_build sorts those references by (step, row) once per
 * epoch (stable: batch order inside a row).  With a plan, the scoring kernel writes two rows per positive with plain stores and the
 * optimiser kernel (apply_rows_plan) gathers every listed row's sum in the plan's order -- deterministic for those rows -- and
 * visits exactly the rows that received gradient; relation rows, the corrupted rows of ACTIVE negatives and positives outside the
 * rule (negatives on mixed sides, entries that are no corruption of their positive) keep the atomic scratch and the flag-driven
 * optimiser pass.  Result = oea_triple_epoch_range up to the order of fp32 additions.
 *   oea_step_plan_supported  1 when an epoch under `cfg` would use a plan (TransE score, limited loss, both tables normalised,
 *                            1 <= k <= 10 == cfg->neg_group_k, ld <= 256, SGD / Adagrad, fp32 scratch, tables + state larger than
 *                            128 MB -- cache-resident tables keep the flag-driven optimiser pass, which is faster there --,
 *                            OEA_STEP_PLAN != 0; OEA_STEP_PLAN=2 drops the size condition)
 *   oea_step_plan_bytes      workspace of a plan for n_total positives in `steps` batches of at most max_batch rows (includes the
 *                            2 * max_batch contribution rows)
 *   oea_step_plan_build      pos_all [n_total, 3] in batch order, neg_all [n_total * k, 3] (oea_sample_negatives_epoch), offsets_dev
 *                            [steps + 1] int64 on the device; a few launches + three device primitives on `stream`, no allocation,
 *                            no host read: meant for a side stream behind the previous epoch
 *   oea_triple_epoch_range_plan  = oea_triple_epoch_range with the plan workspace; plan_built = 0: the call builds the plan itself
 *                            (on `stream`, after drawing the negatives when it draws them); plan == NULL or an unsupported
 *                            configuration: exactly oea_triple_epoch_range. */
int32_t oea_step_plan_supported(const oea_step_cfg *cfg, int64_t n_ent, int64_t n_rel, int32_t ld, int32_t k);
size_t oea_step_plan_bytes(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld);
int oea_step_plan_build(const int32_t *pos_all, const int32_t *neg_all, int32_t k, const int64_t *offsets_dev, int64_t n_total,
                        int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, void *plan, size_t plan_bytes, void *stream);
/* byte offsets inside a plan workspace (for tests that hold a built plan to the oracle's restatement): out[0] sorted entry values
 * (uint32: sign << 31 | slot), [1] distinct keys (uint64: step << row_bits | row), [2] first entry of every distinct key (uint32),
 * [3] number of distinct keys (int32), [4] first distinct key of every step (int32 [steps + 1]), [5] contribution rows, [6] row_bits,
 * [7] total bytes, [8] per-positive hub bits (uint32 [n_total]: bit 0 / 1 = the positive's head / tail row has more than 8
 * references in its step and takes this positive's gradient through the atomic scratch instead); out has 9 elements */
int oea_step_plan_offsets(int64_t n_total, int32_t steps, int64_t max_batch, int64_t n_ent, int32_t ld, int64_t *out);
int oea_triple_epoch_range_plan(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                                int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                                uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                                void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                                void *plan, size_t plan_bytes, int32_t plan_built, void *stream);

/* The same for ONE RANK of a job whose ranks take contiguous shares of every batch (rows [n rank / world, n (rank + 1) / world)
 * of the batch's n rows, as models/dist.py:shard_batch) and train on LOCAL copies of the tables between two exchanges
 * (`dp_exchange = 'epoch'`: no collective inside the epoch; the caller reconciles the copies at the epoch's end).  The
 * negatives keep the single-process Philox streams (position inside the batch), so the union of the ranks' draws is the
 * single-process draw.  rank = 0, world = 1: oea_triple_epoch_range. */
int oea_triple_epoch_range_shard(float *ent, float *ent_acc, int64_t n_ent, float *rel, float *rel_acc, int64_t n_rel,
                                 int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                 const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                 const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed,
                                 uint32_t step_base, int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg,
                                 void *workspace, double *loss_accum, const int64_t *offsets_dev, const int64_t *splits_dev,
                                 int32_t rank, int32_t world, void *stream);

/* The steps [step_begin, step_end) of a data-parallel epoch under the partition from ONE call, over the C ABI's own
 * communicator (oea_comm_*, declared below): per step GRAD on this rank's share of the batch (rows nb*rank/world ..
 * nb*(rank+1)/world of batch s, as oea_triple_epoch_range_shard) -> oea_part_pack -> reduce-scatter + relation all-reduce ->
 * oea_part_apply (+ oea_step_apply_normals for TransH) -> all-gather -> oea_part_unpack, all enqueued on `stream` with no host
 * work in between.  Buffers as above: send [world * rpr * (ld + 1)], own [rpr * (ld + 1)], rel_x [n_rel * (ld + 1)], upd
 * [rpr, ld], all [world, rpr, ld]; acc_own [rpr, ld] (Adagrad).  The other arguments as oea_triple_epoch_range. */
struct oea_comm;
int oea_triple_epoch_range_comm(struct oea_comm *comm, float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc,
                                int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed, uint32_t step_base,
                                int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                                const int64_t *offsets_dev, const int64_t *splits_dev, void *send, void *own, void *rel_x,
                                float *upd, float *all, void *stream);


/* Negative LINKS of AliNet.generate_input_batch (approaches/alinet.py:988-1006), drawn on the device.
 *   uniform   (nbr1 == NULL): pair q = round * n_pos + i, round < k:  (ents1[pi1_round(i)], ents2[pi2_round(i)])
 *             -- zip(random.sample(ents1, n_pos), random.sample(ents2, n_pos)) per round; needs n_pos <= n1, n2;
 *   truncated (nbr1 != NULL): pair q = link * 2k + slot:  slot < k: (e1, nbr1[row1[e1]][pi(slot)]),
 *             else (nbr2[row2[e2]][pi'(slot - k)], e2) -- random.sample(neighbors[e], k); needs k <= nbr_k;
 *             nbr1/nbr2: int32 [rows, nbr_k] neighbour entity ids, row1/row2: entity id -> row.
 * pi = keyed pseudo-random permutation (oea_perm_index; key from Philox4x32-10(seed; round | link, step)), i.e. draws
 * without replacement.  out_pairs int32 [m, 2] holds every draw, out_valid fp32 [m] is 1 for the pairs of
 * set(pairs) - exclude (first of equal pairs; exclude = oea_tripleset_build over (e1, 0, e2) triples, or NULL), 0 for
 * the rest.  m = k * n_pos (uniform) or 2 * k * n_pos.  scratch_keys / scratch_vals: uint64 / int32 [scratch_cap],
 * scratch_cap a power of two >= 2 m.  Errors mirror random.sample ("Sample larger than population"). */
uint32_t oea_perm_index(uint32_t i, uint32_t n, uint32_t key);
int oea_sample_link_negatives(const int32_t *pos_links, int64_t n_pos, int32_t k, const int32_t *ents1, int32_t n1,
                              const int32_t *ents2, int32_t n2, const int32_t *nbr1, const int32_t *row1,
                              const int32_t *nbr2, const int32_t *row2, int32_t nbr_k, const uint64_t *exclude,
                              uint64_t exclude_cap, uint64_t seed, uint32_t step, int32_t *out_pairs, float *out_valid,
                              uint64_t *scratch_keys, int32_t *scratch_vals, uint64_t scratch_cap, void *stream);

/* HOST arrays (the only entry point that takes host pointers besides oea_store_load/save_host): one-to-one
 * selection among candidate pairs by descending weight (ties: smaller left, then smaller right id), standing in for
 * the heuristic matching of modules/bootstrapping/alignment_finder.py:83-112.  selected[e] = 1 for the kept edges. */
int oea_greedy_matching(const int32_t *left, const int32_t *right, const float *weight, int64_t n_edges,
                        uint8_t *selected);
/* out[i] = <e1[ii[i]], e2[jj[i]]> over the first dim columns (device arrays): similarities of listed pairs
 * (sim_mat[x, y] lookups of the bootstrapping code). */
int oea_pair_dots(const float *e1, int32_t ld1, const float *e2, int32_t ld2, int32_t dim, const int32_t *ii,
                  const int32_t *jj, int64_t n, float *out, void *stream);

/* The k best (largest != 0: largest values; else smallest) of every row of a SHORT candidate matrix vals [n_rows, ld] (first nc <= 1,024
 * columns) by ranking, ties to the earlier column: out_sel int32 [n_rows, k] = the selected columns in ascending column order,
 * mapped through ids [n_rows, ld_ids] when given (NULL: the columns themselves); out_kth [n_rows] = the k-th best value.  Either
 * output may be NULL.  What np.argsort / np.partition do on the [t, k + margin] candidate lists of RDGCN's get_neg
 * (approaches/rdgcn.py:75-87) and of calculate_nearest_k (modules/finding/similarity.py:80-83) after the grid prefilter. */
int oea_row_rank_select_f32(const float *vals, int64_t n_rows, int32_t nc, int64_t ld, int32_t k, int32_t largest, const int32_t *ids,
                            int64_t ld_ids, int32_t *out_sel, float *out_kth, void *stream);
int oea_row_rank_select_f64(const double *vals, int64_t n_rows, int32_t nc, int64_t ld, int32_t k, int32_t largest, const int32_t *ids,
                            int64_t ld_ids, int32_t *out_sel, double *out_kth, void *stream);

/* ---------------------------------------------------------------------------------------
 * Neighbour search -- replaces find_neighbours (modules/train/batch.py:157-165):
 * np.matmul(sub_embed, embed.T) + per-row np.argpartition(-row, k)[:k].
 * out_idx [nq, k]: the k columns with the largest inner product per query, selected by
 * (value desc, column asc), listed in ascending column order.  If id_map != NULL the
 * output holds id_map[col] (entity_list[neighbors_index], batch.py:163).
 * workspace: oea_topk_workspace_bytes(nq, nc) bytes (may be smaller: the call processes
 * query rows in chunks that fit `ws_bytes`; minimum one 128-row strip).
 * ------------------------------------------------------------------------------------- */
size_t oea_topk_workspace_bytes(int64_t nq, int64_t nc);
/* queries == candidates (q == c, the truncated-sampling refresh of one KG's entities against themselves): with a
 * workspace of this many bytes (0: shape not covered) oea_topk_inner computes only the tiles on and above the diagonal of
 * S = E E^T and feeds rows and columns from them (bit-identical result: S_ij == S_ji in the k-ordered fmaf chain).
 * Round 4: that sweep multiplies the bf16 hi / lo split of the rows (approximate survivors, the neighbourhood of the k-th value
 * decided by exact chains: the same neighbour sets) and collects the survivors in per-wave record streams that a second kernel
 * deals to per-row lists; OEA_TOPK_BF16=0 / OEA_TOPK_STREAM=0 (read once per process) select the fp32 sweep / the per-row
 * segment lists of round 3. */
size_t oea_topk_sym_workspace_bytes(int64_t n, int32_t k);
int oea_topk_inner(const float *q, int64_t nq, int32_t ldq, const float *c, int64_t nc, int32_t ldc,
                   int32_t dim, int32_t k, const int32_t *id_map, int32_t *out_idx, void *workspace,
                   size_t ws_bytes, void *stream);

/* The selection half alone, on a similarity strip that already exists: out_idx[i, :] = the k columns
 * with the largest s[i, :] ((value desc, column asc), ascending column order).  RDGCN's hard-negative
 * mining (approaches/rdgcn.py:75-87: cdist cityblock + argsort()[0:k]) = oea_sim_matrix(manhattan)
 * + this call.  s 16-byte aligned, ld % 4 == 0 (rows are read 16 bytes at a time; the pad columns
 * nc..ld-1 are never selected whatever they hold). */
int oea_topk_rows(const float *s, int64_t n_rows, int64_t nc, int64_t ld, int32_t k, const int32_t *id_map,
                  int32_t *out_idx, void *stream);

/* ---------------------------------------------------------------------------------------
 * Alignment evaluation -- replaces sim() + calculate_rank() of greedy_alignment
 * (modules/finding/similarity.py:11-83, modules/finding/alignment.py:13-84,146-168).
 * Gold of row i is column gold_offset + i (gold_offset = 0 in the reference; a rank that owns
 * query rows [lo, hi) of a row-sharded evaluation passes gold_offset = lo and its csls_r slice).
 *   rank[i]   = #{j != i : S_ij > S_ii  or (S_ij == S_ii and j < i)}     (0-based)
 *   argmax[i] = smallest j maximising S_ij                               (rank[0])
 * metric: OEA_METRIC_*; csls_r / csls_c: per-row / per-column top-k means (NULL = no CSLS),
 * S'_ij = (2*S_ij - r_i) - c_j (similarity.py:74-76).
 * rank / argmax: int32 [n1].  workspace: oea_rank_workspace_bytes(n1) bytes.
 * Inner-product tiles (this call, oea_sim_matrix, oea_topk_inner): both operands are first copied into two
 * process-wide, grow-only scratch buffers in the packed layout the LDS-DMA staging reads ([n_pad, Kp], see
 * DESIGN.md "Packed similarity operands"); reuse is ordered by `stream` (a call on another stream waits for
 * the previous use).  OEA_TILE_GLDS=0 in the environment selects the register-staged tiles (same bits, no scratch).
 * ------------------------------------------------------------------------------------- */
enum { OEA_METRIC_INNER = 0, OEA_METRIC_MANHATTAN = 1, OEA_METRIC_EUCLIDEAN = 2,
       OEA_METRIC_MANHATTAN_F32 = 3 /* oea_sim_matrix only: 1 - sum |a - b| accumulated in fp32 -- a ranking pre-filter, not scipy's bits */ };
size_t oea_rank_workspace_bytes(int64_t n1);
int oea_rank_eval(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2,
                  int32_t dim, int32_t metric, const float *csls_r, const float *csls_c, int64_t gold_offset,
                  int32_t *rank, int32_t *argmax, void *workspace, void *stream);
/* integer reductions of rank[]: hits[i] = #{rank < top_k[i]}, rank_sum = sum(rank+1) (int64),
 * rr_sum = sum 1/(rank+1) (double, fixed summation order).  alignment.py:163-168. */
/* Inner-product evaluation through a CERTIFIED bf16 prefilter (round 4): the tile sweep multiplies bf16 splits of the operands
 * (x = hi + lo: hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, 3/16 of the fp32 matrix time), counts the candidates that are
 * greater than the gold value beyond the error bound, and RECORDS the (query, candidate) pairs inside the bound of the gold value or
 * of the row's running maximum; a second kernel decides the records with the exact k-ordered fmaf chain.  rank / argmax are
 * those of oea_rank_eval(OEA_METRIC_INNER) bit for bit.  status int32[2] (device): [0] != 0 = the record buffer overflowed,
 * the results are INVALID and the caller must take oea_rank_eval; [1] = records written.  Workspace:
 * oea_rank_eval_bf16_workspace_bytes(n1, dim).  No CSLS terms here (oea_rank_eval_metrics_bf16 takes them). */
size_t oea_rank_eval_bf16_workspace_bytes(int64_t n1, int32_t dim);
int oea_rank_eval_bf16(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                       int64_t gold_offset, int32_t *rank, int32_t *argmax, int32_t *status, void *workspace, void *stream);
/* oea_rank_eval_bf16 with the CSLS means (csls_r [n1] of the query rows, csls_c [n2]; both NULL = oea_rank_eval_bf16): what a rank
 * of a row-sharded evaluation calls for its block of query rows (gold_offset = its first row). */
int oea_rank_eval_bf16_csls(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                            const float *csls_r, const float *csls_c, int64_t gold_offset, int32_t *rank, int32_t *argmax,
                            int32_t *status, void *workspace, void *stream);
/* the same + argmax + the metrics of oea_rank_metrics in one go: out int64 [nk + 4] (device) = hits[nk], sum(rank + 1), the bits of
 * the double sum 1 / (rank + 1) (the reduction order of oea_rank_metrics), overflow flag (!= 0: INVALID, take
 * oea_rank_eval_metrics), records written -- one device-to-host copy brings results and status back.  csls_r / csls_c (both or
 * neither): the CSLS means; the values ranked are then (2 s - csls_r[i]) - csls_c[j] as in oea_rank_eval. */
int oea_rank_eval_metrics_bf16(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                               const float *csls_r, const float *csls_c, int64_t gold_offset, const int32_t *top_k_host, int32_t nk,
                               int32_t *rank, int32_t *argmax, int64_t *out_dev, void *workspace, void *stream);
/* the prefilter's approximate similarities out[i, j] ~ <e1[i], e2[j]> (tests of the error bound, timing) */
int oea_sim_bf16_matrix(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim, float *out,
                        int64_t ld_out, void *stream);
/* The inner-product evaluation in TWO launches (the 10,500^2 call spent a third of its time in eight small
 * launches around the 0.20 ms sweep): a prologue that packs both operands, computes the gold similarities and clears the
 * merge buffers, and the tile sweep, whose last workgroup extracts argmax and reduces the metrics in a fixed order.
 * hits_and_rank_sum_dev: int64 [nk + 1] (Hits@top_k[i] counts, then sum(rank + 1)); rr_sum_dev: double sum 1 / (rank + 1).
 * workspace: oea_rank_eval_metrics_workspace_bytes(n1).  Same ranks / argmax as oea_rank_eval + oea_rank_metrics. */
size_t oea_rank_eval_metrics_workspace_bytes(int64_t n1);
int oea_rank_eval_metrics(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim,
                          const float *csls_r, const float *csls_c, int64_t gold_offset, const int32_t *top_k_host, int32_t nk,
                          int32_t *rank, int32_t *argmax, int64_t *hits_and_rank_sum_dev, double *rr_sum_dev, void *workspace,
                          void *stream);
int oea_rank_metrics(const int32_t *rank, int64_t n, const int32_t *top_k_host, int32_t nk,
                     int64_t *hits_dev, int64_t *rank_sum_dev, double *rr_sum_dev, void *stream);

/* calculate_rank (alignment.py:146-168) on an explicit block of a similarity matrix s [n_rows, ld]
 * (device): gold of row i is column gold_idx[i] (0 <= gold_idx[i] < nc).  rank[i] = position of the
 * gold in the descending order of row i with the tie rule above, argmax[i] = smallest column of the
 * row maximum. */
int oea_rank_rows(const float *s, int64_t n_rows, int64_t nc, int64_t ld, const int32_t *gold_idx,
                  int32_t *rank, int32_t *argmax, void *stream);

/* Similarity strip S[i, j] for i in [0,n1), j in [0,n2): out is [n1, ld_out] fp32
 * (similarity.py:34-48).  Used by the sim() mirror, CSLS and the neighbour search. */
int oea_sim_matrix(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2,
                   int32_t ld2, int32_t dim, int32_t metric, float *out, int64_t ld_out,
                   void *stream);
/* Fixed-point L1 pre-filter (RDGCN's hard-negative mining, approaches/rdgcn.py:75-87; scipy cdist 'cityblock' in the reference).
 * oea_quantize_rows_u16: dst[i, k] = clamp(round((src[i, k] - lo) * inv_step), 0, 65535) for k < dim, 0 for dim <= k < ldq
 * (ldq: a multiple of 8).  oea_l1_u16_strip: out[i, j] = -(float) sum_k |q[i, k] - c[j, k]| (integer sums; larger = nearer, the
 * order oea_topk_rows selects; sums >= 2^24 are rounded to float).  step * sum is within dim * step of the true L1 distance when
 * [lo, lo + 65535 step] covers the table: the caller certifies its candidate lists with that bound and re-ranks them with
 * oea_pair_l1_f64. */
int oea_quantize_rows_u16(const float *src, int64_t n, int32_t ld, int32_t dim, float lo, float inv_step, uint16_t *dst,
                          int32_t ldq, void *stream);
int oea_l1_u16_strip(const uint16_t *q, int64_t nq, const uint16_t *c, int64_t nc, int32_t ldq, float *out, int64_t ld_out,
                     void *stream);
/* Manhattan evaluation of a block of query rows from their grid distances (strip [rows, ld] = oea_l1_u16_strip of rows
 * row0 .. row0 + rows of e1 against all nc rows of e2; step = the grid's step, err = the bound on |distance - G step|):
 * rank[row0 + r] / argmax[row0 + r] as oea_rank_eval(OEA_METRIC_MANHATTAN) gives them -- candidates whose grid distance
 * leaves no doubt are counted from the strip, the others by their exact similarity (sequential fp64 chain).
 * gold of row i = column gold_offset + i.  n_exact_rows (may be NULL): incremented for every row whose candidate lists
 * overflowed and which therefore evaluated every pair exactly inside the kernel -- many of them mean the table's range makes
 * the grid's error bound useless and the all-pairs kernel (oea_rank_eval) is the faster path. */
int oea_rank_l1_grid_rows(const float *strip, int64_t rows, int64_t row0, int64_t nc, int64_t ld, const float *e1, int32_t ld1,
                          const float *e2, int32_t ld2, int32_t dim, int64_t gold_offset, float step, float err, int32_t *rank,
                          int32_t *argmax, int32_t *n_exact_rows, void *stream);
/* The same with CSLS (models/basic_model.py:132-135: test() evaluates a second time with csls = args.csls; GCN-Align and RDGCN
 * do so under the manhattan metric): rank / argmax of v_ij = (2 s_ij - csls_r[i]) - csls_c[j] as oea_rank_eval gives them with
 * the same means.  csls_r [n1], csls_c [nc]. */
int oea_rank_l1_grid_rows_csls(const float *strip, int64_t rows, int64_t row0, int64_t nc, int64_t ld, const float *e1, int32_t ld1,
                               const float *e2, int32_t ld2, int32_t dim, int64_t gold_offset, float step, float err,
                               const float *csls_r, const float *csls_c, int32_t *rank, int32_t *argmax, int32_t *n_exact_rows,
                               void *stream);
/* exact manhattan SIMILARITIES of a candidate list: out[i, j] = float(1 - sum_k |q[i, k] - table[cand[i, j], k]|) with the
 * sequential fp64 chain of oea_sim_matrix(OEA_METRIC_MANHATTAN) (k ascending: scipy cdist's bits) -- the values the CSLS means
 * of the manhattan metric are sums of (modules/finding/similarity.py:46-48,57-83). */
int oea_pair_l1_sim(const float *q, int64_t nq, int32_t ldq, const float *table, int64_t n, int32_t ldt, int32_t dim,
                    const int32_t *cand, int32_t c, float *out, void *stream);
/* exact fp64 L1 distances of a candidate list: out[i, j] = sum_k |q[i, k] - table[cand[i, j], k]| (fixed summation order).
 * With OEA_METRIC_MANHATTAN_F32 + oea_topk_rows this is RDGCN's hard-negative mining (approaches/rdgcn.py:75-87) without the
 * fp64 distance of every (seed, entity) pair: fp32 ranks k + margin candidates, these are re-ranked exactly. */
int oea_pair_l1_f64(const float *q, int64_t nq, int32_t ldq, const float *table, int64_t n, int32_t ldt, int32_t dim,
                    const int32_t *cand, int32_t c, double *out, void *stream);

/* out[i] = mean of the k largest of S[i, 0:n2] (calculate_nearest_k, similarity.py:80-83),
 * summed in descending order in fp32.  k <= 64. */
int oea_row_topk_mean(const float *s, int64_t n1, int64_t n2, int64_t ld, int32_t k, float *out,
                      void *stream);
/* CSLS means in one sweep (calculate_nearest_k of S and of S^T, similarity.py:60-65,80-83) for the inner-product metric,
 * without S in HBM: r_out[i] = mean of the k largest of row i of e1 e2^T, c_out[j] = the same for column j -- equal to
 * oea_row_topk_mean on the strips of S and S^T bit for bit.  n1, n2 >= 4096 and k <= 32 (otherwise OEA_EUNSUPPORTED and
 * the caller takes the strip route); workspace: oea_csls_means_workspace_bytes (0 = shape not covered). */
size_t oea_csls_means_workspace_bytes(int64_t n1, int64_t n2, int32_t k);
int oea_csls_means(const float *e1, int64_t n1, int32_t ld1, const float *e2, int64_t n2, int32_t ld2, int32_t dim, int32_t k,
                   float *r_out, float *c_out, void *workspace, size_t ws_bytes, void *stream);
/* in place S'_ij = (2*S_ij - r_i) - c_j (csls_sim, similarity.py:74-76) */
int oea_csls_apply(float *s, int64_t n1, int64_t n2, int64_t ld, const float *r, const float *c,
                   void *stream);

/* ---------------------------------------------------------------------------------------
 * Neighbour aggregation -- replaces tf.sparse_tensor_dense_matmul(A, X)
 * (approaches/gcn_align.py:83,259; alinet.py:581; rdgcn.py:187,196) and its gradient.
 * CSR: rowptr int32 [n_rows+1], colidx int32 [nnz], vals fp32 [nnz].
 *   y[i, :] = act( sum_e vals[e] * x[colidx[e], :] )      act: 0 none, 1 relu
 * The backward pass is the same call on the transposed CSR.  mask_from (may be NULL):
 * y = y * (mask_from > 0) -- the relu gradient gate fused into the backward aggregate.
 * ------------------------------------------------------------------------------------- */
/* Optional work split for hub rows (degrees are power-law; ids are frequency-ordered, read.py:64-79):
 * rows with more than `threshold` (>= 96) nonzeros are cut on the host into chunks that different
 * workgroups sum and add atomically; `rows` lists those rows (zeroed before, activation / mask applied
 * after).  NULL = no splitting (a hub row is then summed by one workgroup). */
typedef struct oea_csr_split {
    const int32_t *chunk_row;   /* [n_chunks] row of each chunk */
    const int32_t *chunk_e0;    /* [n_chunks] first nonzero */
    const int32_t *chunk_e1;    /* [n_chunks] one past the last nonzero */
    const int32_t *rows;        /* [n_rows]   the split rows */
    int32_t n_chunks, n_rows, threshold;
    /* optional (NULL / 0: the chunks are added to y with atomics after a zeroing launch): with a buffer of
     * partials_floats >= n_chunks * ldy floats and the chunk range of every split row, the chunks are computed inside
     * the row kernel's launch and summed in chunk order by the epilogue -- two launches, deterministic */
    const int32_t *row_chunk0;  /* [n_rows + 1] first chunk of each split row */
    float *partials;
    int64_t partials_floats;
    /* optional, with `partials`: [n_rows] zero-initialised tickets -- the last chunk of a row to finish adds the row's
     * chunks in chunk order and stores the row inside the SAME launch (no epilogue launch; the tickets reset themselves) */
    uint32_t *tickets;
} oea_csr_split;
int oea_spmm_csr(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                 const float *x, int32_t dim, int32_t ldx, int32_t act, const float *mask_from,
                 float *y, int32_t ldy, const oea_csr_split *split, void *stream);

/* ---------------------------------------------------------------------------------------
 * Sparse graph attention -- replaces tf.nn.leaky_relu(values) -> tf.sparse_softmax ->
 * tf.sparse_tensor_dense_matmul and their gradients (approaches/alinet.py:661-676,
 * rdgcn.py:202-215).  Segment s owns edges [seg_ptr[s], seg_ptr[s+1]) (the entries normalised
 * together -- whole rows, or TF1's consecutive runs) and adds its aggregate to output
 * row seg_row[s].  z: per-edge pre-activation logits; v: [n_cols, ld] values.
 *   alpha_e = softmax_seg(leaky_relu(z_e)) ;  out[seg_row[s]] += sum_e alpha_e * v[colidx[e]]
 * Backward: dz[e] and dv [n_cols, ld] from dout.
 * ------------------------------------------------------------------------------------- */
typedef struct oea_attn_graph {
    /* segments are cut into sub-segments of bounded length on the host (power-law degrees: a hub
     * row must not become one wave's serial loop); sub-segments of a segment are consecutive */
    const int32_t *sub_ptr;      /* [n_sub+1] edge range of each sub-segment */
    const int32_t *sub_seg;      /* [n_sub]   segment a sub-segment belongs to */
    const int32_t *seg_sub_ptr;  /* [n_seg+1] sub-segment range of each segment */
    const int32_t *seg_row;      /* [n_seg]   output row of each segment */
    const int32_t *colidx;       /* [nnz]     value row (column) of each edge */
    int64_t n_sub, n_seg, nnz;   /* nnz = sub_ptr[n_sub]: the number of edges */
    /* the aggregate out = P . v as a CSR over the OUTPUT rows (fixed summation order: slot order inside a row) */
    const int32_t *agg_rowptr;   /* [agg_rows+1] slot range of each output row */
    const int32_t *agg_colidx;   /* [nnz] value row of each slot */
    const int32_t *agg_edge;     /* [nnz] edge id of each slot (index into alpha); NULL: slot == edge (canonical order) */
    int64_t agg_rows;
    const oea_csr_split *agg_split;   /* hub rows of the aggregate (relative to agg_row0), or NULL */
    /* the transposed CSR for dv = P^T . dout */
    const int32_t *t_rowptr;     /* [t_rows+1] incoming-slot range of each value row (column) */
    const int32_t *t_row;        /* [nnz] output row of the incoming edge (row of dout) */
    const int32_t *t_edge;       /* [nnz] its edge id (index into alpha) */
    int64_t t_rows;
    const oea_csr_split *t_split;     /* hub columns (relative to t_row0), or NULL */
    /* the ranges a call works on (whole graph: [0, n)); a rank of a row-sharded job passes the whole graph with ITS
     * ranges: a block of segments (sub-segment range [sub0, sub1) = segments [seg0, seg1)), a block of output rows
     * [agg_row0, agg_row1) with slots [agg_slot0, agg_slot1) = agg_rowptr[agg_row0 / agg_row1], a block of columns */
    int64_t sub0, sub1, seg0, seg1;
    int64_t agg_row0, agg_row1, agg_slot0, agg_slot1;
    int64_t t_row0, t_row1, t_slot0, t_slot1;
} oea_attn_graph;
#define OEA_ATTN_ALPHA 1       /* forward: softmax statistics + alpha[e] of the segments [seg0, seg1) */
#define OEA_ATTN_AGGREGATE 2   /* forward: out rows [agg_row0, agg_row1) from alpha of ALL their edges */
#define OEA_ATTN_DZ 1          /* backward: dz[e] of the segments [seg0, seg1) */
#define OEA_ATTN_DV 2          /* backward: dv rows [t_row0, t_row1) */
size_t oea_sparse_attn_workspace_floats(const oea_attn_graph *g);
/* No atomics: results are bit-reproducible.  z, alpha, dz are indexed by edge id; out [agg_rows, ld], dv [t_rows, ld]:
 * every row of the call's range is written (rows without edges get zeros).  `phases` selects what runs (a single
 * process passes both bits; between the phases a sharded job all-gathers alpha / dz). */
int oea_sparse_attn_fwd(const oea_attn_graph *g, const float *z, const float *v, int32_t dim, int32_t ld,
                        float lrelu_slope, float *out, float *alpha, float *workspace, int32_t phases, void *stream);
int oea_sparse_attn_bwd(const oea_attn_graph *g, const float *z, const float *v, const float *alpha,
                        const float *dout, int32_t dim, int32_t ld, float lrelu_slope, float *dz, float *dv,
                        float *workspace, int32_t phases, void *stream);
/* The softmax half of the backward alone, for groups whose values are attached to ANOTHER pattern than the one they were
 * normalised over (the third reading of alinet.py:670-676's tf.sparse_softmax on a non-canonical tensor, SURVEY H3: row softmax
 * of the canonically sorted values, p-th value re-attached to the p-th index as fed): dz holds d alpha per edge on entry (the
 * caller's oea_pair_dots over the as-fed pattern) and d z = alpha (d alpha - sum_group alpha d alpha) lrelu'(z) on exit, for
 * the sub-segments [sub0, sub1).  Same kernels and summation order as oea_sparse_attn_bwd's OEA_ATTN_DZ phase. */
int oea_sparse_attn_dz(const oea_attn_graph *g, const float *z, const float *alpha, float *dz, float lrelu_slope,
                       float *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Row-wise glue of the GNN approaches, fused (csrc/gnn_fused.hip): one pass forward, one pass backward, one wave per
 * row, no atomics.  All tensors fp32 row-major.
 *
 * oea_concat_l2n_*: emb = l2n(concat(l2n(x_0), ..., l2n(x_{k-1}))), k <= 4 blocks [n, lds[i]] with dims[i] columns
 *   (approaches/alinet.py:835-840 / 932-943); out [n, ld_out], ld_out >= sum dims (pad columns get zeros);
 *   inv_blk [n, 4] / inv_all [n]: the inverse norms, kept for the backward.  bwd: dz [n, ld_out] -> dx[i] [n, lds[i]].
 * oea_pair_loss_l2_*: alinet.py:828-850 (compute_loss).  pairs int32 [m, 2], the first n_pos of them positive links:
 *   term_p = ||e_i - e_j||^2 (positive) | balance * weight * relu(margin - ||e_i - e_j||^2) (negative; weight [m - n_pos]
 *   or NULL = 1); the loss is the sum of terms [m] (the caller adds them: fixed order); coef [m] = d term / d ||.||^2.
 *   bwd = oea_pair_grad_rows(norm = 2): rowptr [n + 1] / other [2 m] / slot_pair [2 m] = the pairs' endpoints grouped by
 *   embedding row (slot order = the summation order), gscale = device scalar d L / d loss (NULL = 1); grad [n, ld]: every
 *   row written.  norm = 1: grad[r] = sum coef sign(e_r - e_other) -- the L1 hinge of GCN-Align (oea_align_loss_l1_coef).
 * oea_highway_*: gate = relu(tanh(p)), out = tanh(b' (1 - gate) + a' gate), a' = a gamma + beta, b' = b gamma + beta
 *   (alinet.py:597-622; gamma / beta = the layer's BatchNormalization affine, gamma already divided by sqrt(1 + eps));
 *   bwd: da, db, dp [n, d] and partials [oea_colsum_blocks(n), 2, d] whose sums over dim 0 are d gamma, d beta.
 * oea_bias_tanh_*: y = tanh(x + bias) (alinet.py:583-590); bwd: gx [n, d], partials [oea_colsum_blocks(n), d] -> d bias.
 * ------------------------------------------------------------------------------------- */
int oea_concat_l2n_fwd(const float *const *x, const int32_t *dims, const int32_t *lds, int32_t k, int64_t n, float *out,
                       int32_t ld_out, float *inv_blk, float *inv_all, void *stream);
int oea_concat_l2n_bwd(float *const *dx, const int32_t *dims, const int32_t *lds, int32_t k, int64_t n, const float *z,
                       const float *dz, int32_t ld_out, const float *inv_blk, const float *inv_all, void *stream);
int oea_pair_loss_l2_fwd(const float *emb, int64_t n, int32_t dim, int32_t ld, const int32_t *pairs, int64_t m, int64_t n_pos,
                         const float *weight, float margin, float balance, float *coef, float *terms, void *stream);
int oea_pair_grad_rows(const float *emb, int64_t n, int32_t dim, int32_t ld, const int32_t *rowptr, const int32_t *other,
                       const int32_t *slot_pair, const float *coef, const float *gscale, int32_t norm, float *grad, void *stream);
/* out[s] = sum of vals[order[e]] (order NULL: vals[e]) over e in [seg_ptr[s], seg_ptr[s + 1]) -- one wave per segment, fixed
 * order: the gradient of a gather from few distinct rows (rdgcn.py:202-215: per-relation logits gathered per attention edge) */
int oea_segment_sum_f32(const float *vals, const int32_t *order, const int32_t *seg_ptr, int64_t n_seg, float *out, void *stream);
/* the row-grouped endpoint lists of a pair list for oea_pair_grad_rows: pairs int32 [m, 2] with rows in [0, n_rows) ->
 * rowptr [n_rows + 1], other [2 m] (the partner row of every slot), slot_pair [2 m] (its pair); inside a row the slots keep
 * pair order.  One call, stream-ordered, no host synchronisation (csrc/graph_build.hip: rocPRIM stable sort + scan). */
int oea_pair_rows_build(const int32_t *pairs, int64_t m, int64_t n_rows, int32_t *rowptr, int32_t *other, int32_t *slot_pair,
                        void *stream);
int32_t oea_colsum_blocks(int64_t n);
int oea_highway_fwd(const float *a, const float *b, const float *p, const float *gamma, const float *beta, int64_t n, int32_t d,
                    float *out, void *stream);
int oea_highway_bwd(const float *a, const float *b, const float *p, const float *gamma, const float *beta, const float *out,
                    const float *gout, int64_t n, int32_t d, float *da, float *db, float *dp, float *partials, void *stream);
/* out [k1, ld_out] = A^T B for row-major A [m, lda] (k1 columns used) and B [m, ldb] (k2 columns): the weight gradient
 * dW = X^T dY of the GNN approaches' dense layers (alinet.py:574-582, rdgcn.py:250-256: m = #entities, k <= 500) on
 * v_mfma_f32_32x32x2_f32 (exact fp32 products, fixed summation order: slab order inside a row chunk, then chunk order).
 * k1, k2, lda, ldb multiples of 4, operands 16-byte aligned.  workspace: oea_gemm_tn_workspace_floats(m, k1, k2) floats
 * (partial tiles of the row chunks; 0 when one chunk suffices). */
size_t oea_gemm_tn_workspace_floats(int64_t m, int32_t k1, int32_t k2);
int oea_gemm_tn_f32(const float *a, int32_t lda, int32_t k1, const float *b, int32_t ldb, int32_t k2, int64_t m, float *out,
                    int32_t ld_out, float *workspace, void *stream);
/* The same product cut at its row chunks for a row-sharded job (the weight gradients dW = X^T dY of the GNN approaches' dense
 * layers, approaches/alinet.py:574-582): oea_gemm_tn_plan gives the chunk count and rows per chunk of the single-process call;
 * a rank runs oea_gemm_tn_partial on ITS chunks (slots [chunk_begin, chunk_end) of workspace [chunks][k1 * k2]), the slots are
 * all-gathered, and oea_gemm_tn_reduce adds all of them in chunk order -- the single-process summation order: same bits. */
int oea_gemm_tn_plan(int64_t m, int32_t k1, int32_t k2, int32_t *chunks, int64_t *rows_per_chunk);
int oea_gemm_tn_partial(const float *a, int32_t lda, int32_t k1, const float *b, int32_t ldb, int32_t k2, int64_t m,
                        int32_t chunk_begin, int32_t chunk_end, float *workspace, void *stream);
int oea_gemm_tn_reduce(const float *workspace, int32_t chunks, int32_t k1, int32_t k2, float *out, int32_t ld_out, void *stream);

/* RDGCN's dense glue between its sparse operators, one pass each way (rdgcn.py:184-191, 250-256, 330-333):
 * oea_sigmoid_mix_*: gate = sigmoid(p + bias), out = gate b + (1 - gate) a (highway; p = a W from a library GEMM);
 *   bwd: da, db, dp [n, d] and partials [oea_colsum_blocks(n), d] -> d bias (summed in block order by the caller); b_relu != 0:
 *   b is a relu's output and db is gated by b > 0 (the gradient of the relu's input).
 * oea_relu_axpy_*: out = x + alpha relu(y); bwd: dy = alpha gout where y > 0 (dx = gout).
 * oea_colsum_prod: partials [oea_colsum_blocks(n), d] of the column sums of x * y (d w0 of the diagonal layer). */
int oea_sigmoid_mix_fwd(const float *a, const float *b, const float *p, const float *bias, int64_t n, int32_t d, float *out,
                        void *stream);
int oea_sigmoid_mix_bwd(const float *a, const float *b, const float *p, const float *bias, const float *gout, int64_t n, int32_t d,
                        int32_t b_relu, float *da, float *db, float *dp, float *partials, void *stream);
int oea_relu_axpy_fwd(const float *x, const float *y, float alpha, int64_t total, float *out, void *stream);
int oea_relu_axpy_bwd(const float *y, const float *gout, float alpha, int64_t total, float *dy, void *stream);
int oea_colsum_prod(const float *x, const float *y, int64_t n, int32_t d, float *partials, void *stream);
int oea_bias_tanh_fwd(const float *x, const float *bias, int64_t n, int32_t d, float *y, void *stream);
int oea_bias_tanh_bwd(const float *y, const float *gy, int64_t n, int32_t d, float *gx, float *partials, void *stream);

/* Dense Adam step with tf.train.AdamOptimizer semantics (alinet.py:871, rdgcn.py:332); t = 1-based
 * step count. */
int oea_adam_dense(float *param, const float *grad, float *m, float *v, int64_t n, float lr, float beta1,
                   float beta2, float eps, int64_t t, void *stream);

/* L1 alignment hinge of GCN-Align / RDGCN (approaches/gcn_align.py:298-320,
 * rdgcn.py:293-315): forward + gradient w.r.t. the output embedding table.
 * ILL int32 [t,2]; neg_*: int32 [t*k]; grad [n, ld] must be zero on entry.
 * loss_accum += (sum L1 + sum L2) / (2 k t). */
int oea_align_loss_l1(const float *out_emb, int64_t n, int32_t dim, int32_t ld, const int32_t *ill,
                      int64_t t, int32_t k, float gamma, const int32_t *neg_left,
                      const int32_t *neg_right, const int32_t *neg2_left, const int32_t *neg2_right,
                      float *grad, double *loss_accum, void *stream);

/* The same hinge WITHOUT the gradient: coef_out [t + 2 t k] gets the signed coefficient of every pair -- [0, t): the links,
 * + #active hinges / (2 k t); [t + a 2k + i]: negative i of link a (i < k: (neg_left, neg_right), else (neg2_left,
 * neg2_right)), - 1 / (2 k t) if its hinge is active, else 0 -- and oea_pair_grad_rows(norm = 1) over the pairs' endpoints
 * grouped by row gives the gradient without atomics (reproducible; the choice when the negatives stay for several epochs
 * and k is small: GCN-Align).  grad may be NULL then. */
int oea_align_loss_l1_coef(const float *out_emb, int64_t n, int32_t dim, int32_t ld, const int32_t *ill,
                           int64_t t, int32_t k, float gamma, const int32_t *neg_left,
                           const int32_t *neg_right, const int32_t *neg2_left, const int32_t *neg2_right,
                           float *grad, double *loss_accum, float *coef_out, void *stream);

/* W -= lr * dW where dW is the gradient w.r.t. T = l2_normalize(W) pulled back through the
 * normalisation (gcn_align.py:52-56 + GradientDescentOptimizer, gcn_align.py:511).
 * normalize == 0: plain SGD. */
int oea_sgd_rows(float *w, const float *grad_t, int64_t rows, int32_t dim, int32_t ld,
                 int32_t normalize, float lr, void *stream);

/* One full-batch epoch of a GCN_Align_Unit (approaches/gcn_align.py:498-539: GraphConvolution(relu, trunc_normal weight) ->
 * GraphConvolution(identity, no weight) + align_loss + GradientDescentOptimizer; the loop of :737-785) enqueued by ONE call:
 *   T = l2_normalize(W);  X = T (featureless) | F . T;  H1 = relu(A X);  out = A H1;  hinge -> d out;
 *   d H1 = A^T d out (gated by H1 > 0);  d X = A^T d H1;  d T = d X | F^T d X;  W -= lr * (d T through the normalisation).
 * A / A^T / F / F^T: CSR operands (+ optional hub-row splits); pair_*: the hinge's pair endpoints grouped by row
 * (oea_align_loss_l1_coef + oea_pair_grad_rows(norm 1): no atomics) or NULL (oea_align_loss_l1 with atomics).
 * Buffers [n, ld] (t: [w_rows, ld]; x, g_t only with features; coef [t + 2 t k] only with pair lists) are the caller's
 * and hold the forward / backward intermediates afterwards (out = the unit's output embedding). */
typedef struct oea_gcn_unit {
    const int32_t *a_rowptr, *a_colidx; const float *a_vals; const oea_csr_split *a_split;
    const int32_t *at_rowptr, *at_colidx; const float *at_vals; const oea_csr_split *at_split;
    const int32_t *f_rowptr, *f_colidx; const float *f_vals; const oea_csr_split *f_split;       /* NULL: featureless */
    const int32_t *ft_rowptr, *ft_colidx; const float *ft_vals; const oea_csr_split *ft_split;
    const int32_t *row_ids;          /* [w_rows] = 0 .. w_rows - 1 */
    const int32_t *ill;              /* [t, 2] seed links */
    const int32_t *neg_left, *neg_right, *neg2_left, *neg2_right;      /* [t k] */
    const int32_t *pair_rowptr, *pair_other, *pair_slot;               /* [n + 1], [2 (t + 2 t k)] x 2, or NULL */
    int64_t n, w_rows, t;
    int32_t dim, ld, k;
    float gamma, lr;
} oea_gcn_unit;
typedef struct oea_gcn_unit_buffers {
    float *t, *x, *h1, *out, *g_out, *g_pre1, *g_x, *g_t, *coef;
} oea_gcn_unit_buffers;
int oea_gcn_unit_epoch(const oea_gcn_unit *u, float *w, const oea_gcn_unit_buffers *b, double *loss_accum, void *stream);

/* ---------------------------------------------------------------------------------------
 * Graph builders on the device (csrc/graph_build.hip) -- the per-init adjacency construction of the GNN approaches.
 * Inputs: triples int32 [n, 3] (device).  Outputs are sorted by (row, col); their sizes are data dependent: the caller
 * passes buffers of `cap` entries (cap >= 2 * n_tri + n_ent covers every case) and reads *nnz_dev back.  All sums run in
 * a fixed order (stable sort + segmented reduction): two processes build bit-identical operands.  These calls
 * synchronise the stream (they read intermediate counts back): they run once per model init, not in an epoch.
 * ------------------------------------------------------------------------------------- */
/* approaches/alinet.py:155-181: distinct undirected (h, t) pairs as 0/1 entries, + I, D^-1/2 . D^-1/2
 * (preprocess_adj, alinet.py:114-132).  fp64 values like scipy's. */
int oea_build_unweighted_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int32_t *row, int32_t *col, double *val,
                             int64_t cap, int64_t *nnz_dev, void *stream);
/* approaches/gcn_align.py:610-640 (r2f / r2if [n_rel] fp64 = distinct heads / tails per triple of a relation),
 * :642-664 (M[(h,t)] += max(r2if, .3), M[(t,h)] += max(r2f, .3); entry of key (a, b) at row b, column a; optional raw
 * adjacency outputs adj_*: NULL to skip) and :566-578 (support = normalize_adj(adj + I) = (A' D^-1/2)^T D^-1/2). */
int oea_build_weighted_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int64_t n_rel, double *r2f, double *r2if,
                           int32_t *adj_row, int32_t *adj_col, double *adj_val, int64_t *adj_nnz_dev, int32_t *row,
                           int32_t *col, double *val, int64_t cap, int64_t *nnz_dev, void *stream);
/* approaches/rdgcn.py:45-72 (get_mat + get_sparse_tensor): M[sec, fir] = 1 / sqrt(deg[fir]) / sqrt(deg[sec]) over the
 * symmetric closure of the (h, t) pairs with h != t plus the diagonal, with get_mat's degree rule; values as fp32. */
int oea_build_primal_adj(const int32_t *tri, int64_t n_tri, int64_t n_ent, int32_t *row, int32_t *col, float *val,
                         int64_t cap, int64_t *nnz_dev, void *stream);
/* approaches/rdgcn.py:17-42 + :268-277: out [n_rel, n_rel] fp32 = Jaccard overlap of the relations' head sets + of
 * their tail sets. */
int oea_build_dual_adj(const int32_t *tri, int64_t n_tri, int64_t n_rel, float *out, void *stream);
/* approaches/alinet.py:250-287 (generate_2hop_triples): self-join of `tri` (tail == head, left triple major, matches in
 * table order), rows whose (h, t) is an edge of `full_tri` dropped, the n_cut most frequent (r1, r2) patterns dropped
 * (ties: first appearance in the joined table, i.e. python's stable sort), out = distinct (h, r1 + r2, t) of the kept
 * rows plus (h, 0, h) for their heads, sorted.  stats_host[4] = the reference's four log counts (distinct joined
 * (h, r1, r2, t); patterns; patterns kept; triples out). */
int oea_build_2hop(const int32_t *tri, int64_t n_tri, const int32_t *full_tri, int64_t n_full, int64_t n_ent,
                   int64_t n_rel, int32_t n_cut, int32_t *out, int64_t cap, int64_t *n_out_dev, int64_t *stats_host,
                   void *stream);

/* ---------------------------------------------------------------------------------------
 * Collective group (one process per GPU, RCCL over xGMI) -- no reference counterpart (the reference is single-device,
 * SURVEY F2).  The exchange points of the multi-GPU path for a host that is not Python: reduce-scatter / all-gather of
 * the partitioned step (oea_part_*), int64 / fp64 sums of the row-sharded evaluation's metrics, all-gather of row blocks
 * (argmax, neighbour lists, a graph layer's output rows).  RCCL is dlopen'ed by oea_comm_unique_id / oea_comm_init: a
 * single-GPU process never loads it.  The Python host uses torch.distributed (backend "nccl" = the same RCCL).
 *   rank 0: oea_comm_unique_id(id) -> ship the 128 bytes to the other ranks (any channel) -> every rank, after
 *   hipSetDevice(its GPU): oea_comm_init(id, rank, nranks, &comm).  Calls are asynchronous on `stream`.
 * ------------------------------------------------------------------------------------- */
typedef struct oea_comm *oea_comm_t;
int oea_comm_unique_id(void *id_out_128);
int oea_comm_init(const void *unique_id_128, int32_t rank, int32_t nranks, oea_comm_t *out);
/* A communicator over HOST CALLBACKS instead of RCCL: every collective of the entry points below (and of
 * oea_triple_epoch_range_comm) is handed to `fn`: op = OEA_COMM_*, send / recv = device pointers, count = elements PER RANK
 * (all-gather: elements of every rank's send block; reduce-scatter: elements of every rank's recv block; all-reduce:
 * elements of the buffer, send == recv), dtype = OEA_COMM_F32 / F64 / I64.  The callback must order itself after the work
 * already enqueued on `stream` and leave the result visible to work enqueued after it returns; it returns 0 on success.
 * RCCL needs one GPU per rank: with callbacks over torch.distributed's gloo group the build pool's ONE GPU runs the call path
 * with 2 and 4 ranks (tests/test_partition_gpu.py). */
enum { OEA_COMM_ALLREDUCE = 0, OEA_COMM_ALLGATHER = 1, OEA_COMM_REDUCE_SCATTER = 2 };
enum { OEA_COMM_F32 = 0, OEA_COMM_F64 = 1, OEA_COMM_I64 = 2 };
typedef int (*oea_comm_callback)(void *user, int32_t op, const void *send, void *recv, int64_t count, int32_t dtype, void *stream);
int oea_comm_init_callbacks(int32_t rank, int32_t nranks, oea_comm_callback fn, void *user, oea_comm_t *out);
int oea_comm_destroy(oea_comm_t c);
int32_t oea_comm_rank(oea_comm_t c);
int32_t oea_comm_size(oea_comm_t c);
/* recv [nranks][rows_per_rank, ld] <- every rank's send [rows_per_rank, ld] */
int oea_allgather_rows(oea_comm_t c, const float *send, float *recv, int64_t rows_per_rank, int32_t ld, void *stream);
/* recv [n_per_rank] = sum over ranks of send[rank * n_per_rank ...] (send holds nranks * n_per_rank floats) */
int oea_comm_reduce_scatter_f32(oea_comm_t c, const float *send, float *recv, int64_t n_per_rank, void *stream);
int oea_allreduce_f32(oea_comm_t c, float *buf, int64_t n, void *stream);
int oea_allreduce_f64(oea_comm_t c, double *buf, int64_t n, void *stream);
int oea_allreduce_i64(oea_comm_t c, int64_t *buf, int64_t n, void *stream);
/* the same three collectives for any of the dtypes (the deterministic build exchanges int64 fixed-point gradients) */
int oea_comm_allgather(oea_comm_t c, const void *send, void *recv, int64_t n_per_rank, int32_t dtype, void *stream);
int oea_comm_reduce_scatter(oea_comm_t c, const void *send, void *recv, int64_t n_per_rank, int32_t dtype, void *stream);
int oea_comm_allreduce(oea_comm_t c, void *buf, int64_t n, int32_t dtype, void *stream);
/* All-to-all with per-peer element counts known on the host: peer p receives send[send_displs[p] .. + send_counts[p]) and
 * recv[recv_displs[p] ..] takes recv_counts[p] elements from p.  RCCL: one group of ncclSend / ncclRecv; a callback communicator
 * hands it to the function set by oea_comm_set_alltoallv (same ordering contract as oea_comm_callback). */
typedef int (*oea_comm_alltoallv_callback)(void *user, const void *send, const int64_t *send_counts, const int64_t *send_displs,
                                           void *recv, const int64_t *recv_counts, const int64_t *recv_displs, int32_t dtype, void *stream);
int oea_comm_set_alltoallv(oea_comm_t c, oea_comm_alltoallv_callback fn);
int oea_comm_alltoallv(oea_comm_t c, const void *send, const int64_t *send_counts, const int64_t *send_displs, void *recv,
                       const int64_t *recv_counts, const int64_t *recv_displs, int32_t dtype, void *stream);

/* oea_triple_epoch_range_comm with the BOUNDARY-ROW ("halo") exchange (BASELINE.json north_star: "all-gather of boundary
 * embeddings"): instead of every owned row (reduce-scatter + all-gather of [E, ld] per step) a step moves only the rows its batch
 * refers to.  Every rank derives from the epoch's positives and negatives (drawn ahead, the same Philox streams everywhere) the
 * sorted list of rows each rank's share of each step refers to, per owner (owner = id mod world) -- no index travels and all
 * message sizes are known after ONE host read of the [steps][world][world] counts per call.  Per step: GRAD -> all-to-all of the
 * gradient rows (+ flag) to their owners, added into the owner's scratch (exact in the fixed-point build: the G-rank job is the
 * single-GPU job bit for bit) -> relation rows all-reduced -> optimiser on the owned rows -> all-to-all of the current values of
 * the rows the next step's readers refer to.  The last step of the range ends with the dense all-gather of the owned rows
 * (upd / all as above), so outside the call every rank holds the whole table.  TransE / TransH scores, SGD / Adagrad.
 *   halo_ws: oea_halo_workspace_bytes(n_ent, world, step_end - step_begin or more, largest batch, k) device bytes;
 *   buf_a, buf_b: two exchange buffers of oea_halo_buffer_bytes(...) device bytes each;
 *   stats_host (may be NULL): int64 [4] = bytes pushed (gradient rows sent), bytes pulled (rows received), largest number of
 *   rows sent in a step, steps run. */
size_t oea_halo_workspace_bytes(int64_t n_ent, int32_t world, int32_t steps, int64_t max_batch, int32_t k);
/* the plan alone (what oea_triple_epoch_range_halo computes first): counts_host [step_end - step_begin][world][world] = distinct entity
 * rows rank r's share of step s refers to that rank o owns; lists_host (may be NULL) [steps][world][*cap_out] = the lists (local row
 * indices j, entity id = j * world + o; owner-major, ascending).  neg_all = the epoch's negatives [rows * k, 3] (NULL when k = 0). */
int oea_halo_plan(const int32_t *pos_all, const int32_t *neg_all, int32_t k, const int64_t *offsets_host, const int64_t *offsets_dev,
                  int32_t steps, int32_t step_begin, int32_t step_end, int64_t n_ent, int32_t world, void *halo_ws, size_t halo_ws_bytes,
                  int32_t *counts_host, int32_t *lists_host, int64_t *cap_out, void *stream);
size_t oea_halo_buffer_bytes(int64_t n_ent, int32_t world, int64_t max_batch, int32_t k, int32_t ld);
int oea_triple_epoch_range_halo(oea_comm_t comm, float *ent, float *acc_own, int64_t n_ent, float *rel, float *rel_acc,
                                int64_t n_rel, int32_t dim, int32_t ld, const int32_t *pos_all, const int64_t *offsets_host,
                                const int64_t *splits_host, int32_t steps, int32_t step_begin, int32_t step_end, int32_t k,
                                const oea_sampler_side *side0, const oea_sampler_side *side1, uint64_t seed, uint32_t step_base,
                                int32_t *neg_buf, int32_t *err_flag, const oea_step_cfg *cfg, void *workspace, double *loss_accum,
                                const int64_t *offsets_dev, const int64_t *splits_dev, void *halo_ws, size_t halo_ws_bytes,
                                void *buf_a, void *buf_b, size_t buf_bytes, void *rel_x, float *upd, float *all,
                                int64_t *stats_host, void *stream);

/* Phase times of oea_triple_epoch_range_comm: between _begin and _end every step records HIP events at its phase boundaries
 * on the call's stream; _end waits for the last one and returns the summed milliseconds per phase
 * (GRAD | pack | reduce-scatter + relation all-reduce | apply | all-gather | unpack; for oea_triple_epoch_range_halo: GRAD | pack |
 * gradient all-to-all + relation all-reduce | add + apply | row all-to-all (or the closing all-gather) | unpack) and the number of
 * steps recorded. */
enum { OEA_COMM_PHASES = 6 };
int oea_comm_profile_begin(oea_comm_t c);
int oea_comm_profile_end(oea_comm_t c, double *phase_ms /* [OEA_COMM_PHASES] */, int32_t *steps);

#ifdef __cplusplus
}
#endif
#endif /* OPENEA_HIP_H */
