/*
 * diffsbdd_hip.h -- C-ABI of libdiffsbdd_hip.so: the MI355X (gfx950) kernels for
 * DiffSBDD's DDPM denoising hot path.
 *
 * The reference (/root/reference) is pure Python/PyTorch and has no FFI; the
 * interface replaced here is the Python call boundary of
 *
 *   EGNNDynamics.forward                 equivariant_diffusion/dynamics.py:87-167
 *   EGNNDynamics.get_edges               equivariant_diffusion/dynamics.py:169-187
 *   EGNN / EquivariantBlock / GCL /
 *   EquivariantUpdate .forward           equivariant_diffusion/egnn_new.py:31-244
 *   ConditionalDDPM.sample_p_zs_given_zt equivariant_diffusion/conditional_model.py:432-464
 *   EnVariationalDiffusion
 *     .sample_p_zs_given_zt              equivariant_diffusion/en_diffusion.py:503-557
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *     nothing is allocated, freed or owned by the library except the opaque
 *     engine object (host memory); device workspaces are caller-provided.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued asynchronously, no entry point synchronises.
 *   - all floating point is fp32, row-major; node masks are int64 as in the
 *     reference (constants.py:8-9) and must be sorted ascending (the reference
 *     builds them with repeat_interleave, utils.py:146-154).
 *   - return value: 0 = ok, negative = DSBDD_ERR_*.  Device-side conditions
 *     (NaN in the velocity, edge-capacity overflow) are reported through the
 *     `status` word (bit mask DSBDD_STATUS_*), never by a host sync.
 *
 * The Python binding is diffsbdd_amd/_lib.py (ctypes); INTEGRATION.md shows
 * the stub a reference maintainer would add.
 */
#ifndef DIFFSBDD_HIP_H
#define DIFFSBDD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSBDD_ABI_VERSION 6

enum {
  DSBDD_OK = 0,
  DSBDD_ERR_ARG = -1,         /* bad argument / unsupported configuration   */
  DSBDD_ERR_STATE = -2,       /* weights or workspace not bound             */
  DSBDD_ERR_CAPACITY = -3,    /* sizes exceed the bound workspace           */
  DSBDD_ERR_LAUNCH = -4       /* hip launch error (see dsbdd_last_error)    */
};

enum {
  DSBDD_STATUS_NAN = 1,            /* dynamics.py:155-159                   */
  DSBDD_STATUS_EDGE_OVERFLOW = 2   /* more edges than edge_capacity         */
};

/* EGNNDynamics constructor arguments that reach the kernels
 * (dynamics.py:11-19; lightning_modules.py:137-159). */
typedef struct dsbdd_config {
  int32_t atom_nf, residue_nf, joint_nf, hidden_nf;
  int32_t n_layers, inv_sublayers;
  int32_t attention, use_tanh, update_pocket_coords, reflection_equivariant;
  int32_t edge_embedding_dim;         /* 0 = no edge-type embedding         */
  int32_t has_cutoff_ligand, has_cutoff_pocket, has_cutoff_interaction;
  float cutoff_ligand, cutoff_pocket, cutoff_interaction;
  float norm_constant, normalization_factor, coords_range;
} dsbdd_config;

typedef struct dsbdd_engine dsbdd_engine;

/* ---- weight slots --------------------------------------------------------
 * Weights are passed as a table of device pointers in the order below.  All
 * matrices are stored TRANSPOSED with respect to nn.Linear ([in][out], out
 * contiguous) with the leading dimension padded up to a multiple of 4 floats;
 * vectors are unpadded.  `JP` = round_up(joint_nf+1, 4), `H` = hidden_nf.
 *
 * global slots (DSBDD_G_*), then per block b = 0..n_layers-1:
 *   inv_sublayers x DSBDD_GCL_* slots, then DSBDD_EQ_* slots.            */
enum {
  DSBDD_G_ATOM_ENC_W0T = 0, DSBDD_G_ATOM_ENC_B0, DSBDD_G_ATOM_ENC_W1T, DSBDD_G_ATOM_ENC_B1,
  DSBDD_G_RES_ENC_W0T, DSBDD_G_RES_ENC_B0, DSBDD_G_RES_ENC_W1T, DSBDD_G_RES_ENC_B1,
  DSBDD_G_ATOM_DEC_W0T, DSBDD_G_ATOM_DEC_B0, DSBDD_G_ATOM_DEC_W1T, DSBDD_G_ATOM_DEC_B1,
  DSBDD_G_RES_DEC_W0T, DSBDD_G_RES_DEC_B0, DSBDD_G_RES_DEC_W1T, DSBDD_G_RES_DEC_B1,
  DSBDD_G_EMB_WT,      /* [JP][H]   rows >= joint_nf+1 are zero                */
  DSBDD_G_EMB_B,       /* [H]                                                  */
  DSBDD_G_EMBOUT_WT,   /* [H][JP]   cols >= joint_nf+1 are zero                */
  DSBDD_G_EMBOUT_B,    /* [JP]                                                 */
  DSBDD_G_COUNT
};
enum {
  DSBDD_GCL_E1_WT = 0, /* [H][2H]: edge_mlp.0.weight[:, :H]^T | [:, H:2H]^T     */
  DSBDD_GCL_E1_WD,     /* [H]    : edge_mlp.0.weight[:, 2H]   (current |d|^2)   */
  DSBDD_GCL_E1_WD0,    /* [H]    : edge_mlp.0.weight[:, 2H+1] (input   |d|^2)   */
  DSBDD_GCL_E1_TAB,    /* [3][H] : bias + edge-type-embedding contribution      */
  DSBDD_GCL_E2_WT,     /* [H][H] */
  DSBDD_GCL_E2_B,      /* [H]    */
  DSBDD_GCL_ATT_W,     /* [H]    (unused if !attention)                         */
  DSBDD_GCL_ATT_B,     /* [1]    */
  DSBDD_GCL_N1_WT,     /* [2H][H]: node_mlp.0.weight^T                          */
  DSBDD_GCL_N1_B,      /* [H]    */
  DSBDD_GCL_N2_WT,     /* [H][H] */
  DSBDD_GCL_N2_B,      /* [H]    */
  DSBDD_GCL_COUNT
};
enum {
  DSBDD_EQ_C1_WT = 0,  /* [H][4H]: coord col | cross col | coord row | cross row parts of
                          {coord,cross_product}_mlp.0.weight^T ([H][2H]: coord col |
                          coord row if reflection_equivariant)                  */
  DSBDD_EQ_C_WD,       /* coord_mlp: same five slots as the GCL edge MLP        */
  DSBDD_EQ_C_WD0, DSBDD_EQ_C_TAB, DSBDD_EQ_C_W2T, DSBDD_EQ_C_B2,
  DSBDD_EQ_X_WD,       /* cross_product_mlp (ignored if reflection_equivariant) */
  DSBDD_EQ_X_WD0, DSBDD_EQ_X_TAB, DSBDD_EQ_X_W2T, DSBDD_EQ_X_B2,
  DSBDD_EQ_W3,         /* [H]: the shared bias-free output layer (egnn_new.py:78) */
  DSBDD_EQ_COUNT
};

/* ---- engine ---------------------------------------------------------------*/
int dsbdd_abi_version(void);
const char* dsbdd_last_error(void);

int dsbdd_engine_create(const dsbdd_config* cfg, dsbdd_engine** out);
void dsbdd_engine_destroy(dsbdd_engine* e);
int dsbdd_engine_weight_slots(const dsbdd_engine* e);
int dsbdd_engine_set_weights(dsbdd_engine* e, const float* const* slots_host, int n_slots);

/* Workspace for up to (n_lig, n_pocket) nodes, `batch` samples and
 * `edge_capacity` directed edges (self loops included). */
size_t dsbdd_engine_workspace_bytes(const dsbdd_engine* e, int64_t n_lig, int64_t n_pocket,
                                    int64_t batch, int64_t edge_capacity);
int dsbdd_engine_bind_workspace(dsbdd_engine* e, void* workspace, size_t bytes, int64_t n_lig,
                                int64_t n_pocket, int64_t batch, int64_t edge_capacity);

/* Pocket frame of a pocket-conditioned sampling chain (update_pocket_coords = 0).
 * During such a chain the pocket moves rigidly (the reverse step only translates it with the ligand's
 * centre of mass, conditional_model.py:688-696), so its pocket-pocket radius graph and distances are
 * constants of the chain.  With a frame set, block 0 of dsbdd_dynamics_forward evaluates
 *     agg[node] = [sum over the node's edges with a ligand endpoint] + [sum over its pocket-pocket edges]
 * with the second part taken from the frame: a static edge list built here ONCE from the raw coordinates of the
 * frame's pockets.  The frame holds one REPRESENTATIVE pocket per group of identical pockets of the batch
 * (identical atom features and coordinates, e.g. prepare_pocket(repeats = n_samples), lightning_modules.py:738-750:
 * one group; a batch packed from several pockets: one per pocket; nothing shared: every sample its own):
 *   x_frame    [n_frame][3]  raw coordinates of the representatives' atoms, sample after sample
 *   mask_frame [n_frame]     sample id 0 .. batch_frame-1 of every frame row (sorted)
 *   frame_rows [n_frame]     row of that atom in the call's pocket array (0-based)
 *   twin_local [n_pocket]    frame row that stands for pocket row i (its own group's representative)
 * The pocket-pocket messages of block 0 are the same in every sample of a group (same features, same time step,
 * same distances) and are evaluated once per representative -- 82 % of that stage's edges at the benchmark batch.
 * Every grouping gives bit-identical results: the association A + B and the raw-coordinate distances are the same.
 * The representatives are also the rows of the canonical (ligand-free) pocket network of the FORWARD CONE of the
 * ligand-output-only calls (eps_pocket == NULL, t_count == 1; see dsbdd_dynamics_forward): stage g then computes the
 * rows within min(g + 1, G - g) hops of a ligand atom only, the rest comes from the representative.
 * A frame also fixes the pocket FEATURES of the calls it applies to (the feature columns of xh_pocket never change in
 * pocket-conditioning mode): the residue encoder runs in the first such call only, later calls reuse its output.
 * Set a new frame (or clear it) before calling with other pocket features.
 * The frame lives in the workspace; it is dropped by bind_workspace and by dsbdd_engine_clear_pocket_frame,
 * and only applies to calls with exactly (n_lig, n_pocket, batch).  edge_bound_frame = upper bound on the
 * frame's edge count (sum of squared pocket sizes of the frame samples + 32 per sample).  This function
 * synchronises the stream once (it reads the frame's edge count back). */
int dsbdd_engine_set_pocket_frame(dsbdd_engine* e, void* stream, const float* x_frame, const int64_t* mask_frame,
                                  const int32_t* frame_rows, const int32_t* twin_local, int64_t n_lig,
                                  int64_t n_pocket, int64_t batch, int64_t n_frame, int64_t batch_frame,
                                  int64_t edge_bound_frame);
int dsbdd_engine_clear_pocket_frame(dsbdd_engine* e);

/* Optional per-block trace (debug / parity tests): after every EquivariantBlock
 * the node features h [N][H] and coordinates x [N][3] are copied to
 * trace_h + b*N*H / trace_x + b*N*3.  NULL disables. */
int dsbdd_engine_set_trace(dsbdd_engine* e, float* trace_h, float* trace_x);

/* EGNNDynamics.forward (dynamics.py:87-167).
 *   xh_lig    [n_lig][3+atom_nf]       xh_pocket [n_pocket][3+residue_nf]
 *   t         [t_count] with t_count == batch or 1 (dynamics.py:104-111)
 *   ext_row/ext_col: optional teacher-forced edge list (int32, sorted by
 *     (row,col), node numbering [ligand | pocket]); NULL -> radius graph is
 *     built on device as dynamics.py:169-187 does.
 *   eps_lig   [n_lig][3+atom_nf]       eps_pocket [n_pocket][3+residue_nf] (may be NULL)
 *   status    int32 device word, OR-ed with DSBDD_STATUS_* (caller zeroes it)
 * eps_pocket == NULL in pocket-conditioning mode (what ConditionalDDPM's chains need, conditional_model.py:268-272)
 * lets the engine skip every row the ligand output does not depend on: message stage g of G evaluates the nodes
 * within G - g hops of a ligand atom (csrc/graph.h "Level-ordered edge list"; dsbdd_engine_last_plan reports the
 * radii).  eps_lig is the same up to fp32 summation order (different wave-tile boundaries). */
int dsbdd_dynamics_forward(dsbdd_engine* e, void* stream, const float* xh_lig,
                           const float* xh_pocket, const float* t, int64_t t_count,
                           const int64_t* mask_lig, const int64_t* mask_pocket, int64_t n_lig,
                           int64_t n_pocket, int64_t batch, const int32_t* ext_row,
                           const int32_t* ext_col, int64_t ext_n_edges, float* eps_lig,
                           float* eps_pocket, int32_t* status);

/* Timing of the dominant kernel (the fused GCL edge stage): with enable = k > 0,
 * the largest launches of that kernel (all of them when every stage runs on the whole edge list) inside every
 * k-th dsbdd_dynamics_forward call are bracketed by hipEventRecord on the caller's stream (up to max_launches timed
 * launches between reads); those calls run eagerly, the others may replay their
 * captured graph.  enable = 1 times every call, 0 switches timing off.  dsbdd_engine_profile_read waits for the recorded events, returns
 * the summed kernel time and the number of timed launches, and resets. */
int dsbdd_engine_profile(dsbdd_engine* e, int enable, int max_launches);
int dsbdd_engine_profile_read(dsbdd_engine* e, double* total_ms, int64_t* launches);

/* hipGraph bookkeeping.  dsbdd_dynamics_forward captures its launch sequence into a
 * hipGraph the second time it is called with an identical argument signature
 * (pointers + sizes + stream) and replays it afterwards -- a sampling chain calls it
 * with identical arguments every reverse step.  Eager launches are used while the
 * timing / trace hooks are active, for teacher-forced edge lists, or with
 * DSBDD_GRAPH=0.  Counters: graph replays, captures, eager calls. */
int dsbdd_engine_graph_stats(const dsbdd_engine* e, int64_t* replays, int64_t* captures, int64_t* eager);

/* Which rows the message stages of the last forward evaluated (host-side bookkeeping, no sync): radius[g] = hop
 * level up to which stage g computed its rows (4 = every row), ghost[g] = 1 when the stage also evaluated the
 * canonical pocket (identical pockets, csrc/engine.hip "forward cone"), timed_level = radius of the launches that
 * dsbdd_engine_profile brackets.  All 4 / 0 unless the call was a ligand-output-only call in pocket-conditioning
 * mode (eps_pocket == NULL), see csrc/graph.h "Level-ordered edge list". */
int dsbdd_engine_last_plan(const dsbdd_engine* e, int32_t* radius, int32_t* ghost, int32_t capacity,
                           int32_t* n_stages, int32_t* timed_level);

/* Switches of the dead-row elimination (environment: DSBDD_PRUNE, DSBDD_CONE).  DSBDD_OPT_PRUNE: 0 / 1.
 * DSBDD_OPT_CONE: 0 = never, 1 = when the engine's cost model says the canonical-pocket network pays (default: the
 * frame's representatives hold at most 0.4 of the batch's pocket rows; the DDPM modules decide per chain from the
 * pocket groups with the measured break-even 0.2 and pass 0 / 2), 2 = always.  Cone on / off agree to rounding.
 * DSBDD_OPT_GRANULE16: bit mask of the stages that run on the 16-edge-granule variant of the fused edge kernels
 * (csrc/edge_wave16.h: a wave owns 16 edges on v_mfma_f32_16x16x4_f32 instead of 32 on 32x32x2; half the work unit, for
 * launches too small to fill the SIMDs a whole number of times): bit g (g < 16) = message stage g (block * inv_sublayers +
 * sublayer), bit 16 + b = the coordinate stage of block b.  Default 0 (environment: DSBDD_GRANULE16=<mask>).  The two
 * variants agree to rounding (< 1e-6 relative measured); each is bitwise reproducible; the mask is never changed by the
 * engine itself.
 * DSBDD_OPT_SPLITK (round 6; environment DSBDD_SPLITK=<mask>): bit mask, same layout as DSBDD_OPT_GRANULE16, of the stages
 * that run on the split-K variant of the fused edge kernels (csrc/edge_splitk.h: a WORKGROUP owns 32 edges, wave w a
 * quarter of the reduction dimension of the H x H layer for all H columns, B operand straight from L2, the four partial
 * accumulators reduce-scattered through LDS): a quarter of the default kernel's work unit (7 instead of 27 us of matrix
 * time per wave) with the A operand still evaluated once per element -- for launches of fewer tiles than the chip has
 * resident workgroups (crossdock_ca_cond, the free-running full-atom chain).  hidden_nf = 256 only; elsewhere, and with
 * DSBDD_OPT_EMU != 0, the mask is ignored.  A stage named in both masks runs the split-K kernel.  Same aggregation
 * protocol (head slots per 32-edge tile); results differ from the default kernel in the association of the K sum only
 * (four partial sums); bitwise reproducible; the mask is never changed by the engine itself.
 * DSBDD_OPT_EMU (round 5; environment DSBDD_EMU): arithmetic of the H x H layer of the fused edge kernels.  0 (default) =
 * exact fp32 on v_mfma_f32_32x32x2_f32.  6 / 9 = fp32 EMULATED on the bf16 matrix cores: both operands split exactly into
 * three bf16 terms, the 6 largest (or all 9) partial products accumulated in fp32 by v_mfma_f32_32x32x16_bf16
 * (csrc/edge_wave.h, "emulated path").  Same inputs, same edge order, same aggregation protocol; the results differ from
 * the exact path in rounding only (error vs a float64 evaluation <= 2 x the exact path's own -- the tested bound,
 * tests/test_gpu_emu.py; measured <= 1.1 x: profiles/r5_emu_error.md) and every parity test of the exact path holds at
 * the same tolerance.  Part of a chain's definition like the granule mask: never changed by the engine.  The
 * 16-edge-granule kernels and the training kernels have no emulated form: with DSBDD_OPT_EMU != 0 the DSBDD_OPT_GRANULE16
 * mask is IGNORED (every edge stage of a forward call runs the emulated 32-edge kernel, so a chain never mixes exact and
 * emulated stages); the training step (dsbdd_train_*) always computes in exact fp32.
 * (Round 5 also made the fma of cond_update_kernel / cond_repaint_kernel explicit: the un-fused reverse-step path
 * differs from earlier rounds' results by up to 1 ulp.) */
enum { DSBDD_OPT_PRUNE = 0, DSBDD_OPT_CONE = 1, DSBDD_OPT_GRANULE16 = 2, DSBDD_OPT_EMU = 3, DSBDD_OPT_SPLITK = 4 };
int dsbdd_engine_set_option(dsbdd_engine* e, int which, int value);
/* current value of an option (what the environment / set_option left); DSBDD_ERR_ARG (< 0) for an unknown id */
int dsbdd_engine_get_option(const dsbdd_engine* e, int which);

/* Introspection of the last forward (device pointers into the workspace). */
enum {
  DSBDD_BUF_EDGE_ROW = 0, DSBDD_BUF_EDGE_COL, DSBDD_BUF_EDGE_D0, DSBDD_BUF_ROW_PTR,
  DSBDD_BUF_H, DSBDD_BUF_X, DSBDD_BUF_NODE_BATCH, DSBDD_BUF_DEG,
  /* level-ordered list of the last ligand-output-only call in pocket-conditioning mode (csrc/graph.h):
   * level[N], nodes by (level, id) [N], cumulative node counts [5], edge prefix ends [5], and the list */
  DSBDD_BUF_LEVEL, DSBDD_BUF_LEVEL_LIST, DSBDD_BUF_LEVEL_COUNT, DSBDD_BUF_LEVEL_END,
  DSBDD_BUF_LROW_PTR, DSBDD_BUF_LEDGE_ROW, DSBDD_BUF_LEDGE_COL, DSBDD_BUF_LEDGE_D0,
  /* uint64[16]: sums over the calls since the workspace was bound of the 5 node counts, the 5 list prefix
   * lengths, the 5 edge counts (level <= r), and the number of such calls */
  DSBDD_BUF_LEVEL_STATS
};
int dsbdd_engine_buffer(const dsbdd_engine* e, int which, void** ptr_out);

/* ---- DDPM reverse-step updates -------------------------------------------
 * ConditionalDDPM.sample_p_zs_given_zt after the dynamics call
 * (conditional_model.py:448-464): in place
 *   z_lig <- z_lig/alpha_ts - c_eps*eps_lig + sigma*noise ; then the ligand
 *   centre of mass of each sample is subtracted from its ligand AND pocket x.
 * remove_com = 0 reproduces SimpleConditionalDDPM (conditional_model.py:717-721). */
int dsbdd_cond_reverse_update(void* stream, float* z_lig, float* xh_pocket, const float* eps_lig,
                              const float* noise, const int64_t* mask_lig,
                              const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket,
                              int64_t batch, int32_t atom_nf, int32_t residue_nf, float alpha_ts,
                              float c_eps, float sigma, int32_t remove_com);

/* EnVariationalDiffusion.sample_p_zs_given_zt after the dynamics call
 * (en_diffusion.py:530-556): both node sets are updated and the joint COM is
 * removed.  center_noise != 0: the x part (first 3 columns) of noise_* is made
 * COM-free over the sample's ligand + pocket rows inside the kernel
 * (sample_center_gravity_zero_gaussian_batch, en_diffusion.py:932-942);
 * center_noise = 0: the caller passes noise that is already COM-free. */
int dsbdd_joint_reverse_update(void* stream, float* z_lig, float* z_pocket, const float* eps_lig,
                               const float* eps_pocket, const float* noise_lig,
                               const float* noise_pocket, const int64_t* mask_lig,
                               const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket,
                               int64_t batch, int32_t atom_nf, int32_t residue_nf, float alpha_ts,
                               float c_eps, float sigma, int32_t center_noise);

/* ---- the other per-step pieces of the sampling loops ------------------------
 * One workgroup per sample, fixed reduction order: every result is a pure function
 * of that sample's rows (bitwise reproducible, independent of the batch composition).
 *
 * out[b][0..2] = mean over the rows of sample b of x[:, 0..2] (x [n_rows][ld], mask sorted);
 * torch_scatter.scatter_mean semantics (count clamped to >= 1). */
int dsbdd_segment_mean3(void* stream, const float* x, int32_t ld, const int64_t* mask, int64_t n_rows,
                        int64_t batch, float* out);

/* ConditionalDDPM.sample_normal_zero_com / noised_representation / sample_p_zt_given_zs
 * (conditional_model.py:140-183,420-430), in place:
 *   z_lig <- a * z_lig + sigma * noise ; remove_com: ligand COM subtracted from ligand and pocket x. */
int dsbdd_cond_affine_noise(void* stream, float* z_lig, float* xh_pocket, const float* noise,
                            const int64_t* mask_lig, const int64_t* mask_pocket, int64_t n_lig,
                            int64_t n_pocket, int64_t batch, int32_t atom_nf, int32_t residue_nf, float a,
                            float sigma, int32_t remove_com);

/* EnVariationalDiffusion.sample_combined_position_feature_noise / noised_representation /
 * sample_p_zt_given_zs (en_diffusion.py:302-317,479-501,559-578), in place on both node sets:
 *   z <- a * z + sigma * noise, noise optionally COM-centred (x part), result optionally COM-free.
 *   a = 0, sigma = 1 draws z_T. */
int dsbdd_joint_affine_noise(void* stream, float* z_lig, float* z_pocket, const float* noise_lig,
                             const float* noise_pocket, const int64_t* mask_lig, const int64_t* mask_pocket,
                             int64_t n_lig, int64_t n_pocket, int64_t batch, int32_t atom_nf,
                             int32_t residue_nf, float a, float sigma, int32_t center_noise,
                             int32_t remove_com);

/* One RePaint iteration of ConditionalDDPM.inpaint after the reverse step
 * (conditional_model.py:600-660).  On entry z_lig = denoised state, xh_pocket = pocket moved by
 * that step.  Known part noised to level s around the moved pocket, COM of the fixed atoms of both
 * parts aligned, blend by `fixed`, optional q(z_t | z_s) resampling step; the pocket follows every
 * translation.  scratch_lig [n_lig][3 + atom_nf]; com_pocket0 [batch][3]; fixed [n_lig]. */
int dsbdd_cond_repaint_update(void* stream, float* z_lig, float* xh_pocket, float* scratch_lig,
                              const float* xh0_lig, const float* com_pocket0, const float* fixed,
                              const float* noise_known, const float* noise_resample, const int64_t* mask_lig,
                              const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                              int32_t atom_nf, int32_t residue_nf, float alpha_s, float sigma_s,
                              float alpha_ts, float sigma_ts, int32_t resample, int32_t remove_com);

/* One launch per reverse step of a pocket-conditioned chain drawn from the keyed generator (round 5; replaces
 * dsbdd_randn_keyed x 1-3 + dsbdd_cond_reverse_update [+ dsbdd_cond_repaint_update] + the fill of the denoiser's time
 * word): the posterior update of conditional_model.py:448-464 with the noise of draw `draw_index` evaluated in place;
 * repaint = 1: followed by the RePaint iteration of :600-660 (known part noised with draw + 1), 2: and the q(z_t | z_s)
 * jump (draw + 2).  alpha_ts / sigma_ts are shared by the update and the jump (the same step); alpha_s / sigma_s noise
 * the known part.  t_word (optional): device float that receives t_next at the end -- the `t` of the NEXT
 * dsbdd_dynamics_forward call.  Results are bitwise those of the separate entry points on dsbdd_randn_keyed's output. */
int dsbdd_cond_step_keyed(void* stream, float* z_lig, float* xh_pocket, const float* eps_lig, float* scratch_lig,
                          const float* xh0_lig, const float* com_pocket0, const float* fixed, const int64_t* mask_lig,
                          const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch, int32_t atom_nf,
                          int32_t residue_nf, float alpha_ts, float c_eps, float sigma, int32_t repaint, float alpha_s,
                          float sigma_s, float sigma_ts, int32_t remove_com, uint64_t seed, uint64_t draw_index,
                          int64_t sample_offset, const int64_t* sample_ids, float* t_word, float t_next);

/* One RePaint iteration of EnVariationalDiffusion.inpaint after the reverse step
 * (en_diffusion.py:742-809): known part q(z_s | x) with COM-centred noise, COM alignment over the
 * fixed ligand + pocket nodes, blend, optional jump back q(z_t | z_s) + joint COM removal. */
int dsbdd_joint_repaint_update(void* stream, float* z_lig, float* z_pocket, float* scratch_lig,
                               float* scratch_pocket, const float* xh0_lig, const float* xh0_pocket,
                               const float* fixed_lig, const float* fixed_pocket, const float* noise_known_lig,
                               const float* noise_known_pocket, const float* noise_jump_lig,
                               const float* noise_jump_pocket, const int64_t* mask_lig,
                               const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                               int32_t atom_nf, int32_t residue_nf, float alpha_s, float sigma_s,
                               float alpha_ts, float sigma_ts, int32_t jump);

/* Counter-based Gaussian noise (Philox4x32-10 + Box-Muller), a pure function of
 * (seed, global sample id, row within the sample, column, draw index, stream id), so
 * that a chain's noise does not depend on how samples are sharded over GPUs or packed
 * into batches.  out [n_rows][n_cols]; global sample id of row i =
 * sample_ids[mask[i]] when sample_ids != NULL (int64 [batch], device), else
 * mask[i] + sample_offset. */
int dsbdd_randn_keyed(void* stream, float* out, const int64_t* mask, int64_t n_rows,
                      int32_t n_cols, int64_t batch, int64_t sample_offset,
                      const int64_t* sample_ids, uint64_t seed, uint64_t draw_index,
                      uint32_t stream_id);

/* ---- individual kernels (unit tests, building blocks) --------------------*/
/* C[M][N] = act([A1 | A2] @ WT + bias) (+ R);  WT is [K1+K2][ldw]. act: 0 none, 1 SiLU */
int dsbdd_node_linear(void* stream, const float* A1, int32_t lda1, int32_t K1, const float* A2,
                      int32_t lda2, int32_t K2, const float* WT, int32_t ldw, const float* bias,
                      const float* R, int32_t ldr, float* C, int32_t ldc, int64_t M, int32_t N,
                      int32_t act);

/* Radius graph (dynamics.py:169-187) -> edge list sorted by (row, col).
 * x [n_lig+n_pocket][3]; scratch: node_batch [N], lig_off/poc_off [batch+1],
 * deg [N], row_ptr [N+1] (row_ptr[N] = number of edges). */
int dsbdd_build_edges(void* stream, const float* x, const int64_t* mask_lig,
                      const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                      const dsbdd_config* cfg, int32_t* node_batch, int32_t* lig_off,
                      int32_t* poc_off, int32_t* deg, int32_t* row_ptr, int32_t* edge_row,
                      int32_t* edge_col, float* edge_d0, int64_t edge_capacity, int32_t* status);

/* ---- training step: backward of the fused edge stages -------------------------------------------
 * The reference trains through `loss.backward()` over its eager op stream (lightning_modules.py:337-363 ->
 * conditional_model.py:202-330 / en_diffusion.py:336-469 -> dynamics.py:87-167 -> egnn_new.py:31-58,96-122).  These entry
 * points are the hand-written forward / backward pairs of the two fused edge stages and the weight-gradient GEMM; the
 * Python side (diffsbdd_amd/train_hip.py) wraps them as torch.autograd.Function objects behind EGNNDynamics.forward.
 * Nothing of size [n_edges][H] is kept from the forward pass: the backward recomputes the edge activations from the
 * per-node projections.  All sums over edges are taken in a fixed order (bitwise reproducible gradients, no atomics).
 *
 * graph: the (row, col)-sorted list of dsbdd_build_edges plus rev (dsbdd_train_edge_rev: index of the edge (col, row)). */
typedef struct dsbdd_train_graph {
  const int32_t *erow, *ecol; const float* ed0;     /* [n_edges]                                    */
  const int32_t *row_ptr, *deg;                     /* [n_nodes + 1], [n_nodes]                     */
  const int32_t* rev;                               /* [n_edges] (backward only)                    */
  const int32_t *node_batch, *lig_off, *poc_off;    /* [n_nodes], [batch + 1], [batch + 1]          */
  int64_t n_lig, n_nodes, n_edges, batch;
} dsbdd_train_graph;

/* one edge MLP in its factorised form (csrc/edge_mlp.h): z1 = P[row] + Q[col] + d wd + d0 wd0 + tab[type] */
typedef struct dsbdd_train_mlp {
  const float *P, *Q; int32_t ldpq;   /* [n_nodes][ldpq] first-layer projections of the row / column node             */
  const float *wd, *wd0, *tab;        /* [H], [H], [3][H]                                                             */
  const float *W2, *W2T, *b2;         /* second layer: nn.Linear layout [out][in], its transpose [in][out], bias [H]  */
  const float *head, *head_b;         /* GCL: att_mlp.0.weight [H] and bias [1] (head = NULL: no attention);
                                         coordinate MLPs: the bias-free output layer [H] (egnn_new.py:78)             */
} dsbdd_train_mlp;

/* gradient destinations of one edge MLP */
typedef struct dsbdd_train_mlp_grad {
  float *dP, *dQ; int32_t ldo;        /* [n_nodes][ldo]: every row is written                                         */
  float* d_vec;                       /* [8][H]: d_wd, d_wd0, d_tab[0..2], d_b2, d_head, d_head_b (element 0)         */
  float* d_W2;                        /* [H][H] nn.Linear layout                                                      */
  float* gd0;                         /* [n_edges] gradient w.r.t. ed0                                                */
} dsbdd_train_mlp_grad;

size_t dsbdd_train_scratch_bytes(int32_t H, int64_t n_nodes, int64_t n_edges);
/* scratch of dsbdd_train_wgrad: enough for every K' <= K (the split-K plan is not monotonic in K, so a scratch sized
   for an edge count E also serves the calls on an edge prefix); dsbdd_train_wgrad_plan_bytes = what a call with exactly
   this K uses (never more than the former; a call whose plan does not fit returns DSBDD_ERR_CAPACITY) */
size_t dsbdd_train_wgrad_scratch_bytes(int64_t K, int64_t M, int64_t N);
size_t dsbdd_train_wgrad_plan_bytes(int64_t K, int64_t M, int64_t N);
int dsbdd_train_edge_rev(void* stream, const dsbdd_train_graph* g, int32_t* rev);
/* mean[b] = mean position of ALL nodes of sample b (coord2cross, egnn_new.py:307-310) */
int dsbdd_train_sample_mean(void* stream, const float* x, const dsbdd_train_graph* g, float* mean);

/* GCL.edge_model + aggregation (egnn_new.py:31-52): agg [n_nodes][H] = sum over the row's edges of m * att / nf */
int dsbdd_train_gcl_forward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m,
                            const float* x, float norm_factor, float* agg, void* scratch, size_t scratch_bytes);
/* d_agg [n_nodes][H] -> gradients of the MLP's inputs and parameters; d_x [n_nodes][3] = gradient through |x_i - x_j|^2 */
int dsbdd_train_gcl_backward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m,
                             const float* x, float norm_factor, const float* d_agg,
                             const dsbdd_train_mlp_grad* out, float* d_x, void* scratch, size_t scratch_bytes);

/* EquivariantUpdate.coord_model (egnn_new.py:96-122): x_out = x + sum over the edges of rows < n_upd of
 * (u T(phi) + cross T(phi_x)) / nf; m[0] = coord_mlp, m[1] = cross_product_mlp (n_mlp = 2), m[0].head = the shared
 * output layer.  mean [batch][3] from dsbdd_train_sample_mean (n_mlp = 2 only). */
int dsbdd_train_coord_forward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m,
                              int32_t n_mlp, const float* x, const float* mean, int64_t n_upd, float norm_constant,
                              float coords_range, int32_t use_tanh, float norm_factor, float* x_out, void* scratch,
                              size_t scratch_bytes);
/* d_xout [n_nodes][3] -> out[0 .. n_mlp) and d_x [n_nodes][3] = the gradient through the geometry (u, cross,
 * |x_i - x_j|^2; the identity path x -> x_out is NOT included), d_mean [batch][3].  e_upd = row_ptr[n_upd] (host value).
 * out[1].d_vec row 6 (d_head) belongs to the shared output layer as well: the caller adds it to out[0]'s. */
int dsbdd_train_coord_backward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m,
                               int32_t n_mlp, const float* x, const float* mean, int64_t n_upd, int64_t e_upd,
                               float norm_constant, float coords_range, int32_t use_tanh, float norm_factor,
                               const float* d_xout, const dsbdd_train_mlp_grad* out, float* d_x, float* d_mean,
                               void* scratch, size_t scratch_bytes);

/* gradient of the squared edge lengths d_e = |x_row - x_col|^2 (the edge list's ed0 when x is the call's input):
 * d_x[i] = sum over row i's edges of 2 (gd[e] + gd[rev e]) (x_i - x_col) */
int dsbdd_train_radial_backward(void* stream, const dsbdd_train_graph* g, const float* x, const float* gd, float* d_x);

/* C[M][N] = sum_k A[k][m] B[k][n]  (A [K][lda], B [K][ldb]; the weight gradient dY^T X of a Linear layer), split-K with
 * an ordered reduction.  out[n] = sum_m A[m][n] (bias gradient); scratch >= 4 * ceil(M / 32) * N bytes. */
int dsbdd_train_wgrad(void* stream, const float* A, int32_t lda, const float* B, int32_t ldb, int64_t K, int32_t M,
                      int32_t N, float* C, void* scratch, size_t scratch_bytes);
int dsbdd_train_colsum(void* stream, const float* A, int32_t lda, int64_t M, int32_t N, float* out, void* scratch,
                       size_t scratch_bytes);

/* ---- the training step as ONE launch sequence per direction (round 6; csrc/train_net.h) ---------------------------
 * Replaces, for the training step, the composition of the building blocks above by PyTorch autograd nodes:
 * lightning_modules.py:337-363 -> conditional_model.py:202-330 / en_diffusion.py:336-469 -> EGNNDynamics.forward
 * (dynamics.py:87-167) under autograd.  The reference-side binding is ONE torch.autograd.Function.
 *
 * params / grads: the module's parameter tensors in nn.Linear layout [out][in], in the order
 *   {atom_encoder.0, atom_encoder.2, atom_decoder.0, atom_decoder.2, residue_encoder.0, residue_encoder.2,
 *    residue_decoder.0, residue_decoder.2}.{weight, bias}, [edge_embedding.weight], egnn.embedding.{weight, bias},
 *   egnn.embedding_out.{weight, bias}, then per block i: per sublayer s gcl_s.{edge_mlp.0, edge_mlp.2, node_mlp.0,
 *   node_mlp.2, [att_mlp.0]}.{weight, bias}, gcl_equiv.{coord_mlp.0, coord_mlp.2}.{weight, bias}, coord_mlp.4.weight,
 *   [cross_product_mlp.{0, 2}.{weight, bias}]
 * (dsbdd_train_net_param_count entries; the aliased cross_product_mlp.4.weight IS coord_mlp.4.weight and is not listed:
 * its gradient is the sum over both MLPs).  Every gradient tensor is OVERWRITTEN.
 * graph: dsbdd_build_edges + dsbdd_train_edge_rev on the call's input coordinates (n_edges is a host value).
 * pack: persistent caller-owned buffer (dsbdd_train_net_pack_bytes) for the re-laid-out weights, rewritten by every
 * forward; ws: per-call workspace (dsbdd_train_net_workspace_bytes) that carries the activations from forward to backward
 * -- including the second-layer pre-activation z2 [E][H] of every edge MLP (4 E H bytes each; DSBDD_TRAIN_STORE_Z2=0:
 * recomputed in the backward pass instead) -- and two scratch sets for the coordinate stage's two MLPs.
 * Nothing is allocated by the library; no host synchronisation inside (beyond the first call's descriptor upload).
 * Streams: everything is ordered on `stream` as seen by the caller.  Inside, the backward runs the second coordinate
 * MLP's chain and the node-level / coordinate weight gradients on two internal non-blocking streams of the handle
 * (created by the first backward call; fork / join by events, all joined before the call's last launches on `stream`;
 * same kernels and reduction orders, so the gradients are bitwise those of DSBDD_TRAIN_STREAMS=0, one stream).
 * zero_nan: training mode replaces NaN velocities by 0 (dynamics.py:155-159); otherwise bit 1 of *status is raised.
 * e_upd (backward): row_ptr[n_lig] as a host value (pocket-conditioned models; ignored when update_pocket_coords).
 * d_xh_lig / d_xh_pocket: gradients w.r.t. the inputs, or NULL. */
typedef struct dsbdd_train_net dsbdd_train_net;
int dsbdd_train_net_create(const dsbdd_config* cfg, dsbdd_train_net** out);
void dsbdd_train_net_destroy(dsbdd_train_net* net);
int dsbdd_train_net_param_count(const dsbdd_train_net* net);
size_t dsbdd_train_net_pack_bytes(const dsbdd_train_net* net);
size_t dsbdd_train_net_workspace_bytes(const dsbdd_train_net* net, const dsbdd_train_graph* g);
int dsbdd_train_net_forward(dsbdd_train_net* net, void* stream, const dsbdd_train_graph* g, const float* const* params,
                            void* pack, size_t pack_bytes, void* ws, size_t ws_bytes, const float* xh_lig,
                            const float* xh_pocket, const float* t, int64_t t_count, int32_t zero_nan, float* eps_lig,
                            float* eps_pocket, int32_t* status);
int dsbdd_train_net_backward(dsbdd_train_net* net, void* stream, const dsbdd_train_graph* g, const float* const* params,
                             float* const* grads, void* pack, size_t pack_bytes, void* ws, size_t ws_bytes,
                             int64_t e_upd, const float* d_eps_lig, const float* d_eps_pocket, float* d_xh_lig,
                             float* d_xh_pocket);

/* ---- the loss terms of the pocket-conditioned training step --------------------------------------------
 * ConditionalDDPM.forward around the network call (conditional_model.py:202-330 -> en_diffusion.py:109-262 for the terms):
 * normalisation, ligand-COM removal, z_t = alpha_t xh0 + sigma_t eps (centred again), and the twelve per-sample terms.
 * The reference-side binding calls _pre, then EGNNDynamics.forward (dsbdd_train_net_forward), then _post inside ONE
 * torch.autograd.Function whose backward is _post_backward (diffsbdd_amd/loss_head.py).
 * Predefined noise schedules only (gamma_table [timesteps + 1], en_diffusion.py:1158-1190); training mode (t may be 0: the
 * L0 terms are evaluated on z_t and masked by [t = 0], conditional_model.py:285-302).
 *   lig_x [n_lig][3], lig_h [n_lig][atom_nf] (raw one-hot), lig_mask [n_lig] int64 sorted ascending; pocket likewise;
 *   eps [n_lig][3 + atom_nf] standard normal draws; t_int [batch] integer-valued floats in [0, timesteps];
 *   logpn_table [n1_tab][n2_tab] = log p(n_lig | n_pocket) or NULL.
 * _pre writes z_t [n_lig][3 + atom_nf], xh_pocket [n_pocket][3 + residue_nf] and per_sample [dsbdd_loss_rows()][batch]:
 *   rows t, gamma_t, gamma_s, alpha_t, sigma_t, SNR_weight, neg_log_constants, kl_prior, loss_0_h (x [t = 0]), log_pN,
 *   delta_log_px, [t = 0]; optionally (non-NULL) the normalised batch x / norm_value_x, (h - norm_bias_h) / norm_value_h that
 *   normalize() (en_diffusion.py:880-895) leaves in the caller's dictionaries.
 * _post writes xh_hat [n_lig][3 + atom_nf] and out [dsbdd_loss_out_rows()][batch]: error_t (x [t > 0]), loss_0_x (x [t = 0]),
 *   mean |net_x|, mean |net_h| per sample.  _post_backward: d_net from the gradients of error_t / loss_0_x (per sample) and of
 *   xh_hat (or NULL).  Every per-sample sum is taken in a fixed order (the reference's scatter_add uses atomics). */
typedef struct {
  int32_t batch, n_lig, n_pocket, atom_nf, residue_nf, timesteps;
  int32_t remove_com;     /* 1: ConditionalDDPM (ligand COM removed from ligand and pocket, dof = 3 (n - 1)); 0: SimpleConditionalDDPM */
  int32_t vnode_idx;      /* class index of the virtual atom (its coordinates do not enter the error terms) or -1 */
  float norm_value_x, norm_value_h, norm_bias_h;
  int32_t n1_tab, n2_tab;
} dsbdd_loss_cfg;
/* Upper bound on the edge list dsbdd_build_edges writes for these (sorted) batch masks, and their validity, in one launch:
 * out[0] = 1 when both masks are sorted ascending with ids in [0, batch), out[1] = sum over the samples of the complete
 * graph's edges (dynamics.py:170-172) with every (sample, node set) segment rounded up to 32.  The host reads both words
 * (diffsbdd_amd/engine.py edge_capacity). */
int dsbdd_edge_capacity(void* stream, const int64_t* lig_mask, int64_t n_lig, const int64_t* pocket_mask, int64_t n_pocket,
                        int64_t batch, int64_t* out);
int dsbdd_loss_rows(void);
int dsbdd_loss_out_rows(void);
int dsbdd_loss_cond_pre(void* stream, const dsbdd_loss_cfg* cfg, const float* lig_x, const float* lig_h, const int64_t* lig_mask,
                        const float* pocket_x, const float* pocket_h, const int64_t* pocket_mask, const float* eps,
                        const float* t_int, const float* gamma_table, const float* logpn_table, float* z_t, float* xh_pocket,
                        float* per_sample, float* lig_x_norm, float* lig_h_norm, float* pocket_x_norm, float* pocket_h_norm);
int dsbdd_loss_cond_post(void* stream, const dsbdd_loss_cfg* cfg, const float* net, const float* eps, const float* z_t,
                         const float* lig_h, const int64_t* lig_mask, const float* per_sample, float* xh_hat, float* out);
int dsbdd_loss_cond_post_backward(void* stream, const dsbdd_loss_cfg* cfg, const float* net, const float* eps, const float* lig_h,
                                  const int64_t* lig_mask, const float* per_sample, const float* g_err, const float* g_l0x,
                                  const float* g_hat, float* d_net);

/* ---- post-processing of a finished batch ------------------*/
/* Distance-based bond orders of a batch of molecules: replaces
 * get_bond_order_batch + the (X, A, E) step of make_mol_edm
 * (analysis/molecule_builder.py:30-55, :101-118).  x [N][3] in Angstrom,
 * atom_type [N] indices into the [n_types][n_types] single/double/triple
 * length tables (pm), mol_off [batch+1] first atom of each molecule.
 * order [batch][n_max][n_max] int8 receives the strictly lower triangle
 * (order[b][i][j], i > j: 0 none, 1 single, 2 double, 3 triple); everything
 * else is set to 0.  Molecules longer than n_max are truncated to n_max atoms. */
int dsbdd_bond_orders(void* stream, const float* x, const int32_t* atom_type,
                      const int32_t* mol_off, int64_t batch, int32_t n_types,
                      const float* bonds1, const float* bonds2, const float* bonds3,
                      float margin1, float margin2, float margin3, int32_t n_max,
                      int8_t* order);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSBDD_HIP_H */
