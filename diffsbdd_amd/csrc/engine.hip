// libdiffsbdd_hip.so -- C-ABI implementation (see include/diffsbdd_hip.h).
// Host-side orchestration of one EGNNDynamics.forward call
// (/root/reference/equivariant_diffusion/dynamics.py:87-167) as a fixed sequence
// of asynchronous launches on the caller's stream: no allocation, no host sync.
#include "../../include/diffsbdd_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "ddpm.h"
#include "edge_mlp.h"
#include "edge_wave.h"
#include "edge_wave16.h"
#include "edge_splitk.h"
#include "graph.h"
#include "lig_head.h"
#include "molecule.h"
#include "node_chain.h"
#include "node_linear.h"
#include "train.h"

using namespace dsbdd;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return fail(DSBDD_ERR_LAUNCH, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// Every entry point launches on the caller's stream: make that stream's device the current one for the duration of the
// call (a module living on a GPU that is not torch's current device must not launch through another device's
// context), and restore it afterwards.  The NULL stream belongs to the current device by definition.
struct StreamDevice {
  int prev = -1;
  bool switched = false;
  explicit StreamDevice(void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!s) return;
    hipDevice_t d = 0;
    int cur = 0;
    if (hipStreamGetDevice(s, &d) != hipSuccess || hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    if ((int)d != cur) { prev = cur; switched = hipSetDevice((int)d) == hipSuccess; }
  }
  ~StreamDevice() { if (switched) (void)hipSetDevice(prev); }
  StreamDevice(const StreamDevice&) = delete;
  StreamDevice& operator=(const StreamDevice&) = delete;
};

static inline int pad4(int v) { return (v + 3) & ~3; }
static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct dsbdd_engine {
  dsbdd_config cfg;
  std::vector<const float*> slots;
  bool has_weights = false;
  // workspace
  char* ws = nullptr;
  size_t ws_bytes = 0;
  int64_t cap_lig = 0, cap_poc = 0, cap_batch = 0, cap_edges = 0;
  int *node_batch, *lig_off, *poc_off, *deg, *row_ptr, *erow, *ecol;
  int *act_flag, *act_ptr, *act_list;
  float *ed0, *x, *x_in, *xagg, *mean, *h0, *enc_tmp, *h, *t1, *agg, *pq, *pqg, *hout, *w2tp;
  float *agg_head, *xagg_head;          // partial sums of rows continuing from the previous wave tile
  int *scan_tmp, *seg_base, *tile_ctr;
  // second per-call list (edges with a ligand endpoint) and the static pocket-pocket list of the pocket frame
  int *erow2, *ecol2, *row_ptr2, *deg2, *scan_tmp2, *seg_base2;
  int *erow3, *ecol3, *row_ptr3, *deg3, *scan_tmp3, *seg_base3, *node_batch3, *lig_off3, *poc_off3, *twin;
  float *ed02, *ed03, *xframe, *aggB, *agg_headB;
  int* frame_rows;                      // pocket row (0-based in the pocket array) of every frame row
  // level-ordered list (graph.h, "Level-ordered edge list"): pocket-conditioned calls that return the ligand part only
  int *lvl, *seg_rows, *seg_edges, *node_base, *edge_base, *lvl_cnt, *lvl_end, *lvl_list, *row_ptrL, *erowL, *ecolL;
  float* ed0L;
  unsigned long long* lvl_stats = nullptr;
  bool lvl_stats_zeroed = false;
  int64_t cap_edgesL = 0;
  int prune = 1;                        // DSBDD_PRUNE=0: evaluate every row in every stage
  // forward cone (identical pockets): the first message stages evaluate only the rows the ligand can have influenced;
  // the rest take the values of the canonical pocket, computed once on ghost rows N .. N + n_ghost
  int cone = 1;                         // DSBDD_CONE=0: off, 1: when the cost model says it pays (default), 2: always
  int64_t ghost_slots = 0;              // slots of the ghost segment at the front of the level-ordered list
  // plan of the last forward (host side): radius and ghost use of every message stage, level of the timed launches
  std::vector<int> plan_radius, plan_ghost;
  int plan_timed_level = kLevels - 1;
  // pocket frame of the running chain (dsbdd_engine_set_pocket_frame): raw pocket coordinates are rigid in
  // pocket-conditioning mode, so block 0's pocket-pocket messages are evaluated on them, separately
  bool frame = false;
  int64_t frame_nlig = 0, frame_npoc = 0, frame_batch = 0, frame_n3 = 0, frame_cap3 = 0;
  bool ghost_dirty = true;              // the ghost rows / ghost list segment must be (re)written before the next framed call
  bool h0_pocket_valid = false;         // the pocket rows of h0 hold this chain's encoded pocket features (a frame fixes the
                                        // pocket of a chain: coordinates up to translation AND features, which never change in
                                        // pocket-conditioning mode) -> framed calls after the first run the ligand encoder only
  int64_t cap_tiles = 0;                // wave tiles (32 edges) of the edge capacity
  int64_t cap_tiles16 = 0;              // the same in 16-edge tiles (edge_wave16.h)
  unsigned granule16 = 0;               // DSBDD_OPT_GRANULE16: bit g = message stage g, bit 16 + b = coordinate stage of block b
                                        // run on the 16-edge-granule kernels (default: none; DSBDD_GRANULE16=<mask> in the environment)
  bool w2tp16_ready = false;            // their lane-grouped W2^T copies are current
  unsigned splitk = 0;                  // DSBDD_OPT_SPLITK: the same bit layout -- stages that run on the split-K kernels (edge_splitk.h:
                                        // a workgroup owns 32 edges, wave w a quarter of the reduction dimension; hidden_nf 256 only)
  bool w2sk_ready = false;              // their per-wave rotated W2^T copies are current
  int emu = 0;                          // DSBDD_OPT_EMU: 0 = exact fp32 edge kernels (default), 6 / 9 = fp32 emulated on the bf16 matrix
                                        // cores with 6 / 9 partial products (edge_wave.h, "emulated path"); DSBDD_EMU=<k> in the environment
  bool w2e_ready = false;               // the bf16 planes of every W2^T are current
  float *trace_h = nullptr, *trace_x = nullptr;
  int n_cu = 256;
  bool w2tp_ready = false;   // lane-grouped W2^T copies in the workspace are current
  // row-owning node-phase kernel (node_chain.h): lane-major packed copies of the node-level weights in the workspace
  float* wchain = nullptr;
  bool wchain_ready = false;
  int chain = 1;             // DSBDD_NODE_CHAIN=0: the three-launch node phase (node_linear.h) everywhere
  int64_t chain_min_rows = 0;      // (test hook; the choice of kernel must not depend on the batch size: bitwise batch invariance)
  int fork_front = 0;  // DSBDD_FORK=1: encoders / embedding on a side stream at the head of a call (measured slower, see forward_impl)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int level_rows = 0;  // DSBDD_LEVEL_ROWS=1: all-row stages of a pruned call walk the level list (measured slower, see rows_of)
  int fold_scan = 1;   // DSBDD_FOLD_SCAN: 1 (default) = the level ordering's exclusive scans are computed by level_place_kernel
                       // itself (no single-workgroup level_scan_kernel launch) and the block-0 sample mean rides in levels_kernel;
                       // 0 = the separate launches.  (The same fold for the radius graph's own scan -- segment totals by
                       // integer atomics in the count pass -- was measured SLOWER in round 5, 19.8 k atomics on 128 counters:
                       // edges_kernel<false> 12 -> 129 us, profiles/r5k_*; its code was removed in round 6.)
  int lig_head = 1;    // DSBDD_LIG_HEAD=0: embedding_out / decoder / finalize as three launches also for ligand-only calls
  int edge_bperm = 1;  // edge_wave.h reads the B operand with 16-byte LDS loads from those copies (DSBDD_EDGE_BPERM=0: off)
  // optional timing of the dominant kernel (GCL edge stage) with HIP events
  int profile = 0;        // 0 off, k: the GCL launches of every k-th forward call are timed with HIP events
  int64_t prof_call = 0;
  bool time_now = false;
  std::vector<hipEvent_t> ev;   // pairs: start, stop
  size_t ev_used = 0;
  // hipGraph cache: the launch sequence of one dynamics call is captured once per argument
  // signature (pointers + sizes) and replayed; a sampling chain calls with identical
  // arguments every reverse step.  DSBDD_GRAPH=0 disables.
  struct GraphEntry {
    std::vector<uint64_t> key;
    int seen = 0;                 // 1st call runs eagerly (warm-up), 2nd captures, then replay
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<int> plan_radius, plan_ghost;   // what dsbdd_engine_last_plan reports after a replay of this graph
    int plan_timed_level = 0;
  };
  std::vector<GraphEntry> graphs;
  int use_graph = 1;
  int coord_split = 1; // edge_wave MODE_COORD: one workgroup per (tile, MLP) (DSBDD_COORD_SPLIT=0: per tile)
  int node_group = 1;  // coordinate projections + next block's P|Q in one launch (DSBDD_NODE_GROUP=0: separate)
  int edge_max_wg = 0;  // test hook (DSBDD_EDGE_MAX_WG): cap on the persistent edge grid, so that small problems
                        // run several tiles per workgroup (the path large batches take)
  int64_t n_replay = 0, n_capture = 0, n_eager = 0;
  unsigned long long* ts_buf = nullptr;   // -DDSBDD_TIMESTAMPS builds: [launches][64][16] marks of the edge kernels
  int ts_cap = 0, ts_next = 0;
  hipStream_t cap_stream = nullptr;   // capture happens here (the caller's stream may be the
                                      // legacy default stream, which cannot be captured)
  // captured graphs hold the raw weight / workspace pointers of the moment they were captured
  void drop_graphs() {
    for (GraphEntry& g : graphs) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    graphs.clear();
  }
  ~dsbdd_engine() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    drop_graphs();
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }
};

static int n_slots(const dsbdd_config& c) {
  return DSBDD_G_COUNT + c.n_layers * (c.inv_sublayers * DSBDD_GCL_COUNT + DSBDD_EQ_COUNT);
}
static int gcl_slot(const dsbdd_config& c, int block, int sub, int which) {
  return DSBDD_G_COUNT + block * (c.inv_sublayers * DSBDD_GCL_COUNT + DSBDD_EQ_COUNT) +
         sub * DSBDD_GCL_COUNT + which;
}
static int eq_slot(const dsbdd_config& c, int block, int which) {
  return DSBDD_G_COUNT + block * (c.inv_sublayers * DSBDD_GCL_COUNT + DSBDD_EQ_COUNT) +
         c.inv_sublayers * DSBDD_GCL_COUNT + which;
}

struct WsLayout {
  size_t off[72];
  size_t total;
};

static WsLayout carve(const dsbdd_config& c, int64_t nl, int64_t np, int64_t B, int64_t E) {
  const int64_t N = nl + np;
  const int H = c.hidden_nf, JP = pad4(c.joint_nf + 1);
  const int LE = pad4(2 * (c.atom_nf > c.residue_nf ? c.atom_nf : c.residue_nf));
  const int PQ = (c.reflection_equivariant ? 2 : 4) * H;
  const int64_t EL = 2 * E + 32 * kLevels * B;   // level-ordered list: ghost segment + one padded segment per (level, sample)
  const int64_t NG = N + np;                 // + the ghost rows of a canonical pocket (forward cone)
  const int64_t T = EL / 32 + 2;             // wave tiles
  size_t sizes[] = {
      (size_t)N * 4, (size_t)(B + 1) * 4, (size_t)(B + 1) * 4, (size_t)NG * 4, (size_t)(N + 1) * 4, // 0-4 (3 deg: + ghosts)
      (size_t)E * 4, (size_t)E * 4, (size_t)E * 4,                                                  // 5-7 erow ecol ed0
      (size_t)NG * 12, (size_t)N * 12, (size_t)N * 24, (size_t)B * 12,                              // 8-11 x x_in xagg[2] mean
      (size_t)N * JP * 4, (size_t)N * LE * 4,                                                       // 12 h0, 13 enc_tmp
      (size_t)NG * H * 4, (size_t)NG * H * 4, (size_t)NG * H * 4, (size_t)N * PQ * 4,               // 14 h 15 t1 16 agg 17 pq
      (size_t)N * JP * 4,                                                                           // 18 hout
      (size_t)N * 4, (size_t)(N + 1) * 4, (size_t)N * 4,                                            // 19-21 act flag/ptr/list
      (size_t)NG * 2 * H * 4,                                                                       // 22 pqg (GCL P|Q)
      (size_t)c.n_layers * (c.inv_sublayers + 2) * H * H * 18 + 4096,                              // 23 lane-grouped W2^T copies (32- and 16-edge kernels: 2 x 4 B) + the bf16 planes of the emulated path (6 B) + the split-K copies (4 B)
      (size_t)2 * T * H * 4, (size_t)2 * T * 2 * 16,                                                // 24 agg_head, 25 xagg_head[2][T][4] (16-edge tiles: 2 T slots)
      (size_t)(N + 1) * 4, (size_t)(2 * B + 1) * 4, (size_t)kTileCtrInts * 4,                       // 26 scan_tmp 27 seg_base 28 tile_ctr
      (size_t)E * 4, (size_t)E * 4, (size_t)E * 4, (size_t)(N + 1) * 4, (size_t)N * 4,              // 29-33 list 2: erow ecol ed0 row_ptr deg
      (size_t)(N + 1) * 4, (size_t)(2 * B + 1) * 4,                                                 // 34 scan_tmp2 35 seg_base2
      (size_t)E * 4, (size_t)E * 4, (size_t)E * 4, (size_t)(N + 1) * 4, (size_t)N * 4,              // 36-40 list 3
      (size_t)(N + 1) * 4, (size_t)(2 * B + 1) * 4,                                                 // 41 scan_tmp3 42 seg_base3
      (size_t)N * 4, (size_t)(B + 1) * 4, (size_t)(B + 1) * 4, (size_t)N * 4,                       // 43 node_batch3 44 lig_off3 45 poc_off3 46 twin
      (size_t)N * 12, (size_t)NG * H * 4, (size_t)T * H * 4,                                        // 47 xframe 48 aggB (ghost ids) 49 agg_headB
      (size_t)N * 4, (size_t)kLevels * B * 4, (size_t)kLevels * B * 4,                              // 50 lvl 51 seg_rows 52 seg_edges
      (size_t)(kLevels * B + 1) * 4, (size_t)(kLevels * B + 1) * 4, 64, 64,                         // 53 node_base 54 edge_base 55 lvl_cnt 56 lvl_end
      (size_t)NG * 4, (size_t)(NG + 1) * 4, (size_t)EL * 4, (size_t)EL * 4, (size_t)EL * 4,         // 57 lvl_list 58 row_ptrL 59-61 erowL ecolL ed0L
      128, (size_t)N * 4,                                                                           // 62 lvl_stats 63 frame_rows
      (size_t)c.n_layers * ((size_t)c.inv_sublayers * 5 * H * H + (size_t)H * PQ) * 4 + 4096};      // 64 packed node-phase weights
  WsLayout L;
  size_t o = 0;
  const int n = sizeof(sizes) / sizeof(sizes[0]);
  for (int i = 0; i < n; ++i) { L.off[i] = o; o += al256(sizes[i] + 16); }
  L.total = o;
  return L;
}

static int build_edges_impl(hipStream_t s, const float* x, int n_lig, int N, int B, const dsbdd_config& c,
                            const int* node_batch, const int* lig_off, const int* poc_off, int* deg,
                            int* row_ptr, int* erow, int* ecol, float* ed0, int64_t cap, int* status,
                            int* act_flag = nullptr, int* scan_tmp = nullptr, int* seg_base = nullptr,
                            const EdgeList2* list2 = nullptr, int id_offset = 0, int* lvl = nullptr);

extern "C" {

int dsbdd_abi_version(void) { return DSBDD_ABI_VERSION; }
const char* dsbdd_last_error(void) { return g_err.c_str(); }

int dsbdd_engine_create(const dsbdd_config* cfg, dsbdd_engine** out) {
  if (!cfg || !out) return fail(DSBDD_ERR_ARG, "null argument");
  const int H = cfg->hidden_nf;
  if (!(H == 64 || H == 128 || H == 192 || H == 256))
    return fail(DSBDD_ERR_ARG, "hidden_nf must be one of 64, 128, 192, 256");
  if (cfg->n_layers < 1 || cfg->inv_sublayers < 1 || cfg->atom_nf < 1 || cfg->residue_nf < 1 ||
      cfg->joint_nf < 1 || cfg->edge_embedding_dim < 0)
    return fail(DSBDD_ERR_ARG, "bad layer/feature counts");
  if (!(cfg->normalization_factor > 0.f)) return fail(DSBDD_ERR_ARG, "normalization_factor must be > 0");
  dsbdd_engine* e = new dsbdd_engine();
  e->cfg = *cfg;
  e->slots.assign(n_slots(*cfg), nullptr);
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      prop.multiProcessorCount > 0)
    e->n_cu = prop.multiProcessorCount;
  const char* ug = getenv("DSBDD_GRAPH");
  if (ug && atoi(ug) == 0) e->use_graph = 0;
  const char* bpe = getenv("DSBDD_EDGE_BPERM");
  if (bpe && atoi(bpe) == 0) e->edge_bperm = 0;
  const char* csp = getenv("DSBDD_COORD_SPLIT");
  if (csp && atoi(csp) == 0) e->coord_split = 0;
  const char* ngp = getenv("DSBDD_NODE_GROUP");
  if (ngp && atoi(ngp) == 0) e->node_group = 0;
  const char* prn = getenv("DSBDD_PRUNE");
  if (prn && atoi(prn) == 0) e->prune = 0;
  if (const char* lh = getenv("DSBDD_LIG_HEAD")) e->lig_head = atoi(lh) != 0;
  if (const char* fs = getenv("DSBDD_FOLD_SCAN")) e->fold_scan = atoi(fs) > 0 ? 1 : 0;
  if (const char* lr = getenv("DSBDD_LEVEL_ROWS")) e->level_rows = atoi(lr) != 0;
  if (const char* fk = getenv("DSBDD_FORK")) e->fork_front = atoi(fk) != 0;
  if (const char* g16 = getenv("DSBDD_GRANULE16")) e->granule16 = (unsigned)strtoul(g16, nullptr, 0);
  if (const char* sk = getenv("DSBDD_SPLITK")) e->splitk = (unsigned)strtoul(sk, nullptr, 0);
  if (const char* em = getenv("DSBDD_EMU")) { const int v = atoi(em); e->emu = (v == 6 || v == 9) ? v : 0; }
  const char* cn = getenv("DSBDD_CONE");
  if (cn) e->cone = atoi(cn) <= 0 ? 0 : (atoi(cn) >= 2 ? 2 : 1);
  const char* nch = getenv("DSBDD_NODE_CHAIN");
  if (nch && atoi(nch) == 0) e->chain = 0;
  const char* ncm = getenv("DSBDD_NODE_CHAIN_MIN_ROWS");
  if (ncm && atoi(ncm) >= 0) e->chain_min_rows = atoi(ncm);
  const char* mwg = getenv("DSBDD_EDGE_MAX_WG");
  if (mwg && atoi(mwg) > 0) e->edge_max_wg = atoi(mwg);
  *out = e;
  return DSBDD_OK;
}

void dsbdd_engine_destroy(dsbdd_engine* e) { delete e; }

int dsbdd_engine_weight_slots(const dsbdd_engine* e) { return e ? n_slots(e->cfg) : DSBDD_ERR_ARG; }

int dsbdd_engine_set_weights(dsbdd_engine* e, const float* const* slots_host, int n) {
  if (!e || !slots_host) return fail(DSBDD_ERR_ARG, "null argument");
  if (n != n_slots(e->cfg)) return fail(DSBDD_ERR_ARG, "wrong number of weight slots");
  const dsbdd_config& c = e->cfg;
  for (int i = 0; i < n; ++i) {
    const float* ptr = slots_host[i];
    bool optional = false;
    // slots that may legitimately be absent
    const int per = c.inv_sublayers * DSBDD_GCL_COUNT + DSBDD_EQ_COUNT;
    if (i >= DSBDD_G_COUNT) {
      const int r = (i - DSBDD_G_COUNT) % per;
      if (r < c.inv_sublayers * DSBDD_GCL_COUNT) {
        const int wch = r % DSBDD_GCL_COUNT;
        if (!c.attention && (wch == DSBDD_GCL_ATT_W || wch == DSBDD_GCL_ATT_B)) optional = true;
      } else {
        const int wch = r - c.inv_sublayers * DSBDD_GCL_COUNT;
        if (c.reflection_equivariant && wch >= DSBDD_EQ_X_WD && wch <= DSBDD_EQ_X_B2) optional = true;
      }
    }
    if (!ptr && !optional) return fail(DSBDD_ERR_ARG, "null weight slot " + std::to_string(i));
    if (ptr && (reinterpret_cast<uintptr_t>(ptr) & 15))
      return fail(DSBDD_ERR_ARG, "weight slot " + std::to_string(i) + " not 16-byte aligned");
    e->slots[i] = ptr;
  }
  e->drop_graphs();
  e->w2tp_ready = false;
  e->w2tp16_ready = false;
  e->w2sk_ready = false;
  e->w2e_ready = false;
  e->wchain_ready = false;
  e->h0_pocket_valid = false;
  e->has_weights = true;
  return DSBDD_OK;
}

size_t dsbdd_engine_workspace_bytes(const dsbdd_engine* e, int64_t nl, int64_t np, int64_t B,
                                    int64_t E) {
  if (!e || nl < 0 || np < 0 || B < 1 || E < 0) return 0;
  return carve(e->cfg, nl, np, B, E).total;
}

int dsbdd_engine_bind_workspace(dsbdd_engine* e, void* ws, size_t bytes, int64_t nl, int64_t np,
                                int64_t B, int64_t E) {
  if (!e || !ws) return fail(DSBDD_ERR_ARG, "null argument");
  if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(DSBDD_ERR_ARG, "workspace must be 256-byte aligned");
  if ((nl + np) >= (1ll << 30) || E >= (1ll << 31) - 256) return fail(DSBDD_ERR_ARG, "problem too large for int32 indices");
  WsLayout L = carve(e->cfg, nl, np, B, E);
  if (bytes < L.total) return fail(DSBDD_ERR_CAPACITY, "workspace too small");
  char* b = static_cast<char*>(ws);
  e->drop_graphs();
  e->ws = b; e->ws_bytes = bytes;
  e->cap_lig = nl; e->cap_poc = np; e->cap_batch = B; e->cap_edges = E;
  e->node_batch = (int*)(b + L.off[0]); e->lig_off = (int*)(b + L.off[1]); e->poc_off = (int*)(b + L.off[2]);
  e->deg = (int*)(b + L.off[3]); e->row_ptr = (int*)(b + L.off[4]);
  e->erow = (int*)(b + L.off[5]); e->ecol = (int*)(b + L.off[6]); e->ed0 = (float*)(b + L.off[7]);
  e->x = (float*)(b + L.off[8]); e->x_in = (float*)(b + L.off[9]); e->xagg = (float*)(b + L.off[10]);
  e->mean = (float*)(b + L.off[11]); e->h0 = (float*)(b + L.off[12]); e->enc_tmp = (float*)(b + L.off[13]);
  e->h = (float*)(b + L.off[14]); e->t1 = (float*)(b + L.off[15]); e->agg = (float*)(b + L.off[16]);
  e->pq = (float*)(b + L.off[17]); e->hout = (float*)(b + L.off[18]);
  e->act_flag = (int*)(b + L.off[19]); e->act_ptr = (int*)(b + L.off[20]); e->act_list = (int*)(b + L.off[21]);
  e->pqg = (float*)(b + L.off[22]);
  e->w2tp = (float*)(b + L.off[23]);
  e->agg_head = (float*)(b + L.off[24]); e->xagg_head = (float*)(b + L.off[25]);
  e->scan_tmp = (int*)(b + L.off[26]); e->seg_base = (int*)(b + L.off[27]); e->tile_ctr = (int*)(b + L.off[28]);
  e->cap_edgesL = 2 * E + 32 * kLevels * B;
  e->cap_tiles = e->cap_edgesL / 32 + 2;
  e->cap_tiles16 = e->cap_edgesL / 16 + 2;   // head slots of the 16-edge-granule kernels (edge_wave16.h); <= 2 * cap_tiles
  e->lvl = (int*)(b + L.off[50]); e->seg_rows = (int*)(b + L.off[51]); e->seg_edges = (int*)(b + L.off[52]);
  e->node_base = (int*)(b + L.off[53]); e->edge_base = (int*)(b + L.off[54]);
  e->lvl_cnt = (int*)(b + L.off[55]); e->lvl_end = (int*)(b + L.off[56]);
  e->lvl_list = (int*)(b + L.off[57]); e->row_ptrL = (int*)(b + L.off[58]);
  e->erowL = (int*)(b + L.off[59]); e->ecolL = (int*)(b + L.off[60]); e->ed0L = (float*)(b + L.off[61]);
  e->lvl_stats = (unsigned long long*)(b + L.off[62]);
  e->lvl_stats_zeroed = false;          // cleared by the first (eager) call that uses it
  e->erow2 = (int*)(b + L.off[29]); e->ecol2 = (int*)(b + L.off[30]); e->ed02 = (float*)(b + L.off[31]);
  e->row_ptr2 = (int*)(b + L.off[32]); e->deg2 = (int*)(b + L.off[33]);
  e->scan_tmp2 = (int*)(b + L.off[34]); e->seg_base2 = (int*)(b + L.off[35]);
  e->erow3 = (int*)(b + L.off[36]); e->ecol3 = (int*)(b + L.off[37]); e->ed03 = (float*)(b + L.off[38]);
  e->row_ptr3 = (int*)(b + L.off[39]); e->deg3 = (int*)(b + L.off[40]);
  e->scan_tmp3 = (int*)(b + L.off[41]); e->seg_base3 = (int*)(b + L.off[42]);
  e->node_batch3 = (int*)(b + L.off[43]); e->lig_off3 = (int*)(b + L.off[44]); e->poc_off3 = (int*)(b + L.off[45]);
  e->twin = (int*)(b + L.off[46]);
  e->xframe = (float*)(b + L.off[47]); e->aggB = (float*)(b + L.off[48]); e->agg_headB = (float*)(b + L.off[49]);
  e->frame_rows = (int*)(b + L.off[63]);
  e->wchain = (float*)(b + L.off[64]);
  e->wchain_ready = false;
  e->frame = false;               // a pocket frame lives in the workspace
  e->ghost_dirty = true;
  e->h0_pocket_valid = false;
  e->w2tp_ready = false;
  e->w2tp16_ready = false;
  e->w2sk_ready = false;
  e->w2e_ready = false;
  return DSBDD_OK;
}

#ifdef DSBDD_TIMESTAMPS
// debug builds only (not part of the ABI header): device buffer of [capacity][64][16] uint64 marks
int dsbdd_debug_set_timestamps(dsbdd_engine* e, unsigned long long* buf, int capacity) {
  if (!e) return DSBDD_ERR_ARG;
  e->ts_buf = buf; e->ts_cap = capacity; e->ts_next = 0;
  return DSBDD_OK;
}
#endif

// Ghost rows N .. N + n3 (the frame's pockets as nodes of their own: coordinates, degrees, positions) and the front
// segment of the level-ordered list, from the pristine frame data (list 3, xframe).  Re-run whenever a call without
// the frame may have written over them.
static int ghost_setup(dsbdd_engine* e, hipStream_t s) {
  const int n3 = (int)e->frame_n3, N = (int)(e->frame_nlig + e->frame_npoc);
  int64_t gb = (e->frame_cap3 + 255) / 256;
  if (gb > 1024) gb = 1024;
  hipLaunchKernelGGL(ghost_setup_kernel, dim3((int)gb), dim3(256), 0, s, (const int*)e->erow3, (const int*)e->ecol3,
                     (const float*)e->ed03, (const int*)e->row_ptr3, (const int*)e->deg3, n3, N, N, e->erowL,
                     e->ecolL, e->ed0L, (int)e->cap_edgesL, e->deg, e->row_ptrL, e->lvl_list,
                     (const float*)e->xframe, e->x);
  HIP_TRY(hipGetLastError());
  e->ghost_dirty = false;
  return DSBDD_OK;
}

int dsbdd_engine_set_pocket_frame(dsbdd_engine* e, void* stream, const float* x_frame, const int64_t* mask_frame,
                                  const int32_t* frame_rows, const int32_t* twin_local, int64_t n_lig,
                                  int64_t n_pocket, int64_t batch, int64_t n_frame, int64_t batch_frame,
                                  int64_t edge_bound_frame) {
  StreamDevice stream_device_(stream);
  if (!e || !x_frame || !mask_frame || !frame_rows || !twin_local) return fail(DSBDD_ERR_ARG, "null argument");
  if (!e->ws) return fail(DSBDD_ERR_STATE, "workspace not bound");
  if (e->cfg.update_pocket_coords) return fail(DSBDD_ERR_STATE, "a pocket frame needs rigid pocket coordinates");
  if (n_lig < 0 || n_lig > e->cap_lig || n_pocket < 1 || n_pocket > e->cap_poc || batch < 1 || batch > e->cap_batch ||
      n_frame < 1 || n_frame > n_pocket || batch_frame < 1 || batch_frame > batch || edge_bound_frame < 1 ||
      edge_bound_frame > e->cap_edges)
    return fail(DSBDD_ERR_CAPACITY, "pocket frame exceeds the bound workspace");
  hipStream_t s = static_cast<hipStream_t>(stream);
  e->drop_graphs();
  e->frame = false;
  e->h0_pocket_valid = false;
  const int n3 = (int)n_frame, b3 = (int)batch_frame, N = (int)(n_lig + n_pocket);
  HIP_TRY(hipMemcpyAsync(e->xframe, x_frame, (size_t)n3 * 12, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->frame_rows, frame_rows, (size_t)n3 * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->twin, twin_local, (size_t)n_pocket * 4, hipMemcpyDeviceToDevice, s));
  // the pocket-pocket radius graph of the frame: a pocket-only problem (no ligand nodes) whose nodes are the
  // ghost rows N .. N + n3 of the engine's node arrays
  const int work = n3 > b3 + 1 ? n3 : b3 + 1;
  hipLaunchKernelGGL(prep_kernel, dim3((work + 255) / 256), dim3(256), 0, s, (const int64_t*)nullptr, 0, mask_frame,
                     n3, b3, e->node_batch3, e->lig_off3, e->poc_off3, (int*)nullptr);
  HIP_TRY(hipGetLastError());
  int rc = build_edges_impl(s, x_frame, 0, n3, b3, e->cfg, e->node_batch3, e->lig_off3, e->poc_off3, e->deg3,
                            e->row_ptr3, e->erow3, e->ecol3, e->ed03, e->cap_edges, e->tile_ctr + 24, nullptr,
                            e->scan_tmp3, e->seg_base3, nullptr, N);
  if (rc) return rc;
  e->frame_nlig = n_lig; e->frame_npoc = n_pocket; e->frame_batch = batch;
  e->frame_n3 = n3; e->frame_cap3 = edge_bound_frame;
  rc = ghost_setup(e, s);
  if (rc) return rc;
  int slots = 0;                         // one host sync per chain (the chain start has one already)
  HIP_TRY(hipMemcpyAsync(&slots, e->row_ptr3 + n3, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (slots < 0 || slots > e->cap_edges || (slots & (kEdgeAlign - 1)))
    return fail(DSBDD_ERR_CAPACITY, "pocket frame: pocket-pocket list exceeds the edge capacity");
  e->ghost_slots = slots;
  e->frame = true;
  return DSBDD_OK;
}

int dsbdd_engine_clear_pocket_frame(dsbdd_engine* e) {
  if (!e) return fail(DSBDD_ERR_ARG, "null argument");
  if (e->frame) e->drop_graphs();
  e->frame = false;
  e->h0_pocket_valid = false;
  return DSBDD_OK;
}

int dsbdd_engine_last_plan(const dsbdd_engine* e, int32_t* radius, int32_t* ghost, int32_t capacity,
                           int32_t* n_stages, int32_t* timed_level) {
  if (!e || !radius || !ghost || !n_stages || !timed_level) return fail(DSBDD_ERR_ARG, "null argument");
  const int n = (int)e->plan_radius.size();
  if (capacity < n) return fail(DSBDD_ERR_CAPACITY, "plan arrays too short");
  for (int i = 0; i < n; ++i) { radius[i] = e->plan_radius[i]; ghost[i] = e->plan_ghost[i]; }
  *n_stages = n; *timed_level = e->plan_timed_level;
  return DSBDD_OK;
}

int dsbdd_engine_set_option(dsbdd_engine* e, int which, int value) {
  if (!e) return fail(DSBDD_ERR_ARG, "null argument");
  switch (which) {
    case DSBDD_OPT_PRUNE: e->prune = value ? 1 : 0; break;
    case DSBDD_OPT_CONE: e->cone = value <= 0 ? 0 : (value >= 2 ? 2 : 1); break;   // 0 off, 1 by the cost model, 2 always
    case DSBDD_OPT_GRANULE16: e->granule16 = (unsigned)value; break;               // bit g: message stage g, bit 16 + b: coordinate stage b
    case DSBDD_OPT_SPLITK: e->splitk = (unsigned)value; break;                     // the same layout: stages on the split-K kernels
    case DSBDD_OPT_EMU:                                                            // 0 exact fp32; 6 / 9: emulated on the bf16 matrix cores
      if (value != 0 && value != 6 && value != 9) return fail(DSBDD_ERR_ARG, "DSBDD_OPT_EMU takes 0, 6 or 9");
      e->emu = value; break;
    default: return fail(DSBDD_ERR_ARG, "unknown option");
  }
  e->drop_graphs();
  return DSBDD_OK;
}

int dsbdd_engine_get_option(const dsbdd_engine* e, int which) {
  if (!e) return fail(DSBDD_ERR_ARG, "null argument");
  switch (which) {
    case DSBDD_OPT_PRUNE: return e->prune;
    case DSBDD_OPT_CONE: return e->cone;
    case DSBDD_OPT_GRANULE16: return (int)e->granule16;
    case DSBDD_OPT_SPLITK: return (int)e->splitk;
    case DSBDD_OPT_EMU: return e->emu;
  }
  return fail(DSBDD_ERR_ARG, "unknown option");
}

int dsbdd_engine_set_trace(dsbdd_engine* e, float* th, float* tx) {
  if (!e) return DSBDD_ERR_ARG;
  e->trace_h = th; e->trace_x = tx;
  return DSBDD_OK;
}

int dsbdd_engine_profile(dsbdd_engine* e, int enable, int max_launches) {
  if (!e || max_launches < 0) return fail(DSBDD_ERR_ARG, "bad argument");
  e->profile = enable < 0 ? 0 : enable;      // 0 off, k: the launches of every k-th forward call are timed
  e->prof_call = 0;
  e->ev_used = 0;
  while (e->profile && e->ev.size() < (size_t)2 * max_launches) {
    hipEvent_t a;
    HIP_TRY(hipEventCreate(&a));
    e->ev.push_back(a);
  }
  return DSBDD_OK;
}

int dsbdd_engine_profile_read(dsbdd_engine* e, double* total_ms, int64_t* launches) {
  if (!e || !total_ms || !launches) return fail(DSBDD_ERR_ARG, "null argument");
  double tot = 0.0;
  for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
    HIP_TRY(hipEventSynchronize(e->ev[i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)(e->ev_used / 2);
  e->ev_used = 0;
  return DSBDD_OK;
}

int dsbdd_engine_graph_stats(const dsbdd_engine* e, int64_t* replays, int64_t* captures, int64_t* eager) {
  if (!e || !replays || !captures || !eager) return fail(DSBDD_ERR_ARG, "null argument");
  *replays = e->n_replay; *captures = e->n_capture; *eager = e->n_eager;
  return DSBDD_OK;
}

int dsbdd_engine_buffer(const dsbdd_engine* e, int which, void** out) {
  if (!e || !out || !e->ws) return fail(DSBDD_ERR_STATE, "no workspace bound");
  switch (which) {
    case DSBDD_BUF_EDGE_ROW: *out = e->erow; break;
    case DSBDD_BUF_EDGE_COL: *out = e->ecol; break;
    case DSBDD_BUF_EDGE_D0: *out = e->ed0; break;
    case DSBDD_BUF_ROW_PTR: *out = e->row_ptr; break;
    case DSBDD_BUF_H: *out = e->h; break;
    case DSBDD_BUF_X: *out = e->x; break;
    case DSBDD_BUF_NODE_BATCH: *out = e->node_batch; break;
    case DSBDD_BUF_DEG: *out = e->deg; break;
    case DSBDD_BUF_LEVEL: *out = e->lvl; break;
    case DSBDD_BUF_LEVEL_LIST: *out = e->lvl_list; break;
    case DSBDD_BUF_LEVEL_COUNT: *out = e->lvl_cnt; break;
    case DSBDD_BUF_LEVEL_END: *out = e->lvl_end; break;
    case DSBDD_BUF_LROW_PTR: *out = e->row_ptrL; break;
    case DSBDD_BUF_LEDGE_ROW: *out = e->erowL; break;
    case DSBDD_BUF_LEDGE_COL: *out = e->ecolL; break;
    case DSBDD_BUF_LEDGE_D0: *out = e->ed0L; break;
    case DSBDD_BUF_LEVEL_STATS: *out = e->lvl_stats; break;
    default: return fail(DSBDD_ERR_ARG, "unknown buffer id");
  }
  return DSBDD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
static hipError_t nl(hipStream_t s, const float* A1, int lda1, int K1, const float* A2, int lda2, int K2,
                     const float* WT, int ldw, const float* bias, const float* R, int ldr, float* C,
                     int ldc, int64_t M, int N, int act) {
  NodeLinearArgs a{A1, lda1, K1, A2, lda2, K2, WT, ldw, bias, R, ldr, C, ldc, (int)M, N, act, nullptr, nullptr};
  return launch_node_linear(s, a);
}

// same, over the gathered row subset row_idx[0 .. *m_count)
static hipError_t nl_rows(hipStream_t s, const float* A1, int lda1, int K1, const float* WT, int ldw, float* C,
                          int ldc, int64_t M_cap, int N, const int* row_idx, const int* m_count) {
  NodeLinearArgs a{A1, lda1, K1, nullptr, 0, 0, WT, ldw, nullptr, nullptr, 0, C, ldc, (int)M_cap, N, 0, row_idx, m_count};
  return launch_node_linear(s, a);
}

template <int H>
static hipError_t launch_wave_emu_t(hipStream_t s, int mode, const EdgeArgs& a, int grid, int emu) {
  if (emu == 9) {
    if (mode == MODE_GCL) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, false, 9>), dim3(grid), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, false, 9>), dim3(grid), dim3(kThreads), 0, s, a);
  } else {
    if (mode == MODE_GCL) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, false, 6>), dim3(grid), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, false, 6>), dim3(grid), dim3(kThreads), 0, s, a);
  }
  return hipGetLastError();
}

template <int H>
static hipError_t launch_wave_t(hipStream_t s, int mode, const EdgeArgs& a, int grid) {
  // lane-grouped W2^T copies present (EdgeMlpW::W2TP): 16-byte B-operand reads
  constexpr bool can_perm = (H == 256 || H == 128);
  if constexpr (can_perm) {
    if (a.mlp[0].W2TP && a.mlp[1].W2TP) {
      if (mode == MODE_GCL)
        hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, true>), dim3(grid), dim3(kThreads), 0, s, a);
      else
        hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, true>), dim3(grid), dim3(kThreads), 0, s, a);
      return hipGetLastError();
    }
  }
  if (mode == MODE_GCL)
    hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, false>), dim3(grid), dim3(kThreads), 0, s, a);
  else
    hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, false>), dim3(grid), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}

template <int H>
static hipError_t launch_wave16_t(hipStream_t s, int mode, const EdgeArgs& a, int grid) {
  if (mode == MODE_GCL) hipLaunchKernelGGL((edge_wave16_kernel<H, MODE_GCL>), dim3(grid), dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((edge_wave16_kernel<H, MODE_COORD>), dim3(grid), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}

// 16-edge-granule variant (edge_wave16.h): 64-edge workgroup items, one per (tile, MLP), persistent over 2 workgroups per CU
static hipError_t launch_edge16(const dsbdd_engine* e, hipStream_t s, int mode, const EdgeArgs& a, int64_t edge_bound) {
  const int H = e->cfg.hidden_nf;
  int64_t items = (edge_bound + 63) / 64 * (mode == MODE_COORD ? a.n_mlp : 1);
  int64_t resident = 2LL * e->n_cu;
  if (e->edge_max_wg > 0 && e->edge_max_wg < resident) resident = e->edge_max_wg;
  int grid = (int)(items < resident ? items : resident);
  if (grid < 1) grid = 1;
  switch (H) {
    case 64: return launch_wave16_t<64>(s, mode, a, grid);
    case 128: return launch_wave16_t<128>(s, mode, a, grid);
    case 192: return launch_wave16_t<192>(s, mode, a, grid);
    case 256: return launch_wave16_t<256>(s, mode, a, grid);
  }
  return hipErrorInvalidValue;
}

// split-K variant (edge_splitk.h): one workgroup item per (32-edge tile, MLP), persistent over 2 workgroups per CU
static hipError_t launch_edge_sk(const dsbdd_engine* e, hipStream_t s, int mode, const EdgeArgs& a, int64_t edge_bound) {
  const bool two = mode == MODE_COORD && a.n_mlp == 2;
  int64_t items = (edge_bound + 31) / 32 * (two ? 2 : 1);
  int64_t resident = 2LL * e->n_cu;
  if (e->edge_max_wg > 0 && e->edge_max_wg < resident) resident = e->edge_max_wg;
  if (items > resident) items = resident;
  const int q8 = two ? 16 : 8;                        // 8 XCDs (x 2 MLPs)
  int grid = (int)((items + q8 - 1) / q8 * q8);
  if (grid < q8) grid = q8;
  if (e->cfg.hidden_nf != 256) return hipErrorInvalidValue;
  if (mode == MODE_GCL) hipLaunchKernelGGL((edge_splitk_kernel<256, MODE_GCL>), dim3(grid), dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((edge_splitk_kernel<256, MODE_COORD>), dim3(grid), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}

static hipError_t launch_edge(const dsbdd_engine* e, hipStream_t s, int mode, const EdgeArgs& a,
                              int64_t edge_bound, bool g16 = false, bool sk = false) {
  if (sk && a.mlp[0].W2SK) return launch_edge_sk(e, s, mode, a, edge_bound);
  if (g16) return launch_edge16(e, s, mode, a, edge_bound);
  const int H = e->cfg.hidden_nf;
  // 128-edge workgroup tiles (4 waves x 32 edges), 2 workgroups per CU, persistent over tiles
  int64_t tiles = (edge_bound + 127) / 128;
  int64_t resident = 2LL * e->n_cu;
  if (e->edge_max_wg > 0 && e->edge_max_wg < resident) resident = e->edge_max_wg;
  const bool split = mode == MODE_COORD && a.pass_split && a.n_mlp == 2;
  int64_t g = split ? 2 * tiles : tiles;             // split: one workgroup per (tile, MLP)
  if (g > resident) g = resident;
  const int q8 = split ? 16 : 8;                      // 8 XCDs (x 2 MLPs)
  int grid = (int)((g + q8 - 1) / q8 * q8);
  if (grid < q8) grid = q8;
  if (e->emu && a.mlp[0].W2E && a.mlp[1].W2E) {   // fp32 emulated on the bf16 matrix cores (engine option, opt-in)
    switch (H) {
      case 64: return launch_wave_emu_t<64>(s, mode, a, grid, e->emu);
      case 128: return launch_wave_emu_t<128>(s, mode, a, grid, e->emu);
      case 192: return launch_wave_emu_t<192>(s, mode, a, grid, e->emu);
      case 256: return launch_wave_emu_t<256>(s, mode, a, grid, e->emu);
    }
  }
  switch (H) {
    case 64: return launch_wave_t<64>(s, mode, a, grid);
    case 128: return launch_wave_t<128>(s, mode, a, grid);
    case 192: return launch_wave_t<192>(s, mode, a, grid);
    case 256: return launch_wave_t<256>(s, mode, a, grid);
  }
  return hipErrorInvalidValue;
}

static Cutoffs cutoffs_of(const dsbdd_config& c) {
  return Cutoffs{c.has_cutoff_ligand, c.has_cutoff_pocket, c.has_cutoff_interaction,
                 c.cutoff_ligand, c.cutoff_pocket, c.cutoff_interaction};
}

static int build_edges_impl(hipStream_t s, const float* x, int n_lig, int N, int B, const dsbdd_config& c,
                            const int* node_batch, const int* lig_off, const int* poc_off, int* deg,
                            int* row_ptr, int* erow, int* ecol, float* ed0, int64_t cap, int* status,
                            int* act_flag, int* scan_tmp, int* seg_base, const EdgeList2* list2, int id_offset,
                            int* lvl) {
  const int waves_per_block = kThreads / 64;
  int blocks = (N + waves_per_block - 1) / waves_per_block;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const Cutoffs cut = cutoffs_of(c);
  // with scan_tmp / seg_base: every (sample, node set) segment of the edge list starts at a wave-tile
  // boundary (graph.h scan_kernel); without: a compact list (the public dsbdd_build_edges)
  const int aligned = scan_tmp && seg_base;
  SegAlign sg{node_batch, lig_off, poc_off, n_lig, B, scan_tmp, aligned ? seg_base : nullptr};
  EdgeList2 l2{};
  if (list2 && aligned) l2 = *list2;
  hipLaunchKernelGGL((edges_kernel<false>), dim3(blocks), dim3(kThreads), 0, s, x, node_batch, lig_off,
                     poc_off, n_lig, N, cut, deg, (const int*)nullptr, (int*)nullptr, (int*)nullptr,
                     (float*)nullptr, 0, status, act_flag, SegAlign{}, (int*)nullptr, l2, 0, lvl);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)deg, row_ptr, N, sg,
                     (const int*)l2.deg, l2.row_ptr, l2.seg);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL((edges_kernel<true>), dim3(blocks), dim3(kThreads), 0, s, x, node_batch, lig_off,
                     poc_off, n_lig, N, cut, deg, (const int*)row_ptr, erow, ecol, ed0, (int)cap, status,
                     (int*)nullptr, sg, row_ptr, l2, id_offset, (int*)nullptr);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

// The launch sequence of one EGNNDynamics.forward (enqueue only).
static int forward_impl(dsbdd_engine* e, hipStream_t s, const float* xh_lig, const float* xh_pocket,
                           const float* t, int64_t t_count, const int64_t* mask_lig,
                           const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                           const int32_t* ext_row, const int32_t* ext_col, int64_t ext_n_edges,
                           float* eps_lig, float* eps_pocket, int32_t* status) {
  const bool ext = ext_row != nullptr;
  const dsbdd_config& c = e->cfg;
  const int H = c.hidden_nf, J = c.joint_nf, JP = pad4(J + 1);
  const int a = c.atom_nf, r = c.residue_nf, dl = 3 + a, dp = 3 + r;
  const int LE = pad4(2 * (a > r ? a : r));
  const int nlig = (int)n_lig, N = (int)(n_lig + n_pocket), B = (int)batch;
  const int n_mlp = c.reflection_equivariant ? 1 : 2;
  const int PQ = (c.reflection_equivariant ? 2 : 4) * H;
  const float* const* W = e->slots.data();
  if (N == 0) return DSBDD_OK;
  // pocket-conditioning mode: the coordinate MLPs only touch edges whose row is a ligand
  // node, so their first-layer projections are needed for a subset of the nodes only
  const bool subset = !c.update_pocket_coords;

  // Pocket frame (pocket-conditioning chains): block 0's first message stage runs on the edges with a
  // ligand endpoint; the pocket-pocket part comes from the static list built by set_pocket_frame.
  const bool split0 = e->frame && subset && !ext && n_lig == e->frame_nlig && n_pocket == e->frame_npoc &&
                      batch == e->frame_batch;
  // Ligand output only (eps_pocket == nullptr) in pocket-conditioning mode: the stages evaluate the rows the
  // ligand output depends on, prefixes of the level-ordered list (graph.h, "Level-ordered edge list")
  const bool prune = e->prune && subset && !ext && !eps_pocket && !e->trace_h && !e->trace_x && nlig > 0;
  const int G_stages = c.n_layers * c.inv_sublayers;
  // Forward cone (identical pockets, one t for the batch): after message stage g only the nodes within g + 1 hops of
  // a ligand node can differ from the ligand-free ("canonical") pocket network, which is evaluated once, on the ghost
  // rows N .. N + n_ghost (its stage-0 messages are the frame's pocket-pocket launch).  Stage g then computes the
  // rows of level <= min(g + 1, G - g); the rows the next stage reads beyond those get the canonical values.
  // Cost model: the canonical network is extra work -- its G/2 ascending stages on every ghost row (the frame's pockets:
  // frame_n3 rows, one per group of identical pockets) -- against the rows the cone's first stages skip.  On the
  // benchmark pocket the skipped part is (1 - 0.28) + (1 - 0.69) + (1 - 0.97) = 1.06 edge lists per call and the
  // pocket-pocket edges of the ghosts are 0.83 of a list per stage, so by edge counts the cone pays while the frame holds less
  // than 1.06 / (3 x 0.83) = 0.43 of the batch's pocket rows: this rule (option value 1) is kept for direct C-API callers.
  // MEASURED at B = 64 (profiles/r4n_cone_rule.md) the break-even is lower -- 12 distinct pockets of 64: 8.8 ms per chain
  // and ghost pocket (the ascending stages' short launches run at 0.63 instead of 0.75 of the peak) against 1.7 ms saved
  // per sample -- and a size-dependent rule inside the engine would let a batch and its half take different modes; the
  // DDPM modules therefore decide per chain from the pocket groups (5 groups <= batch) and pass 0 / 2.
  const bool cone_pays = e->cone >= 2 || 5 * e->frame_n3 <= 2 * (int64_t)n_pocket;
  const bool cone = prune && split0 && e->cone && cone_pays && t_count == 1 && G_stages >= 2;
  // with a frame, the frame's pockets are the ghost rows N .. N + n_frame_rows; in the level-ordered list they own the
  // first n_ghost entries of lvl_list and the first ghost_slots edge slots
  const int n_frame_rows = split0 ? (int)e->frame_n3 : 0;
  const int n_ghost = (split0 && prune) ? n_frame_rows : 0;
  const int64_t ghost_slots = (split0 && prune) ? e->ghost_slots : 0;
  // (the ghost rows are kept valid by dsbdd_dynamics_forward, which also covers the calls that replay a graph)
  constexpr int LV = kLevels - 1;                       // "everything"
  auto radius_of = [&](int g) {
    const int bw = G_stages - g, fw = g + 1;
    const int r = cone ? (bw < fw ? bw : fw) : bw;
    return r < LV ? r : LV;
  };
  // ghost rows are evaluated while a later stage still reads canonical values: the ascending part of the radii
  int g_ghost_last = -1;
  if (cone)
    for (int g = 0; g + 1 < G_stages; ++g) {
      const int rd = radius_of(g + 1) + 1 < LV ? radius_of(g + 1) + 1 : LV;
      if (rd > radius_of(g)) g_ghost_last = g;
    }
  e->plan_radius.assign(G_stages, LV); e->plan_ghost.assign(G_stages, 0);
  e->plan_timed_level = LV;
  if (prune) {
    int rt = 0;
    for (int g = 0; g < G_stages; ++g) {
      e->plan_radius[g] = radius_of(g); e->plan_ghost[g] = g <= g_ghost_last;
      if (!(split0 && g == 0) && radius_of(g) > rt) rt = radius_of(g);
    }
    e->plan_timed_level = rt;
  }
  // ---- masks -> offsets, split inputs ---------------------------------------
  {
    int work = N > B + 1 ? N : B + 1;
    if (work < 2 * B) work = 2 * B;
    hipLaunchKernelGGL(prep_assemble_kernel, dim3((work + 255) / 256), dim3(256), 0, s, mask_lig, nlig, mask_pocket,
                       (int)n_pocket, B, e->node_batch, e->lig_off, e->poc_off, e->tile_ctr, xh_lig, dl, xh_pocket, dp,
                       t, (int)t_count, e->x, e->x_in, e->h0, J, JP);
    HIP_TRY(hipGetLastError());
  }
  // ---- two independent chains at the head of a call: A = encoders -> embedding (-> ghost-row features), needs only the
  // assembled inputs; B = radius graph -> scan -> fill -> hop levels -> level-ordered list, needs only the coordinates.
  // Both are strings of short latency-bound kernels (A: 46 us, B: 78 us per call at the benchmark size).  Round 4
  // experiment, DSBDD_FORK=1: A on an engine-owned side stream, forked and joined by events -- inside a captured graph the
  // two become parallel branches.  Parity-green (146 GPU tests) and measured SLOWER: 37.56 vs 38.00 ligands/s (full-atom),
  // 52.7 vs 55.0 (C-alpha), i.e. +40 us per call: a fork / join inside a replayed graph costs more than the 46 us of
  // serial kernels it hides (profiles/r4g_fork_ab.md; the same finding as round 2's second-stream experiment).  Off.
  const bool fork = e->fork_front && !ext;
  hipStream_t sa = s;
  if (fork) {
    if (!e->side_stream) HIP_TRY(hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking));
    if (!e->ev_fork) HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    if (!e->ev_join) HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
    sa = e->side_stream;
  }
  // ---- encoders (dynamics.py:96-97) -> h0[:, 0:J] ----------------------------
  {
    Mlp2Problem enc[2] = {
        {xh_lig + 3, dl, a, W[DSBDD_G_ATOM_ENC_W0T], pad4(2 * a), W[DSBDD_G_ATOM_ENC_B0], 2 * a,
         W[DSBDD_G_ATOM_ENC_W1T], pad4(J), W[DSBDD_G_ATOM_ENC_B1], J, e->h0, JP, (int)n_lig},
        {xh_pocket + 3, dp, r, W[DSBDD_G_RES_ENC_W0T], pad4(2 * r), W[DSBDD_G_RES_ENC_B0], 2 * r,
         W[DSBDD_G_RES_ENC_W1T], pad4(J), W[DSBDD_G_RES_ENC_B1], J, e->h0 + (size_t)n_lig * JP, JP, (int)n_pocket}};
    if (mlp2_fits(enc[0]) && mlp2_fits(enc[1])) {
      // both node sets, both layers: one launch.  With a pocket frame the pocket's encoding is a constant of the chain:
      // dsbdd_dynamics_forward computed it before this call (eagerly, so that replayed graphs find it too)
      HIP_TRY(launch_mlp2(sa, enc, (split0 && e->h0_pocket_valid) ? 1 : 2));
    } else {
      HIP_TRY(nl(sa, xh_lig + 3, dl, a, nullptr, 0, 0, W[DSBDD_G_ATOM_ENC_W0T], pad4(2 * a), W[DSBDD_G_ATOM_ENC_B0],
                 nullptr, 0, e->enc_tmp, LE, n_lig, 2 * a, 1));
      HIP_TRY(nl(sa, e->enc_tmp, LE, 2 * a, nullptr, 0, 0, W[DSBDD_G_ATOM_ENC_W1T], pad4(J),
                 W[DSBDD_G_ATOM_ENC_B1], nullptr, 0, e->h0, JP, n_lig, J, 0));
      float* tmp_p = e->enc_tmp + (size_t)n_lig * LE;
      HIP_TRY(nl(sa, xh_pocket + 3, dp, r, nullptr, 0, 0, W[DSBDD_G_RES_ENC_W0T], pad4(2 * r), W[DSBDD_G_RES_ENC_B0],
                 nullptr, 0, tmp_p, LE, n_pocket, 2 * r, 1));
      HIP_TRY(nl(sa, tmp_p, LE, 2 * r, nullptr, 0, 0, W[DSBDD_G_RES_ENC_W1T], pad4(J), W[DSBDD_G_RES_ENC_B1],
                 nullptr, 0, e->h0 + (size_t)n_lig * JP, JP, n_pocket, J, 0));
    }
  }
  // ---- embedding (egnn_new.py:233) ---------------------------------------------
  HIP_TRY(nl(sa, e->h0, JP, JP, nullptr, 0, 0, W[DSBDD_G_EMB_WT], H, W[DSBDD_G_EMB_B], nullptr, 0, e->h, H, N, H, 0));
  if (split0) {   // the ghost rows start from the embedded features of the pockets they stand for
    hipLaunchKernelGGL(gather_rows_kernel, dim3((e->frame_n3 + 3) / 4), dim3(kThreads), 0, sa, e->h + (size_t)N * H,
                       (const float*)(e->h + (size_t)nlig * H), (const int*)e->frame_rows, (int)e->frame_n3, H);
    HIP_TRY(hipGetLastError());
  }
  if (fork) HIP_TRY(hipEventRecord(e->ev_join, sa));
  // ---- edges (dynamics.py:114, 169-187) ---------------------------------------
  int64_t edge_bound = e->cap_edges;
  bool mean_in_levels = false;
  if (ext) {
    HIP_TRY(zero_async(e->deg, (size_t)N * 4, s));
    if (ext_n_edges > 0) {
      hipLaunchKernelGGL(ext_edges_kernel, dim3((int)((ext_n_edges + 255) / 256)), dim3(256), 0, s, ext_row,
                         ext_col, (int)ext_n_edges, (const float*)e->x, e->erow, e->ecol, e->ed0, e->deg);
      HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)e->deg, e->row_ptr, N, SegAlign{},
                       (const int*)nullptr, (int*)nullptr, SegAlign{});
    HIP_TRY(hipGetLastError());
    edge_bound = ext_n_edges > 0 ? ext_n_edges : 1;
    if (subset) {
      hipLaunchKernelGGL(ext_flags_init_kernel, dim3((N + 255) / 256), dim3(256), 0, s, e->act_flag, nlig, N);
      HIP_TRY(hipGetLastError());
      if (ext_n_edges > 0) {
        hipLaunchKernelGGL(ext_flags_kernel, dim3((int)((ext_n_edges + 255) / 256)), dim3(256), 0, s, ext_row,
                           ext_col, (int)ext_n_edges, nlig, e->act_flag);
        HIP_TRY(hipGetLastError());
      }
    }
  } else {
    EdgeList2 l2{e->deg2, e->row_ptr2, e->erow2, e->ecol2, e->ed02, (int)e->cap_edges,
                 SegAlign{e->node_batch, e->lig_off, e->poc_off, nlig, B, e->scan_tmp2, e->seg_base2}};
    int rc = build_edges_impl(s, e->x, nlig, N, B, c, e->node_batch, e->lig_off, e->poc_off, e->deg,
                              e->row_ptr, e->erow, e->ecol, e->ed0, e->cap_edges, status,
                              subset ? e->act_flag : nullptr, e->scan_tmp, e->seg_base,
                              split0 ? &l2 : nullptr, 0, prune ? e->lvl : nullptr);
    if (rc) return rc;
    if (prune) {
      LevelArgs la{e->node_batch, e->lig_off, e->poc_off, nlig, B, e->lvl, e->deg, e->row_ptr, e->erow, e->ecol,
                   e->ed0, e->seg_rows, e->seg_edges, e->node_base, e->edge_base, e->lvl_cnt, e->lvl_end,
                   e->lvl_list, e->row_ptrL, e->erowL, e->ecolL, e->ed0L, (int)e->cap_edgesL, e->lvl_stats,
                   n_ghost, (int)ghost_slots, (int)e->cap_edges, nullptr, nullptr};
      // block 0's per-sample mean (coord2cross) rides in the levels launch: one launch less per pruned call
      mean_in_levels = e->fold_scan && n_mlp == 2;
      if (mean_in_levels) { la.mean_x = e->x; la.mean_out = e->mean; }
      if (!e->lvl_stats_zeroed) {
        HIP_TRY(zero_async(e->lvl_stats, 128, s));
        e->lvl_stats_zeroed = true;
      }
      hipLaunchKernelGGL(levels_kernel, dim3(B), dim3(kThreads), 0, s, la);
      HIP_TRY(hipGetLastError());
      if (!e->fold_scan) {
        hipLaunchKernelGGL(level_scan_kernel, dim3(1), dim3(1024), 0, s, la, N);
        HIP_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(level_place_kernel, dim3(B), dim3(kThreads), 0, s, la, e->fold_scan ? 1 : 0);
      HIP_TRY(hipGetLastError());
      int64_t cb = (e->cap_edges + 255) / 256;
      if (cb > 2048) cb = 2048;
      if (cb < 1) cb = 1;
      hipLaunchKernelGGL(level_copy_kernel, dim3((int)cb), dim3(256), 0, s, la, N);
      HIP_TRY(hipGetLastError());
    }
  }
  // the list the stages after block 0's split run on
  const int* L_row = prune ? e->erowL : e->erow;
  const int* L_col = prune ? e->ecolL : e->ecol;
  const float* L_d0 = prune ? e->ed0L : e->ed0;
  const int* L_ptr = prune ? e->row_ptrL : e->row_ptr;
  const int L_cap = prune ? (int)e->cap_edgesL : (int)e->cap_edges;
  const int64_t L_bound = prune ? e->cap_edgesL : edge_bound;
  // rows of the nodes of level <= r (r >= kLevels - 1: everything), with or without the ghost rows in front
  // (round 4 experiment, DSBDD_LEVEL_ROWS=1: the all-row stages of a pruned call walk the level list as well -- a
  //  permutation of the rows -- so that the active nodes and the ligand rows are PREFIXES of every stage's row list and the
  //  coordinate projections ride in the node-phase launch of every stage, not only of the radius-limited ones: 3 launches
  //  fewer per call on the C-alpha and mixed-pocket plans [4,4,4,3,2,1].  Measured 0.5 % SLOWER on both
  //  (profiles/r4f_ab.md: the grouped node GEMM launch beats the chain's projection passes); off by default)
  auto rows_of = [&](int r, bool ghost, NodeLinearArgs& a) {
    if (!prune || (r >= LV && !ghost && !e->level_rows)) return;
    if (r > LV) r = LV;
    a.row_idx = ghost ? e->lvl_list : e->lvl_list + n_ghost;
    a.m_count = ghost ? e->lvl_cnt + r : e->lvl_cnt + kLevels + r;
    a.M = N + n_ghost;
  };
  // active nodes of the coordinate projections (ligand nodes + pocket nodes with a ligand neighbour) = the nodes of
  // level <= 1: with the level list they are its prefix (after the ghost entries), otherwise a scan + compaction
  const int* act_rows = prune ? e->lvl_list + n_ghost : e->act_list;
  const int* act_count = prune ? e->lvl_cnt + kLevels + 1 : e->act_ptr + N;
  if (subset && !prune) {   // sorted list of active nodes; its length stays on the device (act_ptr[N])
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)e->act_flag, e->act_ptr, N, SegAlign{},
                       (const int*)nullptr, (int*)nullptr, SegAlign{});
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(compact_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const int*)e->act_flag,
                       (const int*)e->act_ptr, e->act_list, N);
    HIP_TRY(hipGetLastError());
  }
  // (the embedding and the ghost rows' features were enqueued with the encoders, on the side stream: join)
  if (fork) HIP_TRY(hipStreamWaitEvent(s, e->ev_join, 0));

  const int n_upd = c.update_pocket_coords ? N : nlig;   // update_coords_mask, dynamics.py:130-132
  const int* e_all = prune ? e->lvl_end + kLevels + LV : e->row_ptr + N;   // (counted from the end of the ghost segment)
  const int* e_upd = prune ? e->lvl_end + kLevels : e->row_ptr + n_upd;    // edges are row-sorted: a prefix (level 0 = ligand rows)

  // lane-grouped copies of the three W2^T matrices of every block (see EdgeMlpW::W2TP)
  const bool bperm = e->edge_bperm && (H == 256 || H == 128);
  auto w2tp_of = [&](int blk, int which) -> const float* {   // which: 0 .. inv_sublayers-1 GCL, then coord, cross
    return bperm ? e->w2tp + ((size_t)blk * (c.inv_sublayers + 2) + which) * H * H : nullptr;
  };
  const size_t n_w2 = (size_t)c.n_layers * (c.inv_sublayers + 2);
  auto w2tp16_of = [&](int blk, int which) -> const float* {
    return e->granule16 ? e->w2tp + (n_w2 + (size_t)blk * (c.inv_sublayers + 2) + which) * H * H : nullptr;
  };
  auto w2e_of = [&](int blk, int which) -> const void* {     // bf16 planes behind the two fp32 copies: 6 H^2 bytes each
    return e->emu ? reinterpret_cast<const char*>(e->w2tp + 2 * n_w2 * H * H) + ((size_t)blk * (c.inv_sublayers + 2) + which) * 6 * H * H
                  : nullptr;
  };
  const bool can_sk = H == 256 && !e->emu;       // (edge_splitk.h: hidden_nf 256; no emulated form -- the mask is ignored with DSBDD_OPT_EMU)
  auto w2sk_of = [&](int blk, int which) -> const float* {   // behind the bf16 planes: byte offset n_w2 H^2 14
    return (e->splitk && can_sk) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(e->w2tp) + n_w2 * H * H * 14) +
                                       ((size_t)blk * (c.inv_sublayers + 2) + which) * H * H
                                 : nullptr;
  };
  if (e->splitk && can_sk && !e->w2sk_ready) {
    for (int blk = 0; blk < c.n_layers; ++blk)
      for (int which = 0; which < c.inv_sublayers + 2; ++which) {
        const float* src = which < c.inv_sublayers ? W[gcl_slot(c, blk, which, DSBDD_GCL_E2_WT)]
                         : W[eq_slot(c, blk, which == c.inv_sublayers ? DSBDD_EQ_C_W2T : DSBDD_EQ_X_W2T)];
        if (!src) continue;
        hipLaunchKernelGGL(pack_w2sk_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, src,
                           const_cast<float*>(w2sk_of(blk, which)), H);
        HIP_TRY(hipGetLastError());
      }
    e->w2sk_ready = true;
  }
  if (e->emu && !e->w2e_ready) {
    for (int blk = 0; blk < c.n_layers; ++blk)
      for (int which = 0; which < c.inv_sublayers + 2; ++which) {
        const float* src = which < c.inv_sublayers ? W[gcl_slot(c, blk, which, DSBDD_GCL_E2_WT)]
                         : W[eq_slot(c, blk, which == c.inv_sublayers ? DSBDD_EQ_C_W2T : DSBDD_EQ_X_W2T)];
        if (!src) continue;
        hipLaunchKernelGGL(pack_w2e_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, src,
                           reinterpret_cast<unsigned short*>(const_cast<void*>(w2e_of(blk, which))), H);
        HIP_TRY(hipGetLastError());
      }
    e->w2e_ready = true;
  }
  if (e->granule16 && !e->w2tp16_ready) {
    for (int blk = 0; blk < c.n_layers; ++blk)
      for (int which = 0; which < c.inv_sublayers + 2; ++which) {
        const float* src = which < c.inv_sublayers ? W[gcl_slot(c, blk, which, DSBDD_GCL_E2_WT)]
                         : W[eq_slot(c, blk, which == c.inv_sublayers ? DSBDD_EQ_C_W2T : DSBDD_EQ_X_W2T)];
        if (!src) continue;
        hipLaunchKernelGGL(pack16_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, src,
                           const_cast<float*>(w2tp16_of(blk, which)), H);
        HIP_TRY(hipGetLastError());
      }
    e->w2tp16_ready = true;
  }
  if (bperm && !e->w2tp_ready) {
    for (int blk = 0; blk < c.n_layers; ++blk)
      for (int which = 0; which < c.inv_sublayers + 2; ++which) {
        const float* src = which < c.inv_sublayers ? W[gcl_slot(c, blk, which, DSBDD_GCL_E2_WT)]
                         : W[eq_slot(c, blk, which == c.inv_sublayers ? DSBDD_EQ_C_W2T : DSBDD_EQ_X_W2T)];
        if (!src) continue;                                   // reflection-equivariant models have no cross MLP
        hipLaunchKernelGGL(permute_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, s, src,
                           const_cast<float*>(w2tp_of(blk, which)), H);
        HIP_TRY(hipGetLastError());
      }
    e->w2tp_ready = true;
  }
  // packed weights of the row-owning node-phase kernel: per (block, sublayer) node MLP layer 1 [2H -> H], layer 2
  // [H -> H] and the message stage's first-layer projection [H -> 2H]; per block the coordinate projections [H -> PQ]
  const bool use_chain = e->chain && (H == 256 || H == 192 || H == 128) && N >= e->chain_min_rows;
  const size_t chain_blk = (size_t)c.inv_sublayers * 5 * H * H + (size_t)H * PQ;
  auto chain_w = [&](int blk, int sub, int which) -> const float* {   // which: 0 N1, 1 N2, 2 E1 (P|Q), 3 coordinate (sub ignored)
    const float* base = e->wchain + (size_t)blk * chain_blk;
    if (which == 3) return base + (size_t)c.inv_sublayers * 5 * H * H;
    return base + (size_t)sub * 5 * H * H + (which == 0 ? 0 : (which == 1 ? 2 * H * H : 3 * H * H));
  };
  if (use_chain && !e->wchain_ready) {
    auto pack = [&](const float* WT, int ldw, int K, int Ncols, const float* dst) {
      hipLaunchKernelGGL(pack_b16_kernel, dim3((K * Ncols + 255) / 256), dim3(256), 0, s, WT, ldw, K, Ncols,
                         const_cast<float*>(dst));
    };
    for (int blk = 0; blk < c.n_layers; ++blk) {
      for (int sub = 0; sub < c.inv_sublayers; ++sub) {
        pack(W[gcl_slot(c, blk, sub, DSBDD_GCL_N1_WT)], H, 2 * H, H, chain_w(blk, sub, 0));
        pack(W[gcl_slot(c, blk, sub, DSBDD_GCL_N2_WT)], H, H, H, chain_w(blk, sub, 1));
        pack(W[gcl_slot(c, blk, sub, DSBDD_GCL_E1_WT)], 2 * H, H, 2 * H, chain_w(blk, sub, 2));
      }
      pack(W[eq_slot(c, blk, DSBDD_EQ_C1_WT)], PQ, H, PQ, chain_w(blk, 0, 3));
    }
    HIP_TRY(hipGetLastError());
    e->wchain_ready = true;
  }
  auto gcl_pq = [&](int blk, int sub) {
    NodeLinearArgs a{e->h, H, H, nullptr, 0, 0, W[gcl_slot(c, blk, sub, DSBDD_GCL_E1_WT)], 2 * H, nullptr,
                     nullptr, 0, e->pqg, 2 * H, (int)N, 2 * H, 0, nullptr, nullptr};
    const int g = blk * c.inv_sublayers + sub;
    rows_of(radius_of(g) + 1, g <= g_ghost_last && g > 0, a);    // the stage reads its neighbours one level out
    return a;
  };
  bool pqg_ready = false, chained_pq = false, chained_coord = false;
  for (int blk = 0; blk < c.n_layers; ++blk) {
    if (n_mlp == 2 && (blk == 0 || !subset) && !(blk == 0 && mean_in_levels)) {   // coord2cross needs the per-sample mean of the block's input x
                                                 // (pocket-conditioning mode, later blocks: computed by the
                                                 // previous block's coordinate update)
      hipLaunchKernelGGL(sample_mean_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)e->x,
                         (const int*)e->lig_off, (const int*)e->poc_off, nlig, e->mean);
      HIP_TRY(hipGetLastError());
    }
    for (int sub = 0; sub < c.inv_sublayers; ++sub) {
      auto G = [&](int which) { return W[gcl_slot(c, blk, sub, which)]; };
      // P | Q projections of the edge MLP's first layer (those of a block's first sublayer
      // were launched together with the previous block's coordinate projections)
      if (sub > 0 && chained_pq) pqg_ready = true;           // produced by the previous sublayer's node-phase launch
      if (!pqg_ready) {
        const bool rows0 = split0 && blk == 0 && sub == 0;
        NodeLinearArgs grp0[2];
        if (rows0) {
          // pocket frame: block 0 reads P|Q only at the active nodes (ligand nodes + pocket nodes with a ligand
          // neighbour: the endpoints of the ligand-endpoint list) and at the ghost rows (the frame's pockets)
          grp0[0] = gcl_pq(blk, sub);
          grp0[0].row_idx = act_rows; grp0[0].m_count = act_count; grp0[0].M = N;
          grp0[1] = gcl_pq(blk, sub);
          grp0[1].A1 = e->h + (size_t)N * H; grp0[1].C = e->pqg + (size_t)N * 2 * H; grp0[1].M = n_frame_rows;
          grp0[1].row_idx = nullptr; grp0[1].m_count = nullptr;
        }
        if (!(rows0 && launch_node_group(s, grp0, 2) == hipSuccess)) {
          (void)hipGetLastError();
          HIP_TRY(launch_node_linear(s, gcl_pq(blk, sub)));
          if (rows0) HIP_TRY(launch_node_linear(s, grp0[1]));       // the ghost rows separately
        }
      }
      pqg_ready = false;
      const int g = blk * c.inv_sublayers + sub;
      const int radius = radius_of(g);               // this stage computes the nodes of level <= radius
      const bool ghost = g <= g_ghost_last;          // ... and the ghost rows of the canonical pocket
      const bool all_rows = !prune || radius >= LV;
      // list range of the stage: from the ghost segment or from its end, up to the end of level `radius`
      const int64_t begin = ghost ? 0 : ghost_slots;
      EdgeArgs ea{};
      ea.erow = L_row + begin; ea.ecol = L_col + begin; ea.ed0 = L_d0 + begin;
      ea.e_count = !prune ? e_all : (ghost ? e->lvl_end + radius : e->lvl_end + kLevels + radius);
      ea.e_cap = L_cap - (int)begin; ea.wt_base = (int)(begin / 32); ea.x = e->x;
      ea.n_lig = nlig; ea.n_nodes = N + n_frame_rows; ea.ldpq = 2 * H;
      ea.mlp[0] = EdgeMlpW{e->pqg, e->pqg + H, G(DSBDD_GCL_E1_WD), G(DSBDD_GCL_E1_WD0), G(DSBDD_GCL_E1_TAB),
                           G(DSBDD_GCL_E2_WT), G(DSBDD_GCL_E2_B), w2tp_of(blk, sub), w2tp16_of(blk, sub), w2e_of(blk, sub),
                           w2sk_of(blk, sub)};
      ea.mlp[1] = ea.mlp[0];
      // 16-edge-granule variant of this stage (engine option; never for block 0's two-list launch of a framed call)
      // (no emulated 16-edge kernel: with DSBDD_OPT_EMU the mask is ignored, a chain never mixes exact and emulated stages)
      const bool g16 = ((e->granule16 >> (g & 15)) & 1u) && !(split0 && blk == 0 && sub == 0) && !e->emu;
      // split-K variant of this stage (engine option; takes precedence over the 16-edge mask; block 0's two-list launch too)
      const bool gsk = ((e->splitk >> (g & 15)) & 1u) && can_sk;
      ea.att_w = G(DSBDD_GCL_ATT_W); ea.att_b = G(DSBDD_GCL_ATT_B); ea.attention = c.attention;
      ea.agg = e->agg; ea.agg_head = e->agg_head; ea.tile_ctr = e->tile_ctr;
      ea.norm_factor = c.normalization_factor;
      if (split0 && blk == 0 && sub == 0) {
        // (A) edges with a ligand endpoint, current coordinates -> agg / agg_head
        EdgeArgs a2 = ea;
        a2.erow = e->erow2; a2.ecol = e->ecol2; a2.ed0 = e->ed02; a2.e_count = e->row_ptr2 + N;
        a2.e_cap = (int)e->cap_edges; a2.wt_base = 0;          // (lists 2 and 3 count their wave tiles from 0)
        // (B) pocket-pocket edges of the frame (all samples, or the representative of identical pockets),
        //     raw pocket coordinates -> aggB / agg_headB.  (Running the small launch (B) on a second stream
        //     beside (A) was measured: 29.13 vs 29.42 ligands/s -- no gain, removed.)
        EdgeArgs a3 = ea;
        a3.erow = e->erow3; a3.ecol = e->ecol3; a3.ed0 = e->ed03; a3.e_count = e->row_ptr3 + e->frame_n3;
        a3.agg = e->aggB; a3.agg_head = e->agg_headB; a3.e_cap = (int)e->cap_edges; a3.wt_base = 0;   // (x: the ghost rows of e->x)
        // one launch: (B)'s few tiles ride behind (A)'s in the same persistent grid instead of paying a launch of
        // single-occupancy tile latency of their own (45 us for 35 tiles)
        a2.erow_b = a3.erow; a2.ecol_b = a3.ecol; a2.ed0_b = a3.ed0; a2.e_count_b = a3.e_count; a2.e_cap_b = a3.e_cap;
        a2.wt_base_b = a3.wt_base; a2.agg_b = a3.agg; a2.agg_head_b = a3.agg_head;
        HIP_TRY(launch_edge(e, s, MODE_GCL, a2, edge_bound + ((e->frame_cap3 + 127) / 128) * 128, false, gsk));
        hipLaunchKernelGGL(agg_complete2_kernel, dim3((N + n_ghost + 3) / 4), dim3(kThreads), 0, s, e->agg,
                           (const float*)e->agg_head, (const int*)e->row_ptr2, (const int*)e->deg2,
                           (const float*)e->aggB, (const float*)e->agg_headB, (const int*)e->row_ptr3,
                           (const int*)e->deg3, (const int*)e->twin, N, nlig, N, H, cone ? n_ghost : 0, (int)e->cap_tiles - 1);
        HIP_TRY(hipGetLastError());
      } else {
        // (timed: the launches over the whole list only, so that every timed launch is the same work)
        const bool timed = e->time_now && (all_rows || radius == e->plan_timed_level) && e->ev_used + 2 <= e->ev.size();
        if (timed) HIP_TRY(hipEventRecord(e->ev[e->ev_used], s));
        HIP_TRY(launch_edge(e, s, MODE_GCL, ea, L_bound, g16 && !gsk, gsk));
        if (timed) {
          HIP_TRY(hipEventRecord(e->ev[e->ev_used + 1], s));
          e->ev_used += 2;
        }
        // complete the rows whose edges span several wave tiles (ordered head partial sums, edge_mlp.h)
        const int n_rows = N + (ghost ? n_ghost : 0);
        hipLaunchKernelGGL(agg_complete_kernel, dim3((n_rows + 3) / 4), dim3(kThreads), 0, s, e->agg,
                           (const float*)e->agg_head, L_ptr, (const int*)e->deg, n_rows, H,
                           (int)((g16 && !gsk) ? e->cap_tiles16 : e->cap_tiles) - 1, (g16 && !gsk) ? 4 : 5);
        HIP_TRY(hipGetLastError());
      }
      // node MLP (egnn_new.py:21-24,56-57): h += W4 SiLU(W3 [h, agg] + b3) + b4
      NodeLinearArgs n1{e->h, H, H, e->agg, H, H, G(DSBDD_GCL_N1_WT), H, G(DSBDD_GCL_N1_B), nullptr, 0, e->t1, H,
                        (int)N, H, 1, nullptr, nullptr};
      NodeLinearArgs n2{e->t1, H, H, nullptr, 0, 0, G(DSBDD_GCL_N2_WT), H, G(DSBDD_GCL_N2_B), e->h, H, e->h, H,
                        (int)N, H, 0, nullptr, nullptr};
      rows_of(radius, ghost, n1); rows_of(radius, ghost, n2);
      chained_pq = false; chained_coord = false;
      bool fill_pq = false;
      if (use_chain) {
        // one launch: the node MLP and every projection of the new h whose rows are a contiguous part of the MLP's
        // row list (node_chain.h) -- the coordinate projections (after the block's last sublayer), the next message
        // stage's P|Q when it reads exactly the rows this stage computes
        NodeChainArgs ca{};
        ca.row_idx = n1.row_idx; ca.m_count = n1.m_count; ca.M = n1.M; ca.do_mlp = 1;
        ca.h = e->h; ca.agg = e->agg;
        ca.W1p = chain_w(blk, sub, 0); ca.b1 = G(DSBDD_GCL_N1_B);
        ca.W2p = chain_w(blk, sub, 1); ca.b2 = G(DSBDD_GCL_N2_B);
        const bool last_sub = sub + 1 == c.inv_sublayers;
        const int first = ghost ? n_ghost : 0;               // the ghost rows lead the list of a ghost stage
        if (last_sub) {
          const int QW = n_mlp * H;
          const float* wc = chain_w(blk, 0, 3);
          if (!subset) {
            ca.proj[ca.n_proj++] = ChainProj{wc, e->pq, PQ, PQ, nullptr, 0};
            chained_coord = true;
          } else if (prune && n1.row_idx) {                  // active / ligand rows = prefixes of the level list
            ca.proj[ca.n_proj++] = ChainProj{wc, e->pq, PQ, QW, act_count, first};
            ca.proj[ca.n_proj++] = ChainProj{wc + (size_t)(QW / 16) * (H / 16) * 256, e->pq + QW, PQ, QW,
                                             e->lvl_cnt + kLevels, first};
            chained_coord = true;
          }
        }
        const bool has_next = !last_sub || blk + 1 < c.n_layers;
        if (has_next) {
          const int nb = last_sub ? blk + 1 : blk, ns = last_sub ? 0 : sub + 1;
          const NodeLinearArgs nx = gcl_pq(nb, ns);
          // the next stage's P|Q rides along when it reads exactly the rows this stage computes -- or, in a ghost stage,
          // those plus rows that are about to take the canonical values: the ghost rows' P|Q is computed here and
          // copied together with their h (canon_fill_kernel)
          const bool same_rows = nx.row_idx == n1.row_idx && nx.m_count == n1.m_count && nx.M == n1.M;
          if (same_rows || ghost) {
            ca.proj[ca.n_proj++] = ChainProj{chain_w(nb, ns, 2), e->pqg, 2 * H, 2 * H, nullptr, 0};
            chained_pq = true;
            fill_pq = ghost && !same_rows;
          }
        }
        HIP_TRY(launch_node_chain(s, ca, H, e->n_cu));
      } else {
        HIP_TRY(launch_node_linear(s, n1));
        HIP_TRY(launch_node_linear(s, n2));
      }
      if (ghost) {
        // the rows the next stage reads but this one did not compute: canonical values (and their P|Q, see above)
        const int hi = radius_of(g + 1) + 1 < LV ? radius_of(g + 1) + 1 : LV;
        if (hi > radius) {
          hipLaunchKernelGGL(canon_fill_kernel, dim3((N - nlig + 3) / 4), dim3(kThreads), 0, s, e->h,
                             (const int*)e->lvl, (const int*)e->twin, nlig, N, N, radius, hi, H,
                             fill_pq ? e->pqg : (float*)nullptr, 2 * H);
          fill_pq = false;
          HIP_TRY(hipGetLastError());
        }
      }
    }
    {
      auto Q = [&](int which) { return W[eq_slot(c, blk, which)]; };
      // first-layer projections, column order [Q_coord | Q_cross | P_coord | P_cross]; the next
      // block's GCL P|Q projection reads the same h and shares the launch when it can
      const int QW = n_mlp * H;   // width of the Q (column-node) part
      NodeLinearArgs grp[kMaxGroup];
      int ng = 0;
      if (!chained_coord) {
        if (subset) {
          grp[ng++] = NodeLinearArgs{e->h, H, H, nullptr, 0, 0, Q(DSBDD_EQ_C1_WT), PQ, nullptr, nullptr, 0, e->pq, PQ,
                                     (int)N, QW, 0, act_rows, act_count};
          grp[ng++] = NodeLinearArgs{e->h, H, H, nullptr, 0, 0, Q(DSBDD_EQ_C1_WT) + QW, PQ, nullptr, nullptr, 0,
                                     e->pq + QW, PQ, (int)n_lig, QW, 0, nullptr, nullptr};
        } else {
          grp[ng++] = NodeLinearArgs{e->h, H, H, nullptr, 0, 0, Q(DSBDD_EQ_C1_WT), PQ, nullptr, nullptr, 0, e->pq, PQ,
                                     (int)N, PQ, 0, nullptr, nullptr};
        }
      }
      const int n_coord = ng;
      const bool want_next = blk + 1 < c.n_layers && !chained_pq;
      if (chained_pq && blk + 1 < c.n_layers) pqg_ready = true;
      if (want_next && use_chain) {
        // the next stage reads more rows than this one computed (ascending radii of the forward cone: the rest were
        // filled with canonical values above): its P|Q as a launch of its own, rows streamed from global memory
        const NodeLinearArgs nx = gcl_pq(blk + 1, 0);
        NodeChainArgs ca{};
        ca.row_idx = nx.row_idx; ca.m_count = nx.m_count; ca.M = nx.M; ca.do_mlp = 0; ca.h = e->h;
        ca.n_proj = 1;
        ca.proj[0] = ChainProj{chain_w(blk + 1, 0, 2), e->pqg, 2 * H, 2 * H, nullptr, 0};
        HIP_TRY(launch_node_chain(s, ca, H, e->n_cu));
        pqg_ready = true;
      } else if (want_next) {
        grp[ng++] = gcl_pq(blk + 1, 0);
      }
      if (ng > 0) {
        if (e->node_group && launch_node_group(s, grp, ng) == hipSuccess) {
          if (ng > n_coord) pqg_ready = true;
        } else {
          (void)hipGetLastError();
          for (int i = 0; i < n_coord; ++i) HIP_TRY(launch_node_linear(s, grp[i]));   // the coordinate projections only
        }
      }
      chained_pq = false;
      EdgeArgs ea{};
      ea.erow = L_row + ghost_slots; ea.ecol = L_col + ghost_slots; ea.ed0 = L_d0 + ghost_slots; ea.e_count = e_upd;
      ea.e_cap = L_cap - (int)ghost_slots; ea.wt_base = (int)(ghost_slots / 32); ea.x = e->x;
      ea.n_lig = nlig; ea.n_nodes = N + n_frame_rows; ea.ldpq = PQ;
      ea.mlp[0] = EdgeMlpW{e->pq + QW, e->pq, Q(DSBDD_EQ_C_WD), Q(DSBDD_EQ_C_WD0), Q(DSBDD_EQ_C_TAB),
                           Q(DSBDD_EQ_C_W2T), Q(DSBDD_EQ_C_B2), w2tp_of(blk, c.inv_sublayers), w2tp16_of(blk, c.inv_sublayers),
                           w2e_of(blk, c.inv_sublayers), w2sk_of(blk, c.inv_sublayers)};
      if (n_mlp == 2)
        ea.mlp[1] = EdgeMlpW{e->pq + QW + H, e->pq + H, Q(DSBDD_EQ_X_WD), Q(DSBDD_EQ_X_WD0), Q(DSBDD_EQ_X_TAB),
                             Q(DSBDD_EQ_X_W2T), Q(DSBDD_EQ_X_B2), w2tp_of(blk, c.inv_sublayers + 1),
                             w2tp16_of(blk, c.inv_sublayers + 1), w2e_of(blk, c.inv_sublayers + 1), w2sk_of(blk, c.inv_sublayers + 1)};
      else
        ea.mlp[1] = ea.mlp[0];
      ea.w3 = Q(DSBDD_EQ_W3); ea.node_batch = e->node_batch; ea.mean = e->mean;
      ea.norm_constant = c.norm_constant; ea.coords_range = c.coords_range; ea.use_tanh = c.use_tanh;
      ea.n_mlp = n_mlp; ea.xagg = e->xagg; ea.xagg_head = e->xagg_head;
      const bool csk = ((e->splitk >> (16 + (blk & 15))) & 1u) && can_sk;        // split-K variant of this stage (one sum per MLP)
      const bool c16 = ((e->granule16 >> (16 + (blk & 15))) & 1u) && !e->emu && !csk;   // 16-edge-granule variant of this stage
      ea.xagg_stride = (size_t)N * 3; ea.xhead_stride = (size_t)(c16 ? e->cap_tiles16 : e->cap_tiles) * 4;
      ea.tile_ctr = e->tile_ctr; ea.norm_factor = c.normalization_factor;
      ea.pass_split = (c16 || csk) ? 1 : e->coord_split;
      if (e->ts_buf && e->ts_next < e->ts_cap) ea.ts = e->ts_buf + (size_t)(e->ts_next++) * 1024;
     
      HIP_TRY(launch_edge(e, s, MODE_COORD, ea, L_bound, c16, csk));
      {
        const int n_q = ((e->coord_split || c16 || csk) && n_mlp == 2) ? 2 : 1;   // (the 16-edge / split-K kernels keep one sum per MLP)
        const int c_shift = c16 ? 4 : 5, c_max = (int)(c16 ? e->cap_tiles16 : e->cap_tiles) - 1;
        // few updated rows (the ligand's): one workgroup per sample updates them and reduces the next block's mean;
        // all rows updated (joint model): the wide per-component kernel, the mean stays a launch of its own
        const bool next_mean = subset && n_mlp == 2 && blk + 1 < c.n_layers;
        if (!subset) {
          if (n_upd > 0) {
            hipLaunchKernelGGL(coord_update_kernel, dim3((3 * n_upd + 255) / 256), dim3(256), 0, s, e->x,
                               (const float*)e->xagg, (const float*)e->xagg_head, n_q, ea.xagg_stride, ea.xhead_stride,
                               L_ptr, (const int*)e->deg, 3 * n_upd, c_max, c_shift);
            HIP_TRY(hipGetLastError());
          }
        } else if (n_upd > 0 || next_mean) {
          hipLaunchKernelGGL(coord_update_mean_kernel, dim3(B), dim3(kThreads), 0, s, e->x, (const float*)e->xagg,
                             (const float*)e->xagg_head, n_q, ea.xagg_stride, ea.xhead_stride, L_ptr,
                             (const int*)e->deg, n_upd, (const int*)e->lig_off, (const int*)e->poc_off, nlig,
                             next_mean ? e->mean : (float*)nullptr, c_max, c_shift);
          HIP_TRY(hipGetLastError());
        }
      }
    }
    if (e->trace_h)
      HIP_TRY(hipMemcpyAsync(e->trace_h + (size_t)blk * N * H, e->h, (size_t)N * H * 4, hipMemcpyDeviceToDevice, s));
    if (e->trace_x)
      HIP_TRY(hipMemcpyAsync(e->trace_x + (size_t)blk * N * 3, e->x, (size_t)N * 12, hipMemcpyDeviceToDevice, s));
  }
  // ---- embedding_out, decoders (egnn_new.py:241, dynamics.py:147-153) --------
  // ligand output only, pocket-conditioning mode: embedding_out + atom decoder + velocity + NaN flag in ONE launch
  // (csrc/lig_head.h; DSBDD_LIG_HEAD=0: the three launches below)
  LigHeadArgs lh{e->h, H, W[DSBDD_G_EMBOUT_WT], JP, W[DSBDD_G_EMBOUT_B], J,
                 W[DSBDD_G_ATOM_DEC_W0T], pad4(2 * a), W[DSBDD_G_ATOM_DEC_B0], 2 * a,
                 W[DSBDD_G_ATOM_DEC_W1T], pad4(a), W[DSBDD_G_ATOM_DEC_B1], a,
                 e->x, e->x_in, nlig, N, eps_lig, dl, status};
  if (e->lig_head && !eps_pocket && !c.update_pocket_coords && lig_head_fits(lh)) {
    HIP_TRY(launch_lig_head(s, lh));
    return DSBDD_OK;
  }
  HIP_TRY(nl(s, e->h, H, H, nullptr, 0, 0, W[DSBDD_G_EMBOUT_WT], JP, W[DSBDD_G_EMBOUT_B], nullptr, 0, e->hout, JP,
             eps_pocket ? N : n_lig, JP, 0));
  {
    Mlp2Problem dec[2] = {
        {e->hout, JP, J, W[DSBDD_G_ATOM_DEC_W0T], pad4(2 * a), W[DSBDD_G_ATOM_DEC_B0], 2 * a,
         W[DSBDD_G_ATOM_DEC_W1T], pad4(a), W[DSBDD_G_ATOM_DEC_B1], a, eps_lig + 3, dl, (int)n_lig},
        {e->hout + (size_t)n_lig * JP, JP, J, W[DSBDD_G_RES_DEC_W0T], pad4(2 * r), W[DSBDD_G_RES_DEC_B0], 2 * r,
         W[DSBDD_G_RES_DEC_W1T], pad4(r), W[DSBDD_G_RES_DEC_B1], r, eps_pocket ? eps_pocket + 3 : nullptr, dp,
         (int)n_pocket}};
    if (mlp2_fits(dec[0]) && mlp2_fits(dec[1])) {
      HIP_TRY(launch_mlp2(s, dec, eps_pocket ? 2 : 1));
    } else {
      HIP_TRY(nl(s, e->hout, JP, J, nullptr, 0, 0, W[DSBDD_G_ATOM_DEC_W0T], pad4(2 * a), W[DSBDD_G_ATOM_DEC_B0], nullptr, 0,
                 e->enc_tmp, LE, n_lig, 2 * a, 1));
      HIP_TRY(nl(s, e->enc_tmp, LE, 2 * a, nullptr, 0, 0, W[DSBDD_G_ATOM_DEC_W1T], pad4(a), W[DSBDD_G_ATOM_DEC_B1], nullptr, 0,
                 eps_lig + 3, dl, n_lig, a, 0));
      if (eps_pocket) {
        float* tmp_p = e->enc_tmp + (size_t)n_lig * LE;
        HIP_TRY(nl(s, e->hout + (size_t)n_lig * JP, JP, J, nullptr, 0, 0, W[DSBDD_G_RES_DEC_W0T], pad4(2 * r),
                   W[DSBDD_G_RES_DEC_B0], nullptr, 0, tmp_p, LE, n_pocket, 2 * r, 1));
        HIP_TRY(nl(s, tmp_p, LE, 2 * r, nullptr, 0, 0, W[DSBDD_G_RES_DEC_W1T], pad4(r), W[DSBDD_G_RES_DEC_B1], nullptr, 0,
                   eps_pocket + 3, dp, n_pocket, r, 0));
      }
    }
  }
  // ---- velocity, NaN guard, joint-mode COM removal (dynamics.py:136,155-164) --
  hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(kThreads), 0, s, (const float*)e->x, (const float*)e->x_in,
                     (const int*)e->lig_off, (const int*)e->poc_off, nlig, c.update_pocket_coords, eps_lig, dl,
                     eps_pocket, dp, status);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

extern "C" {

int dsbdd_dynamics_forward(dsbdd_engine* e, void* stream, const float* xh_lig, const float* xh_pocket,
                           const float* t, int64_t t_count, const int64_t* mask_lig,
                           const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                           const int32_t* ext_row, const int32_t* ext_col, int64_t ext_n_edges,
                           float* eps_lig, float* eps_pocket, int32_t* status) {
  StreamDevice stream_device_(stream);
  if (!e || !xh_lig || !xh_pocket || !t || !mask_lig || !mask_pocket || !eps_lig || !status)
    return fail(DSBDD_ERR_ARG, "null argument");
  if (!e->has_weights) return fail(DSBDD_ERR_STATE, "weights not set");
  if (!e->ws) return fail(DSBDD_ERR_STATE, "workspace not bound");
  if (n_lig > e->cap_lig || n_pocket > e->cap_poc || batch > e->cap_batch || n_lig < 0 || n_pocket < 0 ||
      batch < 1)
    return fail(DSBDD_ERR_CAPACITY, "sizes exceed the bound workspace");
  if (t_count != 1 && t_count != batch) return fail(DSBDD_ERR_ARG, "t must have 1 or batch entries");
  if (ext_row && (!ext_col || ext_n_edges < 0 || ext_n_edges > e->cap_edges))
    return fail(DSBDD_ERR_CAPACITY, "external edge list exceeds edge capacity");
  hipStream_t s = static_cast<hipStream_t>(stream);

  // Ghost rows of a pocket frame (set once per chain) share the node arrays with the real nodes: a call the frame
  // does not apply to (other sizes, teacher-forced edges) may write over them -- eagerly or through a replayed
  // graph --, so they are re-written from the pristine frame data before the next framed call, whichever way it runs.
  {
    const bool framed = e->frame && !e->cfg.update_pocket_coords && !ext_row && n_lig == e->frame_nlig &&
                        n_pocket == e->frame_npoc && batch == e->frame_batch;
    if (!framed) {
      e->ghost_dirty = true;
      e->h0_pocket_valid = false;
    } else {
      if (e->ghost_dirty) {
        int rc = ghost_setup(e, s);
        if (rc) return rc;
      }
      if (!e->h0_pocket_valid) {
        // the residue encoder on the chain's pocket features, once per chain (and again after a call the frame does not
        // apply to overwrote the rows): dynamics.py:97
        const dsbdd_config& c = e->cfg;
        const int r = c.residue_nf, J = c.joint_nf, JP = pad4(J + 1);
        const float* const* W = e->slots.data();
        Mlp2Problem enc{xh_pocket + 3, 3 + r, r, W[DSBDD_G_RES_ENC_W0T], pad4(2 * r), W[DSBDD_G_RES_ENC_B0], 2 * r,
                        W[DSBDD_G_RES_ENC_W1T], pad4(J), W[DSBDD_G_RES_ENC_B1], J, e->h0 + (size_t)n_lig * JP, JP,
                        (int)n_pocket};
        if (mlp2_fits(enc)) {
          HIP_TRY(launch_mlp2(s, &enc, 1));
          e->h0_pocket_valid = true;
        }
      }
    }
  }

  // eager path: graphs off, timing / tracing hooks active (they enqueue event records and
  // copies that must not be frozen into a graph), or teacher-forced edges (test-only)
  // kernel timing: the calls whose launches are bracketed by event records run eagerly, the others may replay
  e->time_now = e->profile > 0 && (e->prof_call++ % e->profile) == 0;
  const bool eager = !e->use_graph || e->time_now || e->trace_h || e->trace_x || ext_row;
  if (eager) {
    ++e->n_eager;
    return forward_impl(e, s, xh_lig, xh_pocket, t, t_count, mask_lig, mask_pocket, n_lig, n_pocket, batch,
                        ext_row, ext_col, ext_n_edges, eps_lig, eps_pocket, status);
  }

  std::vector<uint64_t> key = {(uint64_t)(uintptr_t)xh_lig, (uint64_t)(uintptr_t)xh_pocket, (uint64_t)(uintptr_t)t,
                               (uint64_t)t_count, (uint64_t)(uintptr_t)mask_lig, (uint64_t)(uintptr_t)mask_pocket,
                               (uint64_t)n_lig, (uint64_t)n_pocket, (uint64_t)batch, (uint64_t)(uintptr_t)eps_lig,
                               (uint64_t)(uintptr_t)eps_pocket, (uint64_t)(uintptr_t)status,
                               (uint64_t)(uintptr_t)e->ws, (uint64_t)(uintptr_t)e->slots.data()[0],
                               (uint64_t)(uintptr_t)s, (uint64_t)e->frame};
  dsbdd_engine::GraphEntry* g = nullptr;
  for (auto& ge : e->graphs)
    if (ge.key == key) { g = &ge; break; }
  if (!g) {
    if (e->graphs.size() >= 8) {       // bounded cache: drop the oldest entry
      if (e->graphs.front().exec) (void)hipGraphExecDestroy(e->graphs.front().exec);
      if (e->graphs.front().graph) (void)hipGraphDestroy(e->graphs.front().graph);
      e->graphs.erase(e->graphs.begin());
    }
    e->graphs.emplace_back();
    g = &e->graphs.back();
    g->key = key;
  }
  if (g->exec) {
    ++e->n_replay;
    HIP_TRY(hipGraphLaunch(g->exec, s));
    e->plan_radius = g->plan_radius; e->plan_ghost = g->plan_ghost; e->plan_timed_level = g->plan_timed_level;
    return DSBDD_OK;
  }
  ++e->n_eager;
  if (g->seen++ == 0)                  // first sight of this signature: plain launches
    return forward_impl(e, s, xh_lig, xh_pocket, t, t_count, mask_lig, mask_pocket, n_lig, n_pocket, batch,
                        nullptr, nullptr, 0, eps_lig, eps_pocket, status);
  // second call with the same arguments: capture the sequence, then replay it
  if (!e->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
  // (pack kernels enqueued during a capture have not run if the capture fails: remember what was current before)
  const bool rdy[4] = {e->w2tp_ready, e->w2tp16_ready, e->w2e_ready, e->wchain_ready};
  auto restore_ready = [&]() { e->w2tp_ready = rdy[0]; e->w2tp16_ready = rdy[1]; e->w2e_ready = rdy[2]; e->wchain_ready = rdy[3]; };
  HIP_TRY(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
  const int rc = forward_impl(e, e->cap_stream, xh_lig, xh_pocket, t, t_count, mask_lig, mask_pocket, n_lig,
                              n_pocket, batch, nullptr, nullptr, 0, eps_lig, eps_pocket, status);
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(e->cap_stream, &graph);
  if (rc != DSBDD_OK || ec != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    e->use_graph = 0;                  // do not try again; fall back to plain launches
    restore_ready();
    if (rc != DSBDD_OK) return rc;
    return forward_impl(e, s, xh_lig, xh_pocket, t, t_count, mask_lig, mask_pocket, n_lig, n_pocket, batch,
                        nullptr, nullptr, 0, eps_lig, eps_pocket, status);
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
    (void)hipGraphDestroy(graph);
    e->use_graph = 0;
    restore_ready();
    return forward_impl(e, s, xh_lig, xh_pocket, t, t_count, mask_lig, mask_pocket, n_lig, n_pocket, batch,
                        nullptr, nullptr, 0, eps_lig, eps_pocket, status);
  }
  g->graph = graph;
  g->exec = exec;
  g->plan_radius = e->plan_radius; g->plan_ghost = e->plan_ghost; g->plan_timed_level = e->plan_timed_level;
  ++e->n_capture;
  HIP_TRY(hipGraphLaunch(g->exec, s));
  return DSBDD_OK;
}

int dsbdd_cond_reverse_update(void* stream, float* z_lig, float* xh_pocket, const float* eps_lig,
                              const float* noise, const int64_t* mask_lig, const int64_t* mask_pocket,
                              int64_t n_lig, int64_t n_pocket, int64_t batch, int32_t atom_nf,
                              int32_t residue_nf, float alpha_ts, float c_eps, float sigma, int32_t remove_com) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !xh_pocket || !eps_lig || !noise || !mask_lig || !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(cond_update_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     z_lig, xh_pocket, eps_lig, noise, mask_lig, (int)n_lig, mask_pocket, (int)n_pocket,
                     3 + atom_nf, 3 + residue_nf, alpha_ts, c_eps, sigma, remove_com);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_joint_reverse_update(void* stream, float* z_lig, float* z_pocket, const float* eps_lig,
                               const float* eps_pocket, const float* noise_lig, const float* noise_pocket,
                               const int64_t* mask_lig, const int64_t* mask_pocket, int64_t n_lig,
                               int64_t n_pocket, int64_t batch, int32_t atom_nf, int32_t residue_nf,
                               float alpha_ts, float c_eps, float sigma, int32_t center_noise) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !z_pocket || !eps_lig || !eps_pocket || !noise_lig || !noise_pocket || !mask_lig ||
      !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(joint_update_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     z_lig, z_pocket, eps_lig, eps_pocket, noise_lig, noise_pocket, mask_lig, (int)n_lig,
                     mask_pocket, (int)n_pocket, 3 + atom_nf, 3 + residue_nf, alpha_ts, c_eps, sigma, center_noise);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_segment_mean3(void* stream, const float* x, int32_t ld, const int64_t* mask, int64_t n_rows,
                        int64_t batch, float* out) {
  StreamDevice stream_device_(stream);
  if (!x || !mask || !out || batch < 1 || ld < 3 || n_rows < 0) return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(segment_mean3_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     x, ld, mask, (int)n_rows, out);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_cond_affine_noise(void* stream, float* z_lig, float* xh_pocket, const float* noise,
                            const int64_t* mask_lig, const int64_t* mask_pocket, int64_t n_lig,
                            int64_t n_pocket, int64_t batch, int32_t atom_nf, int32_t residue_nf, float a,
                            float sigma, int32_t remove_com) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !xh_pocket || !noise || !mask_lig || !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(cond_affine_noise_kernel, dim3((int)batch), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), z_lig, xh_pocket, noise, mask_lig, (int)n_lig, mask_pocket,
                     (int)n_pocket, 3 + atom_nf, 3 + residue_nf, a, sigma, remove_com);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_joint_affine_noise(void* stream, float* z_lig, float* z_pocket, const float* noise_lig,
                             const float* noise_pocket, const int64_t* mask_lig, const int64_t* mask_pocket,
                             int64_t n_lig, int64_t n_pocket, int64_t batch, int32_t atom_nf,
                             int32_t residue_nf, float a, float sigma, int32_t center_noise,
                             int32_t remove_com) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !z_pocket || !noise_lig || !noise_pocket || !mask_lig || !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(joint_affine_noise_kernel, dim3((int)batch), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), z_lig, z_pocket, noise_lig, noise_pocket, mask_lig,
                     (int)n_lig, mask_pocket, (int)n_pocket, 3 + atom_nf, 3 + residue_nf, a, sigma, center_noise,
                     remove_com);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_cond_repaint_update(void* stream, float* z_lig, float* xh_pocket, float* scratch_lig,
                              const float* xh0_lig, const float* com_pocket0, const float* fixed,
                              const float* noise_known, const float* noise_resample, const int64_t* mask_lig,
                              const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                              int32_t atom_nf, int32_t residue_nf, float alpha_s, float sigma_s,
                              float alpha_ts, float sigma_ts, int32_t resample, int32_t remove_com) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !xh_pocket || !scratch_lig || !xh0_lig || !com_pocket0 || !fixed || !noise_known ||
      (resample && !noise_resample) || !mask_lig || !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  CondRepaintArgs a{z_lig, xh_pocket, scratch_lig, xh0_lig, com_pocket0, fixed, noise_known, noise_resample,
                    mask_lig, mask_pocket, (int)n_lig, (int)n_pocket, 3 + atom_nf, 3 + residue_nf, alpha_s,
                    sigma_s, alpha_ts, sigma_ts, resample, remove_com};
  hipLaunchKernelGGL(cond_repaint_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_cond_step_keyed(void* stream, float* z_lig, float* xh_pocket, const float* eps_lig, float* scratch_lig,
                          const float* xh0_lig, const float* com_pocket0, const float* fixed, const int64_t* mask_lig,
                          const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch, int32_t atom_nf,
                          int32_t residue_nf, float alpha_ts, float c_eps, float sigma, int32_t repaint, float alpha_s,
                          float sigma_s, float sigma_ts, int32_t remove_com, uint64_t seed, uint64_t draw_index,
                          int64_t sample_offset, const int64_t* sample_ids, float* t_word, float t_next) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !xh_pocket || !eps_lig || !mask_lig || !mask_pocket || batch < 1 || repaint < 0 || repaint > 2 ||
      (repaint && (!scratch_lig || !xh0_lig || !com_pocket0 || !fixed)))
    return fail(DSBDD_ERR_ARG, "bad argument");
  CondStepArgs a{};
  a.rp = CondRepaintArgs{z_lig, xh_pocket, scratch_lig, xh0_lig, com_pocket0, fixed, nullptr, nullptr, mask_lig, mask_pocket,
                         (int)n_lig, (int)n_pocket, 3 + atom_nf, 3 + residue_nf, alpha_s, sigma_s, alpha_ts, sigma_ts,
                         repaint == 2, remove_com};
  a.eps = eps_lig; a.u_alpha_ts = alpha_ts; a.u_c_eps = c_eps; a.u_sigma = sigma; a.repaint = repaint;
  a.seed = seed; a.draw = draw_index; a.sample_offset = sample_offset; a.sample_ids = sample_ids;
  a.t_word = t_word; a.t_next = t_next;
  hipLaunchKernelGGL(cond_step_keyed_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_joint_repaint_update(void* stream, float* z_lig, float* z_pocket, float* scratch_lig,
                               float* scratch_pocket, const float* xh0_lig, const float* xh0_pocket,
                               const float* fixed_lig, const float* fixed_pocket, const float* noise_known_lig,
                               const float* noise_known_pocket, const float* noise_jump_lig,
                               const float* noise_jump_pocket, const int64_t* mask_lig,
                               const int64_t* mask_pocket, int64_t n_lig, int64_t n_pocket, int64_t batch,
                               int32_t atom_nf, int32_t residue_nf, float alpha_s, float sigma_s,
                               float alpha_ts, float sigma_ts, int32_t jump) {
  StreamDevice stream_device_(stream);
  if (!z_lig || !z_pocket || !scratch_lig || !scratch_pocket || !xh0_lig || !xh0_pocket || !fixed_lig ||
      !fixed_pocket || !noise_known_lig || !noise_known_pocket || (jump && (!noise_jump_lig || !noise_jump_pocket)) ||
      !mask_lig || !mask_pocket || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  JointRepaintArgs a{z_lig, z_pocket, scratch_lig, scratch_pocket, xh0_lig, xh0_pocket, fixed_lig, fixed_pocket,
                     noise_known_lig, noise_known_pocket, noise_jump_lig, noise_jump_pocket, mask_lig, mask_pocket,
                     (int)n_lig, (int)n_pocket, 3 + atom_nf, 3 + residue_nf, alpha_s, sigma_s, alpha_ts, sigma_ts,
                     jump};
  hipLaunchKernelGGL(joint_repaint_kernel, dim3((int)batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_randn_keyed(void* stream, float* out, const int64_t* mask, int64_t n_rows, int32_t n_cols,
                      int64_t batch, int64_t sample_offset, const int64_t* sample_ids, uint64_t seed,
                      uint64_t draw_index, uint32_t stream_id) {
  StreamDevice stream_device_(stream);
  (void)batch;
  if (!out || !mask || n_rows < 0 || n_cols < 1) return fail(DSBDD_ERR_ARG, "bad argument");
  const int64_t n = n_rows * n_cols;
  if (n == 0) return DSBDD_OK;
  hipLaunchKernelGGL(randn_keyed_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), out, mask, (int)n_rows, (int)n_cols, sample_offset,
                     sample_ids, seed, draw_index, stream_id);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_node_linear(void* stream, const float* A1, int32_t lda1, int32_t K1, const float* A2, int32_t lda2,
                      int32_t K2, const float* WT, int32_t ldw, const float* bias, const float* R, int32_t ldr,
                      float* C, int32_t ldc, int64_t M, int32_t N, int32_t act) {
  StreamDevice stream_device_(stream);
  if (!A1 || !WT || !C || K1 < 1 || K2 < 0 || (K2 > 0 && !A2) || (ldw & 3) || N > ldw ||
      (reinterpret_cast<uintptr_t>(WT) & 15))
    return fail(DSBDD_ERR_ARG, "bad argument (WT must be 16-byte aligned with ldw % 4 == 0)");
  HIP_TRY(nl(static_cast<hipStream_t>(stream), A1, lda1, K1, A2, lda2, K2, WT, ldw, bias, R, ldr, C, ldc, M, N, act));
  return DSBDD_OK;
}

int dsbdd_bond_orders(void* stream, const float* x, const int32_t* atom_type, const int32_t* mol_off,
                      int64_t batch, int32_t n_types, const float* bonds1, const float* bonds2,
                      const float* bonds3, float margin1, float margin2, float margin3, int32_t n_max,
                      int8_t* order) {
  StreamDevice stream_device_(stream);
  if (!x || !atom_type || !mol_off || !bonds1 || !bonds2 || !bonds3 || !order || batch < 1 || n_types < 1 ||
      n_max < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(order, 0, (size_t)batch * n_max * n_max, s));
  BondArgs a{x, atom_type, mol_off, bonds1, bonds2, bonds3, margin1, margin2, margin3, n_types, n_max,
             reinterpret_cast<signed char*>(order)};
  hipLaunchKernelGGL(bond_orders_kernel, dim3((unsigned)batch), dim3(64), 0, s, a);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_build_edges(void* stream, const float* x, const int64_t* mask_lig, const int64_t* mask_pocket,
                      int64_t n_lig, int64_t n_pocket, int64_t batch, const dsbdd_config* cfg,
                      int32_t* node_batch, int32_t* lig_off, int32_t* poc_off, int32_t* deg, int32_t* row_ptr,
                      int32_t* edge_row, int32_t* edge_col, float* edge_d0, int64_t edge_capacity,
                      int32_t* status) {
  StreamDevice stream_device_(stream);
  if (!x || !mask_lig || !mask_pocket || !cfg || !node_batch || !lig_off || !poc_off || !deg || !row_ptr ||
      !edge_row || !edge_col || !edge_d0 || !status || batch < 1)
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = (int)(n_lig + n_pocket), B = (int)batch;
  const int work = N > B + 1 ? N : B + 1;
  hipLaunchKernelGGL(prep_kernel, dim3((work + 255) / 256), dim3(256), 0, s, mask_lig, (int)n_lig, mask_pocket,
                     (int)n_pocket, B, node_batch, lig_off, poc_off, (int*)nullptr);
  HIP_TRY(hipGetLastError());
  return build_edges_impl(s, x, (int)n_lig, N, B, *cfg, node_batch, lig_off, poc_off, deg, row_ptr, edge_row,
                          edge_col, edge_d0, edge_capacity, status);
}

}  // extern "C"

// ---- training-step building blocks (csrc/train.h) ---------------------------------------------------------------
static int device_cus() {      // CU count of the CURRENT device (cached per device id)
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] == 0) {
    hipDeviceProp_t pr;
    int n = 0;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    cache[dev] = n > 0 ? n : 256;
  }
  return cache[dev];
}

static hipError_t launch_edge_plain(int H, hipStream_t s, int mode, const EdgeArgs& a, int64_t edge_bound) {
  int64_t tiles = (edge_bound + 127) / 128;
  const bool split = mode == MODE_COORD && a.pass_split && a.n_mlp == 2;      // one workgroup per (tile, MLP)
  int64_t g = split ? 2 * tiles : tiles;
  if (g > 2LL * device_cus()) g = 2LL * device_cus();
  const int q8 = split ? 16 : 8;
  int grid = (int)((g + q8 - 1) / q8 * q8);
  if (grid < q8) grid = q8;
  if (a.z2_out) {          // training forward of the network path: the stage keeps z2 (edge_wave_kernel<.., STORE>)
    if (a.e_count_b || a.wt_base) return hipErrorInvalidValue;
#define DSBDD_STORE_CASE(HH) \
    case HH: if (mode == MODE_GCL) hipLaunchKernelGGL((edge_wave_kernel<HH, MODE_GCL, false, 0, true>), dim3(grid), dim3(kThreads), 0, s, a); \
             else hipLaunchKernelGGL((edge_wave_kernel<HH, MODE_COORD, false, 0, true>), dim3(grid), dim3(kThreads), 0, s, a); \
             break;
    switch (H) {
      DSBDD_STORE_CASE(64) DSBDD_STORE_CASE(128) DSBDD_STORE_CASE(192) DSBDD_STORE_CASE(256)
      default: return hipErrorInvalidValue;
    }
#undef DSBDD_STORE_CASE
    return hipGetLastError();
  }
  switch (H) {
    case 64: return launch_wave_t<64>(s, mode, a, grid);
    case 128: return launch_wave_t<128>(s, mode, a, grid);
    case 192: return launch_wave_t<192>(s, mode, a, grid);
    case 256: return launch_wave_t<256>(s, mode, a, grid);
  }
  return hipErrorInvalidValue;
}

// out[i] = sum_p part[p * stride + i] in a fixed order (two levels above 64 parts); tmp: ceil(n_part / 32) * width floats
static hipError_t reduce_parts(hipStream_t s, const float* part, int n_part, size_t stride, int width, float* out,
                               float* tmp) {
  const int bx = (width + 255) / 256;
  if (n_part <= 96) {
    hipLaunchKernelGGL(partial_reduce_kernel, dim3(bx, 1), dim3(256), 0, s, part, n_part, stride, width,
                       n_part > 0 ? n_part : 1, out, (size_t)0);
    return hipGetLastError();
  }
  const int groups = (n_part + 31) / 32;
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(bx, groups), dim3(256), 0, s, part, n_part, stride, width, 32, tmp,
                     (size_t)width);
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(bx, 1), dim3(256), 0, s, (const float*)tmp, groups, (size_t)width,
                     width, groups, out, (size_t)0);
  return hipGetLastError();
}

struct WgradPlan { int chunks, kc; size_t floats; };
static WgradPlan wgrad_plan(int64_t K, int64_t M, int64_t N) {
  // short chunks for the node-level gradients (K = a few thousand rows: the launch is a latency chain), at most
  // 256 / tiles chunks for the edge-level ones: <= 96 partial slabs reduce in ONE ordered launch (round 6: 768 -> 256,
  // 14.46 -> 14.03 ms per training step, profiles/r6_train_step.md; round 4's sweep had preferred 768 when the reduction
  // was two launches either way)
  static const int min_kc = [] { const char* v = getenv("DSBDD_WGRAD_MINKC"); return v && atoi(v) >= 32 ? atoi(v) : 64; }();
  static const int max_wg = [] { const char* v = getenv("DSBDD_WGRAD_MAXWG"); return v && atoi(v) >= 1 ? atoi(v) : 256; }();
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int64_t chunks = (K + min_kc - 1) / min_kc;
  const int64_t cap = max_wg / tiles > 1 ? max_wg / tiles : 1;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  int64_t kc = ((K + chunks - 1) / chunks + 31) / 32 * 32;
  if (kc < 32) kc = 32;
  chunks = (K + kc - 1) / kc;
  if (chunks < 1) chunks = 1;
  WgradPlan p{(int)chunks, (int)kc, 0};
  p.floats = (size_t)chunks * M * N + (size_t)((chunks + 31) / 32) * M * N;
  return p;
}
// Scratch floats that cover wgrad_plan(K, M, N) for EVERY K <= K_max.  The plan is not monotonic in K (kc is rounded up
// to a multiple of 32 after the chunk cap, so a smaller K can end with more chunks: K = 160 000 -> 186, K = 30 000 -> 188
// at 256 x 256), but its chunk count never exceeds min(ceil(K / min_kc), cap), which is.  (the coordinate
// stage's backward calls wgrad with K = e_upd < E on a scratch sized for E.)
static size_t wgrad_floats_upto(int64_t K_max, int64_t M, int64_t N) {
  static const int min_kc = [] { const char* v = getenv("DSBDD_WGRAD_MINKC"); return v && atoi(v) >= 32 ? atoi(v) : 64; }();
  static const int max_wg = [] { const char* v = getenv("DSBDD_WGRAD_MAXWG"); return v && atoi(v) >= 1 ? atoi(v) : 256; }();
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int64_t cap = max_wg / tiles > 1 ? max_wg / tiles : 1;
  int64_t chunks = (K_max + min_kc - 1) / min_kc;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  return (size_t)chunks * M * N + (size_t)((chunks + 31) / 32) * M * N;
}

static int wgrad_impl(hipStream_t s, const float* A, int lda, const float* B, int ldb, int64_t K, int M, int N, float* C,
                      float* scratch, size_t scratch_floats) {
  const WgradPlan pl = wgrad_plan(K, M, N);
  if (pl.floats > scratch_floats) return fail(DSBDD_ERR_CAPACITY, "weight-gradient scratch too small for this plan");
  WgradArgs a{A, lda, B, ldb, (int)K, M, N, scratch, pl.kc};
  hipLaunchKernelGGL(wgrad_kernel, dim3((M + 127) / 128, (N + 127) / 128, pl.chunks), dim3(kThreads), 0, s, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(reduce_parts(s, scratch, pl.chunks, (size_t)M * N, M * N, C, scratch + (size_t)pl.chunks * M * N));
  return DSBDD_OK;
}

struct TrainScratch {
  float *dz2, *a1, *dz1, *partA, *partB, *rtmp, *wg, *gd, *gxr, *gxc, *gm, *agg_head, *xagg, *xagg_head, *vec;
  size_t bytes, wg_floats;
};
static int train_grid(int64_t E) {
  // persistent workgroups of the backward edge kernels per CU (DSBDD_TRAIN_WG_PER_CU, default 1; the kernels fit two:
  // 256 VGPRs, 74 KB of LDS)
  static const int per_cu = [] { const char* v = getenv("DSBDD_TRAIN_WG_PER_CU"); return v && atoi(v) >= 1 && atoi(v) <= 4 ? atoi(v) : 1; }();
  int64_t tiles = (E + 127) / 128;
  int64_t cap = (int64_t)per_cu * device_cus();
  int64_t g = tiles < cap ? tiles : cap;
  return g < 1 ? 1 : (int)g;
}
static TrainScratch carve_train(char* base, int H, int64_t N, int64_t E) {
  TrainScratch t{};
  size_t off = 0;
  auto take = [&](size_t floats) { float* p = base ? reinterpret_cast<float*>(base + off) : nullptr; off += al256(floats * 4); return p; };
  const size_t EH = (size_t)(E > 0 ? E : 1) * H;
  const int slots = 256 * 8;
  t.dz2 = take(EH); t.a1 = take(EH); t.dz1 = take(EH);
  t.partA = take((size_t)slots * kPartAll * H); t.partB = t.partA;      // one [8][H] slot per workgroup for both kernels
  t.rtmp = take((size_t)(slots / 32 + 1) * kPartAll * H);
  t.wg_floats = wgrad_floats_upto(E > N ? E : N, H, H);   // any K <= max(E, N): the coordinate stage runs on an edge prefix
  t.wg = take(t.wg_floats);
  t.gd = take(E + 1); t.gxr = take(3 * (size_t)E + 4); t.gxc = take(3 * (size_t)E + 4); t.gm = take(3 * (size_t)E + 4);
  t.agg_head = take((size_t)((E + 31) / 32 + 2) * H);
  t.xagg = take(2 * (3 * (size_t)N + 4)); t.xagg_head = take(2 * 4 * (size_t)((E + 31) / 32 + 2));   // one sum per MLP (pass split)
  t.vec = take(8 * (size_t)H);
  t.bytes = off;
  return t;
}

template <int H>
static hipError_t launch_bwd_a(hipStream_t s, int mode, const TrainEdgeArgs& a, int grid) {
  if (mode == MODE_GCL) hipLaunchKernelGGL((edge_bwd_a_kernel<H, MODE_GCL>), dim3(grid), dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((edge_bwd_a_kernel<H, MODE_COORD>), dim3(grid), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}
static hipError_t launch_bwd_a(int H, hipStream_t s, int mode, const TrainEdgeArgs& a, int grid) {
  switch (H) {
    case 64: return launch_bwd_a<64>(s, mode, a, grid);
    case 128: return launch_bwd_a<128>(s, mode, a, grid);
    case 192: return launch_bwd_a<192>(s, mode, a, grid);
    case 256: return launch_bwd_a<256>(s, mode, a, grid);
  }
  return hipErrorInvalidValue;
}
static hipError_t launch_bwd_b(int H, hipStream_t s, const TrainEdgeArgs& a, int grid) {
  switch (H) {
    case 64: hipLaunchKernelGGL((edge_bwd_b_kernel<64>), dim3(grid), dim3(kThreads), 0, s, a); break;
    case 128: hipLaunchKernelGGL((edge_bwd_b_kernel<128>), dim3(grid), dim3(kThreads), 0, s, a); break;
    case 192: hipLaunchKernelGGL((edge_bwd_b_kernel<192>), dim3(grid), dim3(kThreads), 0, s, a); break;
    case 256: hipLaunchKernelGGL((edge_bwd_b_kernel<256>), dim3(grid), dim3(kThreads), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static hipError_t launch_bwd_e(int H, hipStream_t s, int mode, const TrainEdgeArgs& a, const float* z2, int grid) {
  if (mode == MODE_COORD) {
    switch (H) {
      case 64: hipLaunchKernelGGL((edge_bwd_ec_kernel<64>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
      case 128: hipLaunchKernelGGL((edge_bwd_ec_kernel<128>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
      case 192: hipLaunchKernelGGL((edge_bwd_ec_kernel<192>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
      case 256: hipLaunchKernelGGL((edge_bwd_ec_kernel<256>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (H) {
    case 64: hipLaunchKernelGGL((edge_bwd_e_kernel<64>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
    case 128: hipLaunchKernelGGL((edge_bwd_e_kernel<128>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
    case 192: hipLaunchKernelGGL((edge_bwd_e_kernel<192>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
    case 256: hipLaunchKernelGGL((edge_bwd_e_kernel<256>), dim3(grid), dim3(kThreadsE), 0, s, a, z2); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static bool train_h_ok(int H) { return H == 64 || H == 128 || H == 192 || H == 256; }
static bool graph_ok(const dsbdd_train_graph* g) {
  return g && g->erow && g->ecol && g->ed0 && g->row_ptr && g->deg && g->node_batch && g->lig_off && g->poc_off &&
         g->n_nodes > 0 && g->n_edges >= 0 && g->batch > 0 && g->n_lig >= 0 && g->n_lig <= g->n_nodes;
}
static bool mlp_ok(const dsbdd_train_mlp* m) {
  return m && m->P && m->Q && m->wd && m->wd0 && m->tab && m->W2 && m->W2T && m->b2 && (m->ldpq & 3) == 0;
}

// Side streams of the training backward (round 6, dsbdd_train_net_backward only).  The weight gradients leave the
// dependency chain of the input gradients, and the two edge MLPs of a coordinate stage are independent of each other:
// `wg` takes the weight-gradient launches, `co` the second MLP's chain, fork / join by events.  Every kernel and every
// reduction order is the one of the single-stream sequence (DSBDD_TRAIN_STREAMS=0), so the gradients keep their bits.
enum { SIDE_CO = 1, SIDE_NODE_WG = 2, SIDE_COORD_WG = 4, SIDE_GCL_WG = 8 };
struct TrainSide {
  hipStream_t wg = nullptr, co = nullptr;
  int mask = 0;                  // SIDE_* : what runs beside the main chain
  int device = -1;               // the device the streams live on (a module moved to another GPU gets new ones)
  hipEvent_t w2_done[2] = {nullptr, nullptr};   // after the last W2 weight gradient queued on `wg` that reads scratch set q
  bool w2_valid[2] = {false, false};
  std::vector<hipEvent_t> ev;
  size_t next = 0;
  // everything queued on `to` after this call starts after everything queued on `from` before it
  hipError_t link(hipStream_t from, hipStream_t to) {
    hipEvent_t e = ev[next++ % ev.size()];
    const hipError_t r = hipEventRecord(e, from);
    return r != hipSuccess ? r : hipStreamWaitEvent(to, e, 0);
  }
  hipError_t create() {
    hipError_t r = hipGetDevice(&device);
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&wg, hipStreamNonBlocking);
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&co, hipStreamNonBlocking);
    ev.resize(32);
    for (size_t i = 0; r == hipSuccess && i < ev.size(); ++i) r = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    for (int q = 0; r == hipSuccess && q < 2; ++q) r = hipEventCreateWithFlags(&w2_done[q], hipEventDisableTiming);
    w2_valid[0] = w2_valid[1] = false;
    return r;
  }
  void destroy() {
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    ev.clear();
    for (int q = 0; q < 2; ++q) { if (w2_done[q]) (void)hipEventDestroy(w2_done[q]); w2_done[q] = nullptr; w2_valid[q] = false; }
    if (wg) (void)hipStreamDestroy(wg);
    if (co) (void)hipStreamDestroy(co);
    wg = co = nullptr;
  }
};

// the backward of ONE edge MLP: kernel A / E -> weight gradient -> kernel B -> node gathers; returns the per-edge gradient
// w.r.t. the current squared distance in ts.gd.  `sd` (optional) -- what waits for what:
//   * kernel A / E overwrites ts.dz2 / ts.a1 of scratch set `set`: it waits for the last W2 weight gradient that read this
//     set on the side stream (its own event), NOT for the whole side stream -- the node-level weight gradients queued there
//     keep running beside the memory-bound kernel E (waiting for them cost the main chain 60 - 70 us per stage);
//   * the message stage's W2 gradient stays on this stream and is preceded by a FULL join of the side stream: it streams
//     373 MB and runs 1.6 x longer with anything beside it, and the join is the one point per block that covers the
//     caller's hazards (dsbdd_train_net_backward: dout, dz / xcat, d_pq / d_pq4 are overwritten after it only);
//   * a coordinate stage's W2 gradient goes to the side stream and records the set's event.
static int mlp_backward(hipStream_t s, int H, int mode, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                        int64_t E, TrainEdgeArgs a, const dsbdd_train_mlp_grad* out, const TrainScratch& ts,
                        TrainSide* sd = nullptr, bool linked = false, const float* z2 = nullptr, int set = 0) {
  const int grid = train_grid(E);
  const bool side_w = sd && (sd->mask & (mode == MODE_GCL ? SIDE_GCL_WG : SIDE_COORD_WG));
  const bool late_join = sd && mode == MODE_GCL && !side_w;      // the full join sits in front of the W2 gradient instead
  if (sd && !linked && !late_join) { HIP_TRY(sd->link(sd->wg, s)); HIP_TRY(sd->link(sd->co, s)); }
  if (sd && sd->w2_valid[set]) HIP_TRY(hipStreamWaitEvent(s, sd->w2_done[set], 0));
  const int slots = grid;                  // one partial-vector slot per workgroup
  a.erow = g->erow; a.ecol = g->ecol; a.ed0 = g->ed0; a.E = (int)E; a.x = x; a.n_lig = (int)g->n_lig;
  a.n_nodes = (int)g->n_nodes; a.P = m->P; a.Q = m->Q; a.ldpq = m->ldpq; a.wd = m->wd; a.wd0 = m->wd0; a.table = m->tab;
  a.b2 = m->b2; a.head = m->head; a.head_b = m->head_b;
  a.a1_out = ts.a1; a.gxr = ts.gxr; a.gxc = ts.gxc; a.gd = ts.gd; a.gd0 = out->gd0;
  // A: dz2, a1, partial bias / head vectors
  a.Bmat = m->W2T; a.dz_out = ts.dz2; a.part = ts.partA;
  if (z2) HIP_TRY(launch_bwd_e(H, s, mode, a, z2, grid));     // the forward pass kept z2: no H x H layer here
  else HIP_TRY(launch_bwd_a(H, s, mode, a, grid));
  // dW2[f][i] = sum_e dz2[e][f] a1[e][i]
  if (side_w) HIP_TRY(sd->link(s, sd->wg));
  if (late_join) { HIP_TRY(sd->link(sd->wg, s)); HIP_TRY(sd->link(sd->co, s)); }
  { const int rc = wgrad_impl(side_w ? sd->wg : s, ts.dz2, H, ts.a1, H, E, H, H, out->d_W2, ts.wg, ts.wg_floats); if (rc != DSBDD_OK) return rc; }
  if (side_w) { HIP_TRY(hipEventRecord(sd->w2_done[set], sd->wg)); sd->w2_valid[set] = true; }
  // B: dz1, partial first-layer vectors, per-edge distance gradients
  a.Bmat = m->W2; a.dz_in = ts.dz2; a.dz_out = ts.dz1; a.part = ts.partB;
  HIP_TRY(launch_bwd_b(H, s, a, grid));
  // ONE ordered reduction for both kernels' partial vectors (they share the grid and the [8][H] slots) -> d_vec [8][H]
  HIP_TRY(reduce_parts(s, ts.partB, slots, (size_t)kPartAll * H, kPartAll * H, out->d_vec, ts.rtmp));
  // dP / dQ
  const int N = (int)g->n_nodes;
  hipLaunchKernelGGL(rows_gather_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, s, (const float*)ts.dz1, H, g->row_ptr,
                     g->deg, g->rev, (int)E, N, out->dP, out->dQ, out->ldo);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

extern "C" {

size_t dsbdd_train_scratch_bytes(int32_t H, int64_t n_nodes, int64_t n_edges) {
  if (!train_h_ok(H) || n_nodes < 1 || n_edges < 0) return 0;
  return carve_train(nullptr, H, n_nodes, n_edges).bytes;
}

size_t dsbdd_train_wgrad_scratch_bytes(int64_t K, int64_t M, int64_t N) {
  if (K < 1 || M < 1 || N < 1) return 0;
  return wgrad_floats_upto(K, M, N) * 4;      // covers every K' <= K (the plan is not monotonic in K)
}

size_t dsbdd_train_wgrad_plan_bytes(int64_t K, int64_t M, int64_t N) {
  if (K < 1 || M < 1 || N < 1) return 0;
  return wgrad_plan(K, M, N).floats * 4;      // what a call with exactly this K writes (tests: <= the bound above)
}

int dsbdd_train_edge_rev(void* stream, const dsbdd_train_graph* g, int32_t* rev) {
  StreamDevice stream_device_(stream);
  if (!graph_ok(g) || !rev) return fail(DSBDD_ERR_ARG, "bad argument");
  if (g->n_edges == 0) return DSBDD_OK;
  hipLaunchKernelGGL(edge_rev_kernel, dim3((unsigned)((g->n_edges + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g->erow, g->ecol, g->row_ptr, g->deg, (int)g->n_edges,
                     (int)g->n_nodes, rev);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_train_sample_mean(void* stream, const float* x, const dsbdd_train_graph* g, float* mean) {
  StreamDevice stream_device_(stream);
  if (!graph_ok(g) || !x || !mean) return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(sample_mean_kernel, dim3((unsigned)g->batch), dim3(kThreads), 0, static_cast<hipStream_t>(stream), x,
                     g->lig_off, g->poc_off, (int)g->n_lig, mean);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

static int gcl_forward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                            float norm_factor, float* agg, void* scratch, size_t scratch_bytes, float* z2_store);
int dsbdd_train_gcl_forward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                            float norm_factor, float* agg, void* scratch, size_t scratch_bytes) {
  return gcl_forward_impl(stream, H, g, m, x, norm_factor, agg, scratch, scratch_bytes, nullptr);
}
static int gcl_forward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                            float norm_factor, float* agg, void* scratch, size_t scratch_bytes, float* z2_store) {
  StreamDevice stream_device_(stream);
  if (!train_h_ok(H) || !graph_ok(g) || !mlp_ok(m) || !x || !agg || !scratch) return fail(DSBDD_ERR_ARG, "bad argument");
  const TrainScratch ts = carve_train(static_cast<char*>(scratch), H, g->n_nodes, g->n_edges);
  if (ts.bytes > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small (dsbdd_train_scratch_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = (int)g->n_nodes;
  EdgeArgs ea{};
  ea.erow = g->erow; ea.ecol = g->ecol; ea.ed0 = g->ed0; ea.e_count = g->row_ptr + N; ea.e_cap = (int)g->n_edges;
  ea.x = x; ea.n_lig = (int)g->n_lig; ea.n_nodes = N; ea.ldpq = m->ldpq;
  ea.mlp[0] = EdgeMlpW{m->P, m->Q, m->wd, m->wd0, m->tab, m->W2T, m->b2, nullptr};
  ea.mlp[1] = ea.mlp[0];
  ea.att_w = m->head; ea.att_b = m->head_b; ea.attention = m->head != nullptr;
  ea.agg = agg; ea.agg_head = ts.agg_head; ea.norm_factor = norm_factor; ea.z2_out = z2_store;
  if (g->n_edges > 0) HIP_TRY(launch_edge_plain(H, s, MODE_GCL, ea, g->n_edges));
  hipLaunchKernelGGL(agg_complete_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, s, agg, (const float*)ts.agg_head,
                     g->row_ptr, g->deg, N, (int)H, (int)((g->n_edges + 31) / 32 + 1), 5);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

static int gcl_backward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                             float norm_factor, const float* d_agg, const dsbdd_train_mlp_grad* out, float* d_x,
                             void* scratch, size_t scratch_bytes, TrainSide* sd, const float* z2 = nullptr) {
  StreamDevice stream_device_(stream);
  if (!train_h_ok(H) || !graph_ok(g) || !g->rev || !mlp_ok(m) || !x || !d_agg || !out || !out->dP || !out->dQ ||
      !out->d_vec || !out->d_W2 || !out->gd0 || (out->ldo & 3) || !d_x || !scratch)
    return fail(DSBDD_ERR_ARG, "bad argument");
  const TrainScratch ts = carve_train(static_cast<char*>(scratch), H, g->n_nodes, g->n_edges);
  if (ts.bytes > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small (dsbdd_train_scratch_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TrainEdgeArgs a{};
  a.d_agg = d_agg; a.norm_factor = norm_factor;
  { const int rc = mlp_backward(s, H, MODE_GCL, g, m, x, g->n_edges, a, out, ts, sd, false, z2); if (rc != DSBDD_OK) return rc; }
  const int N = (int)g->n_nodes;
  hipLaunchKernelGGL(edge_to_node3_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, s, (const float*)ts.gd,
                     (const float*)nullptr, (const float*)nullptr, x, g->ecol, g->row_ptr, g->deg, g->rev,
                     (int)g->n_edges, N, d_x, 0);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

static int coord_forward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, int32_t n_mlp,
                              const float* x, const float* mean, int64_t n_upd, float norm_constant, float coords_range,
                              int32_t use_tanh, float norm_factor, float* x_out, void* scratch, size_t scratch_bytes,
                              float* z2_store, size_t z2_stride);
int dsbdd_train_coord_forward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, int32_t n_mlp,
                              const float* x, const float* mean, int64_t n_upd, float norm_constant, float coords_range,
                              int32_t use_tanh, float norm_factor, float* x_out, void* scratch, size_t scratch_bytes) {
  return coord_forward_impl(stream, H, g, m, n_mlp, x, mean, n_upd, norm_constant, coords_range, use_tanh, norm_factor, x_out,
                            scratch, scratch_bytes, nullptr, 0);
}
static int coord_forward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, int32_t n_mlp,
                              const float* x, const float* mean, int64_t n_upd, float norm_constant, float coords_range,
                              int32_t use_tanh, float norm_factor, float* x_out, void* scratch, size_t scratch_bytes,
                              float* z2_store, size_t z2_stride) {
  StreamDevice stream_device_(stream);
  if (!train_h_ok(H) || !graph_ok(g) || n_mlp < 1 || n_mlp > 2 || !mlp_ok(m) || (n_mlp == 2 && (!mlp_ok(m + 1) || !mean)) ||
      !m->head || !x || !x_out || n_upd < 0 || n_upd > g->n_nodes || !scratch)
    return fail(DSBDD_ERR_ARG, "bad argument");
  const TrainScratch ts = carve_train(static_cast<char*>(scratch), H, g->n_nodes, g->n_edges);
  if (ts.bytes > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small (dsbdd_train_scratch_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = (int)g->n_nodes;
  HIP_TRY(hipMemcpyAsync(x_out, x, (size_t)N * 12, hipMemcpyDeviceToDevice, s));
  if (n_upd == 0 || g->n_edges == 0) return DSBDD_OK;
  EdgeArgs ea{};
  ea.erow = g->erow; ea.ecol = g->ecol; ea.ed0 = g->ed0; ea.e_count = g->row_ptr + n_upd; ea.e_cap = (int)g->n_edges;
  ea.x = x; ea.n_lig = (int)g->n_lig; ea.n_nodes = N; ea.ldpq = m->ldpq;
  for (int q = 0; q < 2; ++q) {
    const dsbdd_train_mlp& mq = m[q < n_mlp ? q : 0];
    ea.mlp[q] = EdgeMlpW{mq.P, mq.Q, mq.wd, mq.wd0, mq.tab, mq.W2T, mq.b2, nullptr};
  }
  ea.w3 = m->head; ea.node_batch = g->node_batch; ea.mean = mean; ea.norm_constant = norm_constant;
  ea.coords_range = coords_range; ea.use_tanh = use_tanh; ea.n_mlp = n_mlp; ea.xagg = ts.xagg; ea.xagg_head = ts.xagg_head;
  ea.xagg_stride = (size_t)N * 3; ea.xhead_stride = 4 * (size_t)((g->n_edges + 31) / 32 + 2); ea.norm_factor = norm_factor;
  ea.z2_out = z2_store; ea.z2_stride = z2_stride;
  // two MLPs: alternate workgroups take one MLP of a tile each (the updated rows' edge prefix is a fraction of a tile per
  // CU: twice as many, half as long work items), one coordinate sum per MLP, added by coord_update_kernel
  ea.pass_split = n_mlp == 2 ? 1 : 0;
  HIP_TRY(launch_edge_plain(H, s, MODE_COORD, ea, g->n_edges));
  hipLaunchKernelGGL(coord_update_kernel, dim3((unsigned)((3 * n_upd + 255) / 256)), dim3(256), 0, s, x_out,
                     (const float*)ts.xagg, (const float*)ts.xagg_head, ea.pass_split ? 2 : 1, ea.xagg_stride, ea.xhead_stride, g->row_ptr,
                     g->deg, (int)(3 * n_upd), (int)((g->n_edges + 31) / 32 + 1), 5);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

static int coord_backward_impl(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, int32_t n_mlp,
                               const float* x, const float* mean, int64_t n_upd, int64_t e_upd, float norm_constant,
                               float coords_range, int32_t use_tanh, float norm_factor, const float* d_xout,
                               const dsbdd_train_mlp_grad* out, float* d_x, float* d_mean, void* scratch,
                               size_t scratch_bytes, TrainSide* sd, const float* z2 = nullptr, size_t z2_stride = 0) {
  StreamDevice stream_device_(stream);
  if (!train_h_ok(H) || !graph_ok(g) || !g->rev || n_mlp < 1 || n_mlp > 2 || !mlp_ok(m) ||
      (n_mlp == 2 && (!mlp_ok(m + 1) || !mean || !d_mean)) || !m->head || !x || !d_xout || !out || !d_x || n_upd < 0 ||
      n_upd > g->n_nodes || e_upd < 0 || e_upd > g->n_edges || !scratch)
    return fail(DSBDD_ERR_ARG, "bad argument");
  for (int q = 0; q < n_mlp; ++q)
    if (!out[q].dP || !out[q].dQ || !out[q].d_vec || !out[q].d_W2 || !out[q].gd0 || (out[q].ldo & 3))
      return fail(DSBDD_ERR_ARG, "bad gradient destination");
  const TrainScratch ts = carve_train(static_cast<char*>(scratch), H, g->n_nodes, g->n_edges);
  if (ts.bytes > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small (dsbdd_train_scratch_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int N = (int)g->n_nodes;
  // with side streams and two MLPs the second chain (kernel A .. node gathers) runs on sd->co out of its own scratch
  // (the second half of `scratch`, 2 x dsbdd_train_scratch_bytes) while the first runs here; d_x is still added in the
  // order q = 0, 1 on this stream
  const bool two = sd && (sd->mask & SIDE_CO) && n_mlp == 2 && scratch_bytes >= 2 * ts.bytes;
  const TrainScratch ts1 = two ? carve_train(static_cast<char*>(scratch) + ts.bytes, H, g->n_nodes, g->n_edges) : ts;
  if (two) HIP_TRY(sd->link(s, sd->co));
  for (int q = 0; q < n_mlp; ++q) {
    TrainEdgeArgs a{};
    a.d_xagg = d_xout; a.node_batch = g->node_batch; a.mean = mean; a.norm_constant = norm_constant;
    a.coords_range = coords_range; a.use_tanh = use_tanh; a.which = q; a.norm_factor = norm_factor;
    const TrainScratch& tq = q == 1 ? ts1 : ts;
    a.gm = q == 1 ? tq.gm : nullptr;
    dsbdd_train_mlp mq = m[q];
    mq.head = m[0].head;                      // the output layer is shared by both MLPs (egnn_new.py:78,85,91)
    hipStream_t sq = two && q == 1 ? sd->co : s;
    { const int rc = mlp_backward(sq, H, MODE_COORD, g, &mq, x, e_upd, a, out + q, tq, sd, true, z2 ? z2 + (size_t)q * z2_stride : nullptr,
                                  two && q == 1 ? 1 : 0);
      if (rc != DSBDD_OK) return rc; }
    if (sq != s) HIP_TRY(sd->link(sq, s));
    hipLaunchKernelGGL(edge_to_node3_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, s, (const float*)tq.gd,
                       (const float*)tq.gxr, (const float*)tq.gxc, x, g->ecol, g->row_ptr, g->deg, g->rev, (int)e_upd, N,
                       d_x, q);
    HIP_TRY(hipGetLastError());
    if (q == 1) {
      hipLaunchKernelGGL(sample_edge_sum3_kernel, dim3((unsigned)g->batch), dim3(kThreads), 0, s, (const float*)tq.gm,
                         g->row_ptr, g->lig_off, g->poc_off, (int)g->n_lig, (int)e_upd, d_mean);
      HIP_TRY(hipGetLastError());
    }
  }
  return DSBDD_OK;
}

int dsbdd_train_gcl_backward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, const float* x,
                             float norm_factor, const float* d_agg, const dsbdd_train_mlp_grad* out, float* d_x,
                             void* scratch, size_t scratch_bytes) {
  return gcl_backward_impl(stream, H, g, m, x, norm_factor, d_agg, out, d_x, scratch, scratch_bytes, nullptr);
}

int dsbdd_train_coord_backward(void* stream, int32_t H, const dsbdd_train_graph* g, const dsbdd_train_mlp* m, int32_t n_mlp,
                               const float* x, const float* mean, int64_t n_upd, int64_t e_upd, float norm_constant,
                               float coords_range, int32_t use_tanh, float norm_factor, const float* d_xout,
                               const dsbdd_train_mlp_grad* out, float* d_x, float* d_mean, void* scratch,
                               size_t scratch_bytes) {
  return coord_backward_impl(stream, H, g, m, n_mlp, x, mean, n_upd, e_upd, norm_constant, coords_range, use_tanh, norm_factor,
                             d_xout, out, d_x, d_mean, scratch, scratch_bytes, nullptr);
}

int dsbdd_train_radial_backward(void* stream, const dsbdd_train_graph* g, const float* x, const float* gd, float* d_x) {
  StreamDevice stream_device_(stream);
  if (!graph_ok(g) || !g->rev || !x || !gd || !d_x) return fail(DSBDD_ERR_ARG, "bad argument");
  const int N = (int)g->n_nodes;
  hipLaunchKernelGGL(edge_to_node3_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, static_cast<hipStream_t>(stream), gd,
                     (const float*)nullptr, (const float*)nullptr, x, g->ecol, g->row_ptr, g->deg, g->rev, (int)g->n_edges,
                     N, d_x, 0);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_train_wgrad(void* stream, const float* A, int32_t lda, const float* B, int32_t ldb, int64_t K, int32_t M,
                      int32_t N, float* C, void* scratch, size_t scratch_bytes) {
  StreamDevice stream_device_(stream);
  if (!A || !B || !C || K < 1 || M < 1 || N < 1 || lda < M || ldb < N || !scratch) return fail(DSBDD_ERR_ARG, "bad argument");
  if (wgrad_plan(K, M, N).floats * 4 > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small");
  return wgrad_impl(static_cast<hipStream_t>(stream), A, lda, B, ldb, K, M, N, C, static_cast<float*>(scratch), scratch_bytes / 4);
}

int dsbdd_train_colsum(void* stream, const float* A, int32_t lda, int64_t M, int32_t N, float* out, void* scratch,
                       size_t scratch_bytes) {
  StreamDevice stream_device_(stream);
  if (!A || !out || M < 1 || N < 1 || lda < N || !scratch) return fail(DSBDD_ERR_ARG, "bad argument");
  if ((size_t)((M + 31) / 32) * N * 4 > scratch_bytes) return fail(DSBDD_ERR_CAPACITY, "scratch too small");
  HIP_TRY(reduce_parts(static_cast<hipStream_t>(stream), A, (int)M, (size_t)lda, N, out, static_cast<float*>(scratch)));
  return DSBDD_OK;
}

}  // extern "C"

// ---- the training step as one launch sequence per direction (round 6) ------------------------------------------------
#include "train_net.h"

// ---- the loss terms of the pocket-conditioned training step around the network call (round 6) --------------------------
#include "loss_head.h"

static LossCfg loss_cfg_of(const dsbdd_loss_cfg* c) {
  return LossCfg{c->batch, c->n_lig, c->n_pocket, c->atom_nf, c->residue_nf, c->timesteps, c->remove_com, c->vnode_idx,
                 c->norm_value_x, c->norm_value_h, c->norm_bias_h, c->n1_tab, c->n2_tab};
}
static bool loss_cfg_ok(const dsbdd_loss_cfg* c) {
  return c && c->batch > 0 && c->n_lig >= 0 && c->n_pocket >= 0 && c->atom_nf > 0 && c->residue_nf > 0 && c->timesteps > 0 &&
         c->norm_value_x > 0.f && c->norm_value_h > 0.f && c->vnode_idx < c->atom_nf;
}

extern "C" {

int dsbdd_edge_capacity(void* stream, const int64_t* lig_mask, int64_t n_lig, const int64_t* pocket_mask, int64_t n_pocket,
                        int64_t batch, int64_t* out) {
  StreamDevice stream_device_(stream);
  if (!out || batch < 1 || n_lig < 0 || n_pocket < 0 || (n_lig > 0 && !lig_mask) || (n_pocket > 0 && !pocket_mask))
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(edge_capacity_kernel, dim3(1), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(lig_mask), (int)n_lig, reinterpret_cast<const long long*>(pocket_mask),
                     (int)n_pocket, (int)batch, reinterpret_cast<long long*>(out));
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_loss_rows(void) { return LS_ROWS; }
int dsbdd_loss_out_rows(void) { return LO_ROWS; }

int dsbdd_loss_cond_pre(void* stream, const dsbdd_loss_cfg* cfg, const float* lig_x, const float* lig_h, const int64_t* lig_mask,
                        const float* pocket_x, const float* pocket_h, const int64_t* pocket_mask, const float* eps,
                        const float* t_int, const float* gamma_table, const float* logpn_table, float* z_t, float* xh_pocket,
                        float* per_sample, float* lig_x_norm, float* lig_h_norm, float* pocket_x_norm, float* pocket_h_norm) {
  StreamDevice stream_device_(stream);
  if (!loss_cfg_ok(cfg) || !t_int || !gamma_table || !per_sample || (cfg->n_lig > 0 && (!lig_x || !lig_h || !lig_mask || !eps || !z_t)) ||
      (cfg->n_pocket > 0 && (!pocket_x || !pocket_h || !pocket_mask || !xh_pocket)) || (logpn_table && (cfg->n1_tab < 1 || cfg->n2_tab < 1)))
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(loss_cond_pre_kernel, dim3((unsigned)cfg->batch), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream),
                     loss_cfg_of(cfg), lig_x, lig_h, reinterpret_cast<const long long*>(lig_mask), pocket_x, pocket_h,
                     reinterpret_cast<const long long*>(pocket_mask), eps, t_int, gamma_table, logpn_table, z_t, xh_pocket, per_sample,
                     lig_x_norm, lig_h_norm, pocket_x_norm, pocket_h_norm);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_loss_cond_post(void* stream, const dsbdd_loss_cfg* cfg, const float* net, const float* eps, const float* z_t,
                         const float* lig_h, const int64_t* lig_mask, const float* per_sample, float* xh_hat, float* out) {
  StreamDevice stream_device_(stream);
  if (!loss_cfg_ok(cfg) || !per_sample || !out || (cfg->n_lig > 0 && (!net || !eps || !z_t || !lig_h || !lig_mask || !xh_hat)))
    return fail(DSBDD_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(loss_cond_post_kernel, dim3((unsigned)cfg->batch), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream),
                     loss_cfg_of(cfg), net, eps, z_t, lig_h, reinterpret_cast<const long long*>(lig_mask), per_sample, xh_hat, out);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_loss_cond_post_backward(void* stream, const dsbdd_loss_cfg* cfg, const float* net, const float* eps, const float* lig_h,
                                  const int64_t* lig_mask, const float* per_sample, const float* g_err, const float* g_l0x,
                                  const float* g_hat, float* d_net) {
  StreamDevice stream_device_(stream);
  if (!loss_cfg_ok(cfg) || !per_sample || (cfg->n_lig > 0 && (!net || !eps || !lig_h || !lig_mask || !d_net)))
    return fail(DSBDD_ERR_ARG, "bad argument");
  if (cfg->n_lig == 0) return DSBDD_OK;
  const size_t n = (size_t)cfg->n_lig * (3 + cfg->atom_nf);
  unsigned grid = (unsigned)((n + kLossThreads - 1) / kLossThreads);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(loss_cond_post_bwd_kernel, dim3(grid), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream),
                     loss_cfg_of(cfg), net, eps, lig_h, reinterpret_cast<const long long*>(lig_mask), per_sample, g_err, g_l0x,
                     g_hat, d_net);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

}  // extern "C"
