"""Host-side handle on the HIP engine: packs an EGNNDynamics `state_dict` into
the kernel weight layout (include/diffsbdd_hip.h "weight slots"), owns the
device workspace (a torch uint8 tensor, i.e. torch's caching allocator owns the
memory), and enqueues `dsbdd_dynamics_forward` on torch's current stream.

PyTorch is plumbing here: tensors, streams, allocation.  All arithmetic of the
hot path runs in libdiffsbdd_hip.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _pad4(v):
    return (v + 3) & ~3


def make_config(atom_nf, residue_nf, joint_nf, hidden_nf, n_layers, inv_sublayers, attention,
                tanh, update_pocket_coords, reflection_equivariant, edge_embedding_dim,
                edge_cutoff_ligand, edge_cutoff_pocket, edge_cutoff_interaction, norm_constant,
                normalization_factor, coords_range=15.0):
    cfg = _lib.Config()
    cfg.atom_nf, cfg.residue_nf, cfg.joint_nf, cfg.hidden_nf = atom_nf, residue_nf, joint_nf, hidden_nf
    cfg.n_layers, cfg.inv_sublayers = n_layers, inv_sublayers
    cfg.attention, cfg.use_tanh = int(bool(attention)), int(bool(tanh))
    cfg.update_pocket_coords = int(bool(update_pocket_coords))
    cfg.reflection_equivariant = int(bool(reflection_equivariant))
    cfg.edge_embedding_dim = int(edge_embedding_dim or 0)
    for name, v in (("ligand", edge_cutoff_ligand), ("pocket", edge_cutoff_pocket),
                    ("interaction", edge_cutoff_interaction)):
        setattr(cfg, "has_cutoff_" + name, int(v is not None))
        setattr(cfg, "cutoff_" + name, float(v) if v is not None else 0.0)
    cfg.norm_constant = float(norm_constant)
    cfg.normalization_factor = float(normalization_factor)
    cfg.coords_range = float(coords_range)
    return cfg


def pack_weights(sd, cfg, device):
    """state_dict (reference key names) -> list of device tensors
    in slot order.  Pure layout work: transposes, splits of the first edge-MLP
    layer into its row / col / distance / edge-type parts (the exact
    factorisation of Linear(cat[h_i, h_j, e]), egnn_new.py:15,35,99), padding."""
    H, J = cfg.hidden_nf, cfg.joint_nf
    JP = _pad4(J + 1)
    enf = cfg.edge_embedding_dim
    f32 = dict(dtype=torch.float32, device=device)

    def g(name):
        return sd[name].detach().to(**f32)

    def wt(name, rows_to=None, cols_to=None):
        w = g(name).t().contiguous()              # [in][out]
        k, n = w.shape
        rows, cols = (rows_to or k), (cols_to or _pad4(n))
        out = torch.zeros((rows, cols), **f32)
        out[:k, :n] = w
        return out

    def vec(name, to=None):
        v = g(name).reshape(-1)
        if to is not None and to != v.numel():
            out = torch.zeros(to, **f32)
            out[:v.numel()] = v
            return out
        return v.contiguous()

    emb = g("edge_embedding.weight") if enf else None   # [3][enf]

    def first_layer(prefix):
        """edge MLP layer 0: weight [H][2H + 2 + enf], bias [H]."""
        w, b = g(prefix + ".weight"), g(prefix + ".bias")
        row_t = w[:, :H].t()
        col_t = w[:, H:2 * H].t()
        wd = w[:, 2 * H].contiguous()
        wd0 = w[:, 2 * H + 1].contiguous()
        tab = b.unsqueeze(0).repeat(3, 1)
        if enf:
            tab = tab + emb @ w[:, 2 * H + 2:].t()   # [3][enf] @ [enf][H]
        return row_t, col_t, wd, wd0, tab.contiguous()

    slots = []
    for nm in ("atom_encoder", "residue_encoder", "atom_decoder", "residue_decoder"):
        slots += [wt(nm + ".0.weight"), vec(nm + ".0.bias"), wt(nm + ".2.weight"), vec(nm + ".2.bias")]
    slots += [wt("egnn.embedding.weight", rows_to=JP, cols_to=H), vec("egnn.embedding.bias"),
              wt("egnn.embedding_out.weight", cols_to=JP), vec("egnn.embedding_out.bias", to=JP)]
    assert len(slots) == len(_lib.G_NAMES)
    for i in range(cfg.n_layers):
        for s in range(cfg.inv_sublayers):
            p = f"egnn.e_block_{i}.gcl_{s}"
            row_t, col_t, wd, wd0, tab = first_layer(p + ".edge_mlp.0")
            blk = [torch.cat([row_t, col_t], 1).contiguous(), wd, wd0, tab,
                   wt(p + ".edge_mlp.2.weight"), vec(p + ".edge_mlp.2.bias")]
            if cfg.attention:
                blk += [vec(p + ".att_mlp.0.weight"), vec(p + ".att_mlp.0.bias")]
            else:
                blk += [None, None]
            blk += [wt(p + ".node_mlp.0.weight"), vec(p + ".node_mlp.0.bias"),
                    wt(p + ".node_mlp.2.weight"), vec(p + ".node_mlp.2.bias")]
            assert len(blk) == len(_lib.GCL_NAMES)
            slots += blk
        p = f"egnn.e_block_{i}.gcl_equiv"
        c_row, c_col, c_wd, c_wd0, c_tab = first_layer(p + ".coord_mlp.0")
        if cfg.reflection_equivariant:
            eq = [torch.cat([c_col, c_row], 1).contiguous(), c_wd, c_wd0, c_tab,
                  wt(p + ".coord_mlp.2.weight"), vec(p + ".coord_mlp.2.bias"),
                  None, None, None, None, None]
        else:
            x_row, x_col, x_wd, x_wd0, x_tab = first_layer(p + ".cross_product_mlp.0")
            eq = [torch.cat([c_col, x_col, c_row, x_row], 1).contiguous(), c_wd, c_wd0, c_tab,
                  wt(p + ".coord_mlp.2.weight"), vec(p + ".coord_mlp.2.bias"),
                  x_wd, x_wd0, x_tab,
                  wt(p + ".cross_product_mlp.2.weight"), vec(p + ".cross_product_mlp.2.bias")]
        eq.append(vec(p + ".coord_mlp.4.weight"))
        assert len(eq) == len(_lib.EQ_NAMES)
        slots += eq
    return slots


def edge_capacity(mask_lig, mask_pocket, batch, check_sorted=True):
    """Upper bound on the length of the engine's edge list: the complete graph
    inside every sample (dynamics.py:170-172, self loops included), with the edges
    of each (sample, node set) segment rounded up to a multiple of 32 (the engine
    starts every segment at a wave-tile boundary, csrc/graph.h).  One host sync;
    sampling chains compute it once.  The same sync validates the masks: the
    kernels locate a sample's rows by binary search, so the masks must be sorted
    ascending with ids in [0, batch) (the reference's scatter ops would accept any
    order; here it is an error, raised before anything is launched)."""
    if mask_lig.is_cuda and check_sorted and mask_lig.dtype == torch.int64 and mask_pocket.dtype == torch.int64:
        # one launch + the one host sync (csrc/loss_head.h edge_capacity_kernel); the torch expressions below are ~ 25 launches
        lib = _lib.load()
        ml, mp = mask_lig.contiguous(), mask_pocket.contiguous()
        out = torch.empty(2, dtype=torch.int64, device=ml.device)
        _lib.check(lib.dsbdd_edge_capacity(torch.cuda.current_stream(ml.device).cuda_stream, ml.data_ptr(), ml.numel(),
                                           mp.data_ptr(), mp.numel(), int(batch), out.data_ptr()), "dsbdd_edge_capacity")
        ok_h, cap_h = out.tolist()
        if not ok_h:
            raise ValueError("batch masks must be sorted ascending with ids in [0, batch): the HIP kernels "
                             "locate a sample's rows by binary search")
        return int(cap_h)
    ok = torch.ones((), dtype=torch.bool, device=mask_lig.device)
    if check_sorted:
        for m in (mask_lig, mask_pocket):
            if m.numel() > 1:
                ok = ok & (m[1:] >= m[:-1]).all()
            if m.numel():
                ok = ok & (m[0] >= 0) & (m[-1] < batch)
    nl = torch.bincount(mask_lig, minlength=batch).to(torch.int64)
    np_ = torch.bincount(mask_pocket, minlength=batch).to(torch.int64)
    n = nl + np_
    seg = lambda v: (v + 31) // 32 * 32
    cap = (seg(nl * n) + seg(np_ * n)).sum()
    ok_h, cap_h = torch.stack((ok.to(torch.int64), cap)).tolist()          # the one host sync
    if not ok_h:
        raise ValueError("batch masks must be sorted ascending with ids in [0, batch): the HIP kernels "
                         "locate a sample's rows by binary search")
    return int(cap_h)


def frame_layout(sizes, representative, mask_pocket):
    """Index arrays of a pocket frame (include/diffsbdd_hip.h, dsbdd_engine_set_pocket_frame) from the pocket sizes
    [batch], the representative sample of every sample [batch] (a sample with an identical pocket, possibly itself)
    and the pocket mask [n_pocket]: (mask_frame [n_frame] = frame sample of every frame row, frame_rows [n_frame] =
    its row in the pocket array, twin [n_pocket] = the frame row standing for each pocket row, sizes of the frame
    samples).  The frame samples are the distinct representatives in ascending order.  Pure index arithmetic on
    whatever device the inputs live on."""
    dev = sizes.device
    sz = sizes.to(torch.int64)
    rep = representative.to(device=dev, dtype=torch.int64)
    batch = sz.numel()
    if rep.numel() != batch or bool((rep < 0).any()) or bool((rep >= batch).any()):
        raise ValueError("representative must name a sample of the batch for every sample")
    if bool((sz[rep] != sz).any()) or bool((rep[rep] != rep).any()):
        raise ValueError("a representative must have the same pocket size as the samples it stands for, "
                         "and must represent itself")
    off = torch.cumsum(sz, 0) - sz                                      # first pocket row of every sample
    reps = torch.unique(rep)                                            # sorted representative samples
    frame_id = torch.zeros(batch, dtype=torch.int64, device=dev)
    frame_id[reps] = torch.arange(reps.numel(), device=dev)
    sz_f = sz[reps]
    off_f = torch.cumsum(sz_f, 0) - sz_f                                # first frame row of every representative
    n3 = int(sz_f.sum().item())
    mask3 = torch.repeat_interleave(torch.arange(reps.numel(), device=dev), sz_f)
    frame_rows = (torch.arange(n3, device=dev) - off_f[mask3] + off[reps][mask3]).to(torch.int32).contiguous()
    mp = mask_pocket.to(device=dev, dtype=torch.int64)
    within = torch.arange(mp.numel(), device=dev) - off[mp]             # atom index inside its sample
    twin = (off_f[frame_id[rep]][mp] + within).to(torch.int32).contiguous()
    return mask3, frame_rows, twin, sz_f


class HipEngine:
    """One dsbdd_engine + its weights and workspace on one GPU."""

    def __init__(self, cfg, state_dict, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HipLibraryError(
                "the HIP engine needs a GPU device (got %s); there is no CPU fallback" % device)
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.lib.dsbdd_engine_create(C.byref(cfg), C.byref(h)), "dsbdd_engine_create")
        self.handle = h
        self.workspace = None
        self.caps = (0, 0, 0, 0)
        self._trace = None
        self.set_weights(state_dict)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.dsbdd_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_weights(self, state_dict):
        self.slots = pack_weights(state_dict, self.cfg, self.device)
        n = self.lib.dsbdd_engine_weight_slots(self.handle)
        assert n == len(self.slots), (n, len(self.slots))
        arr = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in self.slots])
        _lib.check(self.lib.dsbdd_engine_set_weights(self.handle, arr, n), "dsbdd_engine_set_weights")

    def ensure_workspace(self, n_lig, n_pocket, batch, edge_cap):
        c = self.caps
        if n_lig <= c[0] and n_pocket <= c[1] and batch <= c[2] and edge_cap <= c[3]:
            return
        caps = (max(n_lig, c[0]), max(n_pocket, c[1]), max(batch, c[2]), max(edge_cap, c[3], 1))
        nbytes = self.lib.dsbdd_engine_workspace_bytes(self.handle, *caps)
        if nbytes == 0:
            raise _lib.HipLibraryError("dsbdd_engine_workspace_bytes rejected the sizes %r" % (caps,))
        # the old workspace may still be in use by calls enqueued on this stream (a device-wide
        # synchronize is not allowed while another thread captures a graph: streams.py)
        torch.cuda.current_stream(self.device).synchronize()
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.workspace.data_ptr() + 255) & ~255
        _lib.check(self.lib.dsbdd_engine_bind_workspace(self.handle, base, nbytes, *caps),
                   "dsbdd_engine_bind_workspace")
        self.caps = caps

    def set_pocket_frame(self, x_pocket, mask_pocket, sizes_pocket, n_lig, batch, edge_cap, shared=False,
                         representative=None):
        """dsbdd_engine_set_pocket_frame: raw pocket coordinates x_pocket [n_pocket, 3] (fp32, device) of a
        pocket-conditioned chain.  `representative` [batch]: for every sample the index of the sample whose pocket
        stands for it (itself, or an earlier sample with an IDENTICAL pocket); shared=True is the common special case
        "every sample has sample 0's pocket", the default is "every sample its own".  The frame problem is the
        representatives' pockets; all samples of a group read its block-0 pocket-pocket messages and, in the forward
        cone, its canonical pocket network."""
        dev = self.device
        n_pocket = x_pocket.shape[0]
        self.ensure_workspace(n_lig, n_pocket, batch, edge_cap)
        x_pocket = x_pocket.to(device=dev, dtype=torch.float32).contiguous()
        sz = sizes_pocket.to(device=dev, dtype=torch.int64)
        if representative is None:
            representative = torch.zeros(batch, dtype=torch.int64) if shared else torch.arange(batch)
        rep = torch.as_tensor(representative, dtype=torch.int64).to(dev)
        mask3, frame_rows, twin, sz_f = frame_layout(sz, rep, mask_pocket)
        n3, b3 = int(frame_rows.numel()), int(sz_f.numel())
        x_frame = x_pocket[frame_rows.long()].contiguous()
        bound = int((((sz_f * sz_f) + 31) // 32 * 32).sum().item()) + 32
        self._frame_keep = (x_frame, twin, mask3, frame_rows)
        _lib.check(self.lib.dsbdd_engine_set_pocket_frame(
            self.handle, torch.cuda.current_stream(dev).cuda_stream, x_frame.data_ptr(), mask3.data_ptr(),
            frame_rows.data_ptr(), twin.data_ptr(), n_lig, n_pocket, batch, n3, b3, min(bound, self.caps[3])),
            "dsbdd_engine_set_pocket_frame")

    def clear_pocket_frame(self):
        _lib.check(self.lib.dsbdd_engine_clear_pocket_frame(self.handle), "dsbdd_engine_clear_pocket_frame")
        self._frame_keep = None

    def set_trace(self, n_nodes):
        """Allocate per-block trace buffers (debug / parity tests)."""
        L, H = self.cfg.n_layers, self.cfg.hidden_nf
        th = torch.zeros((L, n_nodes, H), dtype=torch.float32, device=self.device)
        tx = torch.zeros((L, n_nodes, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dsbdd_engine_set_trace(self.handle, th.data_ptr(), tx.data_ptr()))
        self._trace = (th, tx)
        return th, tx

    def clear_trace(self):
        _lib.check(self.lib.dsbdd_engine_set_trace(self.handle, None, None))
        self._trace = None

    def forward_async(self, xh_lig, xh_pocket, t, mask_lig, mask_pocket, batch, edge_cap,
                      ext_edges=None, eps_lig=None, eps_pocket=None, status=None,
                      want_pocket=True):
        """Enqueue EGNNDynamics.forward on the current stream; no host sync.
        Inputs must be contiguous fp32 / int64 tensors on this engine's GPU."""
        dev = self.device
        for name, x, dt in (("xh_lig", xh_lig, torch.float32), ("xh_pocket", xh_pocket, torch.float32),
                            ("t", t, torch.float32), ("mask_lig", mask_lig, torch.int64),
                            ("mask_pocket", mask_pocket, torch.int64)):
            if x.device != dev or x.dtype != dt or not x.is_contiguous():
                raise ValueError(f"{name}: expected contiguous {dt} tensor on {dev}, got "
                                 f"{x.dtype} on {x.device} (contiguous={x.is_contiguous()})")
        n_lig, n_pocket = xh_lig.shape[0], xh_pocket.shape[0]
        if ext_edges is not None:
            er = ext_edges[0].to(device=dev, dtype=torch.int32).contiguous()
            ec = ext_edges[1].to(device=dev, dtype=torch.int32).contiguous()
            edge_cap = max(edge_cap, er.numel())
        self.ensure_workspace(n_lig, n_pocket, batch, edge_cap)
        if eps_lig is None:
            eps_lig = torch.empty_like(xh_lig)
        if eps_pocket is None and want_pocket:
            eps_pocket = torch.empty_like(xh_pocket)
        if status is None:
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.dsbdd_dynamics_forward(
            self.handle, stream, xh_lig.data_ptr(), xh_pocket.data_ptr(), t.data_ptr(), t.numel(),
            mask_lig.data_ptr(), mask_pocket.data_ptr(), n_lig, n_pocket, batch,
            er.data_ptr() if ext_edges is not None else None,
            ec.data_ptr() if ext_edges is not None else None,
            er.numel() if ext_edges is not None else 0,
            eps_lig.data_ptr(), eps_pocket.data_ptr() if eps_pocket is not None else None,
            status.data_ptr())
        _lib.check(rc, "dsbdd_dynamics_forward")
        return eps_lig, eps_pocket, status

    def profile(self, enable, max_launches=16384):
        """enable: 0/False off, k > 0: time the GCL launches of every k-th forward call."""
        _lib.check(self.lib.dsbdd_engine_profile(self.handle, int(enable), int(max_launches)))

    def profile_read(self):
        """(total kernel ms, launches) of the timed GCL edge kernel since the last read."""
        ms, n = C.c_double(0), C.c_int64(0)
        _lib.check(self.lib.dsbdd_engine_profile_read(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def graph_stats(self):
        """(graph replays, captures, eager calls) of dsbdd_dynamics_forward so far."""
        r, c, g = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.dsbdd_engine_graph_stats(self.handle, C.byref(r), C.byref(c), C.byref(g)))
        return r.value, c.value, g.value

    def edge_count(self, n_nodes):
        """Number of edges of the last forward (syncs): the sum of the row degrees.  The edge
        list itself is longer: every (sample, node set) segment is padded to a multiple of 32."""
        import numpy as np
        torch.cuda.synchronize(self.device)
        return int(self._read(self.buffer_ptr(_lib.BUF_DEG), n_nodes, np.int32).astype(np.int64).sum())

    def edge_slots(self, n_nodes):
        """Length of the last forward's edge list including the padding entries (syncs)."""
        import numpy as np
        torch.cuda.synchronize(self.device)
        return int(self._read(self.buffer_ptr(_lib.BUF_ROW_PTR) + 4 * n_nodes, 1, np.int32)[0])

    def buffer_ptr(self, which):
        p = C.c_void_p()
        _lib.check(self.lib.dsbdd_engine_buffer(self.handle, which, C.byref(p)))
        return p.value

    def last_edges(self, n_nodes):
        """(row, col) of the last forward's edge list, as int64 CPU tensors
        (debug / tests; syncs)."""
        import numpy as np
        torch.cuda.synchronize(self.device)
        rp = self._read(self.buffer_ptr(_lib.BUF_ROW_PTR), n_nodes + 1, np.int32)
        E = int(rp[-1])
        row = self._read(self.buffer_ptr(_lib.BUF_EDGE_ROW), E, np.int32)
        col = self._read(self.buffer_ptr(_lib.BUF_EDGE_COL), E, np.int32)
        keep = row >= 0                                   # padding entries carry row = -1
        return torch.from_numpy(row[keep].astype("int64")), torch.from_numpy(col[keep].astype("int64"))

    def last_levels(self, n_nodes):
        """Level structures of the last ligand-output-only call in pocket-conditioning mode (csrc/graph.h,
        "Level-ordered edge list"), as numpy arrays (debug / tests; syncs): dict with level[N], order[N] (nodes by
        (level, id)), count[5] (nodes with level <= r), end[5] (list position where the rows of level <= r end),
        row_ptr[N], deg[N], the re-ordered list row / col / d0 (padding entries have row = -1), and the number of
        ghost nodes / list slots of a canonical pocket in front of them (forward cone)."""
        import numpy as np
        torch.cuda.synchronize(self.device)
        rd = lambda which, n, dt: self._read(self.buffer_ptr(which), n, dt)
        cnt, end = rd(_lib.BUF_LEVEL_COUNT, 10, np.int32), rd(_lib.BUF_LEVEL_END, 10, np.int32)
        ng, gs = int(cnt[4] - cnt[9]), int(end[4] - end[9])
        out = {"level": rd(_lib.BUF_LEVEL, n_nodes, np.int32), "order": rd(_lib.BUF_LEVEL_LIST, ng + n_nodes, np.int32)[ng:],
               "count": cnt[5:], "end": end[:5], "ghost_nodes": ng, "ghost_slots": gs,
               "row_ptr": rd(_lib.BUF_LROW_PTR, n_nodes, np.int32), "deg": rd(_lib.BUF_DEG, n_nodes, np.int32)}
        E = int(end[4])
        out["row"] = rd(_lib.BUF_LEDGE_ROW, E, np.int32)
        out["col"] = rd(_lib.BUF_LEDGE_COL, E, np.int32)
        out["d0"] = rd(_lib.BUF_LEDGE_D0, E, np.float32)
        return out

    def level_stats(self, raw=False, since=None):
        """Running sums over the ligand-output-only calls since the workspace was bound (syncs): per call, the mean
        number of nodes with level <= r, list slots and edges of those rows (r = 0..4), and the number of calls.
        `raw`: the 16 sums themselves (to pass back later as `since`: statistics of the calls in between).
        None when there were no such calls."""
        import numpy as np
        if self.workspace is None:
            return np.zeros(16) if raw else None
        torch.cuda.synchronize(self.device)
        st = self._read(self.buffer_ptr(_lib.BUF_LEVEL_STATS), 16, np.uint64).astype(np.float64)
        if raw:
            return st
        if since is not None:
            st = st - since
        if st[15] <= 0:
            return None
        return {"nodes": (st[:5] / st[15]).tolist(), "list_slots": (st[5:10] / st[15]).tolist(),
                "edges": (st[10:15] / st[15]).tolist(), "calls": int(st[15])}

    def last_plan(self):
        """(radius per message stage, ghost flag per stage, level of the timed launches) of the last forward."""
        r = (C.c_int32 * 64)()
        g = (C.c_int32 * 64)()
        n, tl = C.c_int32(), C.c_int32()
        _lib.check(self.lib.dsbdd_engine_last_plan(self.handle, r, g, 64, C.byref(n), C.byref(tl)))
        return list(r[:n.value]), list(g[:n.value]), tl.value

    def set_option(self, which, value):
        """Engine switches (include/diffsbdd_hip.h DSBDD_OPT_*).  Setting a value again is free; a change drops the
        captured graphs."""
        if not hasattr(self, "_options"):
            self._options = {}
        if self._options.get(which) == int(value):
            return
        _lib.check(self.lib.dsbdd_engine_set_option(self.handle, which, int(value)))
        self._options[which] = int(value)

    def get_option(self, which):
        """Current value of an engine switch (what the environment / set_option left)."""
        return int(self.lib.dsbdd_engine_get_option(self.handle, which))

    def _read(self, ptr, count, dtype):
        import numpy as np
        itemsize = np.dtype(dtype).itemsize
        base = (self.workspace.data_ptr() + 255) & ~255
        off = ptr - self.workspace.data_ptr()
        raw = self.workspace[off:off + count * itemsize].cpu().numpy()
        return raw.view(dtype).copy()
