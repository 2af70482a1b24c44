"""REJECTED EXPERIMENT (round 2; kept under tools/ for the record, not part of the product package).

Intra-GPU replicas: one sampling batch as several concurrent sub-batches.

A launch needs far more than 256 workgroups to fill an MI355X (256 CUs in 8 XCDs).
The C-alpha workloads (BASELINE.json configs[1]: 32 pockets, ~20 k edges per EGNN
call = 157 edge tiles) cannot do that: every launch of a call is one tile's latency
long and most CUs idle.  Because samples are independent chains and every per-sample
result of the HIP path is bitwise independent of the batch composition (aligned edge
segments + fixed summation order, csrc/edge_mlp.h; noise keyed by the global sample
index), a batch can be cut into S contiguous sub-batches that run on S HIP streams
-- each with its own engine (workspace, captured graph) but the same parameter
tensors -- and the concatenated result is identical, bit for bit, to the single-batch
run (verified in round 2).

MEASURED (profiles/README.md, round 2): it does NOT pay inside one process.  A
sub-batch call is as long as the full-batch call (latency-bound), and replaying the
~70-node captured graph costs ~0.57 ms of host time per call under the runtime's
process-wide submission lock, so S streams become host-bound: 48.8 / 42.1 / 23.9 /
13.9 ligands/s for S = 1 / 2 / 4 / 8 on the C-alpha workload.  What the latency regime
needs is fewer, larger launches per call, not more submitters.

The reference has nothing comparable (one batch, one stream:
/root/reference/lightning_modules.py:797-852).
"""
from __future__ import annotations

import copy
import threading

import torch

__all__ = ["StreamReplicas", "auto_streams"]


def auto_streams(n_nodes_total: int, batch: int, max_streams: int = 4) -> int:
    """Number of concurrent sub-batches to use by default: 1 (see the module docstring: splitting
    was measured slower for small and for large batches)."""
    return 1


def _replica(ddpm):
    """A second handle on the same model: shared parameter tensors, private engine / chain state."""
    rep = copy.copy(ddpm)
    rep._modules = dict(ddpm._modules)
    dyn = copy.copy(ddpm.dynamics)
    dyn._engine = None
    rep._modules["dynamics"] = dyn
    rep._dyn_bufs = {}
    rep._chain = None
    rep._coef_cache = {}
    return rep


def _slice_batch(d, lo, hi):
    """Samples [lo, hi) of a {'x','one_hot','size','mask'} dict with a sorted mask."""
    mask = d["mask"]
    bounds = torch.searchsorted(mask.contiguous(), torch.tensor([lo, hi], device=mask.device, dtype=mask.dtype))
    r0, r1 = int(bounds[0]), int(bounds[1])
    return {"x": d["x"][r0:r1].clone(), "one_hot": d["one_hot"][r0:r1].clone(), "size": d["size"][lo:hi].clone(),
            "mask": (mask[r0:r1] - lo).clone()}


class StreamReplicas:
    """`ddpm` (ConditionalDDPM / EnVariationalDiffusion on a GPU) driven as `n_streams`
    concurrent replicas.  `sample_given_pocket` has the model method's signature and result."""

    def __init__(self, ddpm, n_streams: int):
        assert n_streams >= 1
        self.ddpm = ddpm
        self.device = next(ddpm.dynamics.parameters()).device
        self.replicas = [ddpm] + [_replica(ddpm) for _ in range(n_streams - 1)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n_streams)]

    def _run(self, fn_name, parts, seed, sample_offset, draw0):
        """parts: list of (lo, args, kwargs) per replica; all replicas start at draw index draw0."""
        out, err = [None] * len(parts), [None] * len(parts)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))

        def work(i):
            try:
                lo, args, kwargs = parts[i]
                with torch.cuda.stream(self.streams[i]):
                    self.streams[i].wait_event(ready)              # inputs were produced on the caller's stream
                    rep = self.replicas[i]
                    rep.noise_source = self.ddpm.noise_source
                    rep.seed(seed, sample_offset=sample_offset + lo)   # (explicit id tables are not split here)
                    rep._draw = draw0
                    out[i] = getattr(rep, fn_name)(*args, **kwargs)
            except BaseException as exc:   # re-raised in the caller's thread
                err[i] = exc

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for s in self.streams[:len(parts)]:
            torch.cuda.current_stream(self.device).wait_stream(s)
        for e in err:
            if e is not None:
                raise e
        # replica 0 is the model itself: its draw counter has advanced like in an unsplit call
        self.ddpm._sample_offset = sample_offset
        return out

    @torch.no_grad()
    def sample_given_pocket(self, pocket, num_nodes_lig, return_frames=1, timesteps=None, seed=None,
                            sample_offset=0):
        """ConditionalDDPM.sample_given_pocket on contiguous sub-batches, one per stream.
        `seed` / `sample_offset` key the noise by the GLOBAL sample index (as model.seed does), which
        is what makes the result independent of the split."""
        n = len(pocket["size"])
        S = min(len(self.replicas), n)
        draw0 = 0
        if seed is None:       # continue the model's own generator (seed(), or torch's RNG on first use)
            if self.ddpm._seed is None:
                self.ddpm._seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            seed, sample_offset, draw0 = self.ddpm._seed, self.ddpm._sample_offset, self.ddpm._draw
        pocket = {k: v.to(self.device) for k, v in pocket.items()}
        pocket["mask"] = pocket["mask"].to(torch.int64)
        per = (n + S - 1) // S
        parts = []
        for i in range(S):
            lo, hi = min(i * per, n), min((i + 1) * per, n)
            if hi > lo:
                nl = num_nodes_lig if isinstance(num_nodes_lig, int) else num_nodes_lig[lo:hi]
                parts.append((lo, (_slice_batch(pocket, lo, hi), nl), dict(return_frames=return_frames,
                                                                         timesteps=timesteps)))
        res = self._run("sample_given_pocket", parts, seed, sample_offset, draw0)
        cat = 0 if return_frames == 1 else 1
        out_l = torch.cat([r[0] for r in res], dim=cat)
        out_p = torch.cat([r[1] for r in res], dim=cat)
        lm = torch.cat([r[2] + parts[i][0] for i, r in enumerate(res)])
        pm = torch.cat([r[3] + parts[i][0] for i, r in enumerate(res)])
        return out_l, out_p, lm, pm
