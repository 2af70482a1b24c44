"""Error study for the fp32-emulated matrix path -- CPU model, numpy only.

The second layer of every edge MLP is  z2 = a1 @ W2^T + b2  with a1 = SiLU(first layer) (egnn_new.py:15-19,80-92),
K = H = 256.  The exact path evaluates it as one fp32 fmaf chain per output (v_mfma_f32_32x32x2_f32).  The emulated
path splits both operands into three bf16 terms (x = hi + mid + lo exactly: 3 x 8 significand bits = 24) and adds the
partial products on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulators); a bf16 x bf16 product is exact
in fp32.  "6 products" drops a_mid*b_lo, a_lo*b_mid, a_lo*b_lo (each <= 2^-24 |a||b|); "9 products" keeps all.

Reference: float64 evaluation of the same fp32 inputs.  Reported: max and rms error relative to max |z2|, for
  fp32 chain      sequential fp32 fma over k (what the exact kernel computes, up to the order of k)
  emu-6 / emu-9   per 16-k step the partial products summed exactly, then ONE fp32 add to the accumulator
                  (matrix core with a wide internal adder), and the pessimistic variant with an fp32 rounding after
                  every product (seq).
The GPU measurement of the same quantity on the real kernels is tools/microbench_emu.hip.
"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)           # exact
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)         # exact
    lo = bf16_rne(r2)                          # exact (<= 8 significant bits left) unless it underflows
    return hi, mid, lo


def silu(x):
    return x / (1.0 + np.exp(-x))


def run(seed, E=512, H=256, scale_w=1.0):
    g = np.random.default_rng(seed)
    # first-layer pre-activations of a trained-ish network: P + Q + radial terms, O(1)
    pre = (g.standard_normal((E, H)) * 1.5).astype(np.float32)
    a = silu(pre.astype(np.float64)).astype(np.float32)
    W = (g.uniform(-1, 1, (H, H)) / np.sqrt(H) * scale_w).astype(np.float32)   # nn.Linear default init range
    b2 = (g.uniform(-1, 1, H) / np.sqrt(H)).astype(np.float32)
    ref = a.astype(np.float64) @ W.astype(np.float64).T + b2
    out = {}
    # exact fp32 chain
    acc = np.broadcast_to(b2, (E, H)).astype(np.float32).copy()
    for k in range(H):
        acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * W[None, :, k].astype(np.float64)).astype(np.float32)
    out["fp32 chain"] = acc        # (fma: one rounding per step)
    ah, am, al = split3(a)
    wh, wm, wl = split3(W)
    assert np.array_equal((ah.astype(np.float64) + am + al), a.astype(np.float64))
    assert np.array_equal((wh.astype(np.float64) + wm + wl), W.astype(np.float64))
    prods6 = [(al, wh), (am, wm), (ah, wl), (am, wh), (ah, wm), (ah, wh)]
    prods9 = [(al, wl), (al, wm), (am, wl)] + prods6
    for name, prods in (("emu-6", prods6), ("emu-9", prods9)):
        acc = np.broadcast_to(b2, (E, H)).astype(np.float32).copy()
        accs = acc.copy()
        for k0 in range(0, H, 16):
            for (x, y) in prods:
                part = x[:, k0:k0 + 16].astype(np.float64) @ y[:, k0:k0 + 16].astype(np.float64).T
                acc = (acc.astype(np.float64) + part).astype(np.float32)       # one MFMA: exact 16-term sum, one rounding
                for k in range(k0, k0 + 16):                                   # pessimistic: a rounding per product
                    accs = (accs.astype(np.float64) + x[:, k:k + 1].astype(np.float64) * y[None, :, k].astype(np.float64)).astype(np.float32)
        out[name] = acc
        out[name + " (seq)"] = accs
    # plain bf16 and 2-way split for scale (these earn nothing; shown to place the 3-way split)
    out["bf16 x1 (not used)"] = (ah.astype(np.float64) @ wh.astype(np.float64).T + b2).astype(np.float32)
    two = sum(x.astype(np.float64) @ y.astype(np.float64).T for x, y in ((ah, wh), (ah, wm), (am, wh))) + b2
    out["bf16 x2, 3 products (not used)"] = two.astype(np.float32)
    m = np.abs(ref).max()
    rows = []
    for k, v in out.items():
        err = np.abs(v.astype(np.float64) - ref)
        rows.append((k, err.max() / m, np.sqrt((err ** 2).mean()) / m))
    return rows


if __name__ == "__main__":
    print("| variant | max err / max|z2| | rms err / max|z2| | max vs fp32 chain |")
    print("|---|---|---|---|")
    agg = {}
    for seed in range(3):
        for k, mx, rms in run(seed):
            agg.setdefault(k, []).append((mx, rms))
    base = max(v[0] for v in agg["fp32 chain"])
    for k, v in agg.items():
        mx = max(x[0] for x in v); rms = np.mean([x[1] for x in v])
        print(f"| {k} | {mx:.3e} | {rms:.3e} | {mx / base:.2f}x |")
