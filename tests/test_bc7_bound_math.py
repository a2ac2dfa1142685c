"""The mathematics behind the sharper second-tier bound of the BC7 kernel (bc7_kernel.hip: subsetBoundSharp; DESIGN.md 4.1,
"Soundness of the bounds", part 3), restated in numpy and checked against encodings found by search: for random subsets of
smooth, photo-like and noisy blocks, NO pair of 8-bit end points with 2-bit or 3-bit BC7 interpolation (any mode's end points
are a subset of those) may reach an error below the bound -- neither with the subset's principal direction nor with a
perturbed one (the bound must hold for any unit vector).  CPU only; the kernel's own arithmetic is covered by the bit-exact
parity tests on the GPU (a bound that is too high there shows up as a block that differs from the reference)."""
import numpy as np

from convectionkernels_amd import synth

W = np.array([np.float32(0.2125) / np.float32(0.7154), 1.0, np.float32(0.0721) / np.float32(0.7154)], np.float64)  # default channel weights
W2 = float((W ** 2).sum())
DELTA = 0.5 * np.sqrt(W2)
BETAS = [0, 0.01, 0.02, 0.03, 0.04, 0.055, 0.07, 0.085, 0.1, 0.125, 0.15, 0.175, 0.2, 0.25, 0.3, 0.375, 0.45, 0.55, 0.7, 0.85, 1.0]
WEIGHTS = {4: [0, 21, 43, 64], 8: [0, 9, 18, 27, 37, 46, 55, 64]}
PARTITIONS2 = [0xcccc, 0x8888, 0xeeee, 0xecc8, 0xc880, 0xfeec, 0xfec8, 0xec80, 0xc800, 0xffec, 0xfe80, 0xe800, 0xffe8, 0xff00, 0xfff0, 0xf000,
               0xf710, 0x008e, 0x7100, 0x08ce, 0x008c, 0x7310, 0x3100, 0x8cce, 0x088c, 0x3110, 0x6666, 0x366c, 0x17e8, 0x0ff0, 0x718e, 0x399c]


def slack(x, s, n):
    """error >= x - 2 s sqrt(n x) when every reconstructed colour is within s of the trial's line (clamped, monotone in x)"""
    return x - 2 * s * np.sqrt(n * x) if x > 4 * n * s * s else 0.0


def kmeans_1d(t, k):
    """exact cost of the best clustering of t into k groups (the kernel's dynamic programme)"""
    t = np.sort(t)
    n = len(t)
    if n <= k:
        return 0.0
    p1 = np.concatenate([[0.0], np.cumsum(t)])
    p2 = np.concatenate([[0.0], np.cumsum(t * t)])
    cost = lambda i, j: (p2[j] - p2[i]) - (p1[j] - p1[i]) ** 2 / (j - i)
    d = [cost(0, j) if j else 0.0 for j in range(n + 1)]
    for _ in range(k - 1):
        d = [min([d[j]] + [d[i] + cost(i, j) for i in range(j)]) for j in range(n + 1)]
    return max(d[n], 0.0)


def sharp_bound(points_w, d, levels):
    """subsetBoundSharp without its float margins: points_w = weighted pixels (n, 3), d = any unit vector"""
    n = len(points_w)
    c = points_w - points_w.mean(0)
    S = c.T @ c
    T = np.trace(S)
    r_true = T - np.linalg.eigvalsh(S)[-1]
    L = d @ S @ d
    Rd = T - L
    rho = np.linalg.norm(S @ d - L * d)
    a = np.abs(d) * W
    md = max(0.0, 2 * a.max() - a.sum())
    perp = np.sqrt(max(W2 - md * md, 0.0))
    sq = np.sqrt(kmeans_1d(c @ d, levels)) if levels == 4 else 0.0  # the kernel computes Q_d for the two-bit modes only
    best = None
    for b0, b1 in zip(BETAS[:-1], BETAS[1:]):
        a1 = np.sqrt(1 - b1 * b1)
        cross = 0.5 if b0 <= np.sqrt(0.5) <= b1 else max(b0 * np.sqrt(1 - b0 * b0), b1 * a1)
        base = min((1 - b0 * b0) * Rd + b0 * b0 * L, (1 - b1 * b1) * Rd + b1 * b1 * L) - 2 * cross * rho
        base = max(base, r_true)
        g = base + max(0.0, a1 * sq - b1 * np.sqrt(max(Rd, 0.0))) ** 2
        inner = max(0.0, a1 * md - b1 * perp)
        v = slack(g, 0.5 * np.sqrt(max(W2 - inner * inner, 0.0)), n)
        best = v if best is None else min(best, v)
    return max(slack(r_true, DELTA, n), best)


def error_of(q, e0, e1, levels):
    wt = np.array(WEIGHTS[levels])
    rec = ((e0[None, :] * (64 - wt[:, None]) + e1[None, :] * wt[:, None] + 32) >> 6).astype(np.float64)
    return ((((q[:, None, :] - rec[None, :, :]) * W) ** 2).sum(-1)).min(1).sum()


def search_encoding(q, levels):
    """a good (not necessarily optimal) encoding of the subset: PCA end points, then coordinate descent on the integers"""
    p = q * W
    c = p - p.mean(0)
    d = np.linalg.eigh(c.T @ c)[1][:, -1]
    t = c @ d
    best = None
    for sc in (1.0, 0.85, 1.1):
        e0 = np.clip(np.rint(q.mean(0) + (d / W) * t.min() * sc), 0, 255).astype(np.int64)
        e1 = np.clip(np.rint(q.mean(0) + (d / W) * t.max() * sc), 0, 255).astype(np.int64)
        cur = error_of(q, e0, e1, levels)
        improved = True
        while improved:
            improved = False
            for which in (0, 1):
                for ch in range(3):
                    for dv in (-2, -1, 1, 2):
                        f0, f1 = e0.copy(), e1.copy()
                        tgt = f0 if which == 0 else f1
                        tgt[ch] = np.clip(tgt[ch] + dv, 0, 255)
                        e = error_of(q, f0, f1, levels)
                        if e < cur:
                            cur, e0, e1, improved = e, f0, f1, True
        best = cur if best is None else min(best, cur)
    return best


def test_sharp_bound_never_exceeds_a_found_encoding():
    rng = np.random.default_rng(11)
    fam = synth.content_families(512)
    worst = 0.0
    for name in ("photo-like", "gradient opaque", "opaque noise"):
        for _ in range(40):
            blk = fam[name][rng.integers(512)][:, :3].astype(np.int64)
            part = PARTITIONS2[rng.integers(len(PARTITIONS2))]
            member = np.array([((part >> k) & 1) == rng.integers(2) for k in range(16)])
            q = blk[member]
            if len(q) < 3:
                continue
            levels = 4 if rng.integers(2) else 8
            p = q * W
            c = p - p.mean(0)
            d = np.linalg.eigh(c.T @ c)[1][:, -1]
            d2 = d + rng.normal(0, 0.05, 3)
            d2 /= np.linalg.norm(d2)
            found = search_encoding(q, levels)
            for dd in (d, d2):
                bound = sharp_bound(p, dd, levels)
                assert bound <= found + 1e-9, (name, levels, len(q), bound, found)
                worst = max(worst, bound / max(found, 1e-9))
    assert worst > 0.5  # ... and it is not vacuous: somewhere it comes within a factor of two of what the search found


def test_clustering_cost_is_exact_on_small_cases():
    rng = np.random.default_rng(3)
    for _ in range(30):
        t = rng.integers(0, 40, rng.integers(5, 9)).astype(np.float64)
        n = len(t)
        best = None
        # brute force over all assignments of the sorted values to 4 runs
        ts = np.sort(t)
        for a in range(1, n):
            for b in range(a, n):
                for c in range(b, n):
                    cost = sum(((g - g.mean()) ** 2).sum() for g in (ts[:a], ts[a:b], ts[b:c], ts[c:]) if len(g))
                    best = cost if best is None else min(best, cost)
        assert abs(kmeans_1d(t, 4) - best) < 1e-9
