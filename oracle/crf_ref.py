"""TEST INFRASTRUCTURE ONLY - CPU restatement of the fully connected CRF the reference runs on every candidate map
(tools/seg_evaluation.py:31-54: `pydensecrf.densecrf.DenseCRF2D`, 2 labels, `addPairwiseGaussian(sxy=3, compat=3)`,
`addPairwiseBilateral(sxy=40, srgb=13, compat=10)`, `inference(3)`, argmax).

`pydensecrf` (requirements.txt, a Cython wrapper of Kraehenbuehl & Koltun's `densecrf` library, NIPS 2011 "Efficient Inference in
Fully Connected CRFs with Gaussian Edge Potentials") is NOT installed here and not vendored under /root/reference, and the reference
has no test that pins its output: PARITY WITH THE LIBRARY ITSELF IS UNPINNED.  What this file restates, function by function:

  * `DenseCRF::inference`  (densecrf.cpp): Q = softmax(-U); per iteration  tmp = -U - sum_k pairwise_k(Q);  Q = softmax(tmp)
  * `PairwisePotential::apply` + `PottsCompatibility::apply` (pairwise.cpp, labelcompatibility.cpp):  pairwise(Q) = -w * filter(Q)
  * `DenseKernel` with NORMALIZE_SYMMETRIC (the default of addPairwise*): norm = 1 / sqrt(K 1 + 1e-20);  filter(Q) = norm * K(norm * Q)
  * `Permutohedral::init / compute` (permutohedral.cpp; Adams, Baek & Davis, "Fast High-Dimensional Filtering Using the
    Permutohedral Lattice", 2010): elevation onto the hyperplane, rounding to the closest remainder-0 point, rank, barycentric
    weights, the d+1 simplex vertices, blur along the d+1 lattice directions (new = old + (n1 + n2) / 2), slice with
    alpha = 1 / (1 + 2^-d).  (Constant factors of K cancel in the symmetric normalisation.)
  * the 2-D feature layouts of DenseCRF2D (densecrf.cpp): (x / sxy, y / sxy) and (x / sxy, y / sxy, r / srgb, g / srgb, b / srgb).

`mean_field_exact` is the same mean-field with the EXACT Gaussian kernels the lattice approximates (O(N^2), small images only):
tests pin the lattice restatement against it (label agreement, |Q| difference) - the published algorithm is the anchor.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------------------- lattice
class Permutohedral:
    """Permutohedral::init (feature [N, d] float32)."""

    def __init__(self, feature):
        f = np.ascontiguousarray(feature, dtype=np.float32)
        N, d = f.shape
        self.N, self.d = N, d
        inv_std_dev = np.float32(np.sqrt(2.0 / 3.0) * (d + 1))
        scale = np.array([1.0 / np.sqrt((i + 1) * (i + 2)) for i in range(d)], dtype=np.float32) * inv_std_dev
        # elevate: elevated[j] = sm - j * cf, sm += cf (j = d .. 1), elevated[0] = sm
        cf = f * scale[None, :]                                      # [N, d]
        elevated = np.zeros((N, d + 1), dtype=np.float32)
        sm = np.zeros(N, dtype=np.float32)
        for j in range(d, 0, -1):
            elevated[:, j] = sm - np.float32(j) * cf[:, j - 1]
            sm = sm + cf[:, j - 1]
        elevated[:, 0] = sm
        # closest remainder-0 point
        down_factor, up_factor = np.float32(1.0 / (d + 1)), np.float32(d + 1)
        v = down_factor * elevated
        up, down = np.ceil(v) * up_factor, np.floor(v) * up_factor
        rem0 = np.where(up - elevated < elevated - down, up, down).astype(np.float32)
        ssum = np.rint((rem0 * down_factor).sum(1)).astype(np.int64)           # integer by construction
        # rank of the differential
        diff = elevated - rem0
        rank = np.zeros((N, d + 1), dtype=np.int64)
        for i in range(d):
            for j in range(i + 1, d + 1):
                lt = diff[:, i] < diff[:, j]
                rank[:, i] += lt
                rank[:, j] += ~lt
        # points off the hyperplane: wrap
        rank = rank + ssum[:, None]
        lo, hi = rank < 0, rank > d
        rank = np.where(lo, rank + d + 1, np.where(hi, rank - (d + 1), rank))
        rem0 = np.where(lo, rem0 + up_factor, np.where(hi, rem0 - up_factor, rem0))
        # barycentric coordinates
        bary = np.zeros((N, d + 2), dtype=np.float32)
        vdf = (elevated - rem0) * down_factor
        rows = np.arange(N)
        for i in range(d + 1):
            np.add.at(bary, (rows, d - rank[:, i]), vdf[:, i])
            np.add.at(bary, (rows, d + 1 - rank[:, i]), -vdf[:, i])
        bary[:, 0] += 1.0 + bary[:, d + 1]
        self.barycentric = bary[:, :d + 1].copy()
        # simplex vertices: key[i] = rem0[i] + canonical[remainder][rank[i]], canonical[r][k] = r for k <= d - r, r - (d + 1) above
        canonical = np.zeros((d + 1, d + 1), dtype=np.int64)
        for r in range(d + 1):
            canonical[r, :d + 1 - r] = r
            canonical[r, d + 1 - r:] = r - (d + 1)
        rem0i = np.rint(rem0).astype(np.int64)
        keys = np.empty((N, d + 1, d), dtype=np.int64)                # first d coordinates identify a lattice point
        for r in range(d + 1):
            keys[:, r, :] = rem0i[:, :d] + canonical[r][rank[:, :d]]
        uniq, inv = np.unique(keys.reshape(-1, d), axis=0, return_inverse=True)
        self.keys = uniq
        self.M = uniq.shape[0]
        self.offset = inv.reshape(N, d + 1)
        # blur neighbours along each of the d + 1 directions
        table = {tuple(k): i for i, k in enumerate(uniq.tolist())}
        self.n1 = np.full((d + 1, self.M), -1, dtype=np.int64)
        self.n2 = np.full((d + 1, self.M), -1, dtype=np.int64)
        for j in range(d + 1):
            k1, k2 = uniq - 1, uniq + 1
            if j < d:
                k1, k2 = k1.copy(), k2.copy()
                k1[:, j] = uniq[:, j] + d
                k2[:, j] = uniq[:, j] - d
            self.n1[j] = [table.get(tuple(k), -1) for k in k1.tolist()]
            self.n2[j] = [table.get(tuple(k), -1) for k in k2.tolist()]

    def compute(self, values):
        """Permutohedral::seqCompute, forward order.  values [N, C] -> [N, C]."""
        x = np.ascontiguousarray(values, dtype=np.float32)
        C = x.shape[1]
        lat = np.zeros((self.M + 1, C), dtype=np.float32)             # last row: the "missing neighbour" zero
        for r in range(self.d + 1):
            np.add.at(lat, self.offset[:, r], self.barycentric[:, r:r + 1] * x)
        lat[self.M] = 0
        for j in range(self.d + 1):
            lat = lat + np.float32(0.5) * (lat[self.n1[j].tolist() + [self.M]] + lat[self.n2[j].tolist() + [self.M]])
            lat[self.M] = 0
        alpha = np.float32(1.0 / (1.0 + 2.0 ** (-self.d)))
        out = np.zeros((self.N, C), dtype=np.float32)
        for r in range(self.d + 1):
            out += self.barycentric[:, r:r + 1] * lat[self.offset[:, r]] * alpha
        return out


class LatticeKernel:
    """DenseKernel(feature, DIAG_KERNEL, NORMALIZE_SYMMETRIC)."""

    def __init__(self, feature):
        self.lattice = Permutohedral(feature)
        k1 = self.lattice.compute(np.ones((feature.shape[0], 1), dtype=np.float32))[:, 0]
        self.norm = (1.0 / np.sqrt(k1 + 1e-20)).astype(np.float32)

    def apply(self, Q):                                               # Q [N, C]
        return self.norm[:, None] * self.lattice.compute(self.norm[:, None] * Q)


class ExactKernel:
    """The kernel the lattice approximates: K_ij = exp(-|f_i - f_j|^2 / 2) (self term included, as the lattice does)."""

    def __init__(self, feature):
        f = feature.astype(np.float64)
        d2 = ((f[:, None, :] - f[None, :, :]) ** 2).sum(-1)
        self.K = np.exp(-0.5 * d2)
        self.norm = 1.0 / np.sqrt(self.K.sum(1) + 1e-20)

    def apply(self, Q):
        return (self.norm[:, None] * (self.K @ (self.norm[:, None] * Q.astype(np.float64)))).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------- DenseCRF2D
def features_2d(H, W, sxy, rgb=None, srgb=None):
    """DenseCRF2D::addPairwiseGaussian / addPairwiseBilateral feature matrices, pixel order row-major (j * W + i)."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cols = [xx.reshape(-1) / np.float32(sxy), yy.reshape(-1) / np.float32(sxy)]
    if rgb is not None:
        im = np.asarray(rgb, dtype=np.float32).reshape(H * W, 3)
        cols += [im[:, k] / np.float32(srgb) for k in range(3)]
    return np.stack(cols, axis=1).astype(np.float32)


def _exp_and_normalize(x):                                            # columns = labels here: [N, M]
    e = np.exp(x - x.max(1, keepdims=True))
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def mean_field(unary, kernels, weights, iters):
    """DenseCRF::inference.  unary [N, M] (energies), kernels: objects with .apply([N, M]) -> [N, M], Potts weights."""
    U = unary.astype(np.float32)
    Q = _exp_and_normalize(-U)
    for _ in range(iters):
        tmp = -U
        for k, w in zip(kernels, weights):
            tmp = tmp + np.float32(w) * k.apply(Q)                    # tmp1 -= (-w * filter(Q))
        Q = _exp_and_normalize(tmp)
    return Q


def dense_crf(img, probs, sxy_g=3.0, compat_g=3.0, sxy_b=40.0, srgb=13.0, compat_b=10.0, iters=3, exact=False, return_q=False):
    """tools/seg_evaluation.py:31-54.  img [H, W, 3] uint8 (RGB), probs [H, W] in [0, 1] -> label map [H, W] in {0, 1}."""
    H, W = probs.shape
    p = np.stack([1.0 - probs.astype(np.float32), probs.astype(np.float32)], axis=0)      # :36-37
    U = (-np.log(p + 1e-8)).reshape(2, -1).T.astype(np.float32)                          # :41-46 -> [N, 2]
    Kern = ExactKernel if exact else LatticeKernel
    kernels = [Kern(features_2d(H, W, sxy_g)), Kern(features_2d(H, W, sxy_b, img, srgb))]
    Q = mean_field(U, kernels, [compat_g, compat_b], iters)
    lab = np.argmax(Q, axis=1).reshape(H, W)                                             # :52 (ties -> label 0)
    return (lab, Q.reshape(H, W, 2)) if return_q else lab


def mean_field_exact(img, probs, **kw):
    return dense_crf(img, probs, exact=True, **kw)
