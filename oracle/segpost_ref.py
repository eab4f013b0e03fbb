"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's per-image segmentation post-processing
(tools/seg_evaluation.py:112-170) in numpy / torch, one image at a time, written as explicit loops.

Pinning: `intersect_and_union` is pinned against the reference's own function (tests/golden/miou.npz, oracle/make_golden.py).
The stage between the normalised map and the morphology is a CPU DenseCRF from `pydensecrf` (requirements.txt) and the
morphology / resize are `cv2` calls; neither package exists in this image, so that stretch is PARITY UNPINNED: the CRF is
replaced by its unary decision (prob > 0.5, see dense_crf :34-41: U = -log([1-p, p]), argmax of the unaries alone), and
cv2.dilate / cv2.erode / cv2.resize(INTER_NEAREST) are restated from their documented semantics (7x7 rectangular kernel,
anchor at the centre, one iteration, border pixels never win; src = floor(dst * src_size / dst_size)) and cross-checked
against scipy.ndimage in tests/test_oracle_golden.py.
"""
import numpy as np
import torch


def select_candidates(scores, top_cls_num, ncand=5):
    """:121-123, :128-133, :145-146.  scores [C] float32 tensor -> (idx list with -1 for skipped slots, scores, threshold)."""
    topk_scores, topk_index = scores.topk(min(top_cls_num, scores.numel()))
    threshold = topk_scores.mean() + 1.0 * topk_scores.std()
    idx, sc = [], []
    broke = False
    for i, index in enumerate(topk_index[:ncand].tolist()):
        s = float(scores[index])
        sc.append(s)
        if index in [0, 255]:
            idx.append(-1)
            continue
        if broke or s < float(threshold):
            broke = True
            idx.append(-1)
            continue
        idx.append(index)
    while len(idx) < ncand:
        idx.append(-1); sc.append(0.0)
    return idx, sc, float(threshold)


def normalised_map(sim_col, n, patch=16):
    """:135-149: column of the similarity map -> n x n -> nearest x16 -> min-max normalise.  Returns (prob [16n,16n], binary).
    n may be (nh, nw): the stitched patch grid of a sliding-window evaluation."""
    nh, nw = (n, n) if isinstance(n, int) else n
    a = sim_col.reshape(nh, nw).astype(np.float32)
    up = np.repeat(np.repeat(a, patch, axis=0), patch, axis=1)
    mn, mx = up.min(), up.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = (up - mn) / (mx - mn)
    binary = (norm > 0.5).astype(np.uint8) * 255            # dense_crf with the pairwise terms off: argmax of [1-p, p]
    return norm, binary


def morph7(img, erode):
    """cv2.dilate / cv2.erode(img, np.ones((7,7)), iterations=1), default border: out-of-image pixels never win."""
    H, W = img.shape
    out = np.empty_like(img)
    for y in range(H):
        y0, y1 = max(0, y - 3), min(H, y + 4)
        for x in range(W):
            x0, x1 = max(0, x - 3), min(W, x + 4)
            win = img[y0:y1, x0:x1]
            out[y, x] = win.min() if erode else win.max()
    return out


def morph7_fast(img, erode):
    """Same result as morph7 (tests/test_segpost.py checks it), vectorised: a 7x7 max / min is separable; out-of-image pixels never win."""
    pad = 255 if erode else 0
    red = np.minimum if erode else np.maximum
    H, W = img.shape
    p = np.full((H + 6, W + 6), pad, dtype=img.dtype)
    p[3:3 + H, 3:3 + W] = img
    rows = p[:, 0:W].copy()
    for d in range(1, 7):
        rows = red(rows, p[:, d:d + W])
    out = rows[0:H].copy()
    for d in range(1, 7):
        out = red(out, rows[d:d + H])
    return out


def resize_nearest(img, H, W):
    """cv2.resize(img, (W, H), interpolation=cv2.INTER_NEAREST): src index = min(floor(dst * src / dst_size), src - 1)."""
    h, w = img.shape
    ys = np.minimum(np.floor(np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def intersect_and_union(pred, label, num_classes, ignore_index=255):
    """simseg/utils/metrics.py:52-74 (pinned by tests/golden/miou.npz through oracle/simseg_ref.intersect_and_union)."""
    pred = torch.as_tensor(pred).long()
    label = torch.as_tensor(label).long()
    m = label != ignore_index
    pred, label = pred[m], label[m]
    inter = pred[pred == label]
    a_i = torch.histc(inter.float(), bins=num_classes, min=0, max=num_classes - 1)
    a_p = torch.histc(pred.float(), bins=num_classes, min=0, max=num_classes - 1)
    a_l = torch.histc(label.float(), bins=num_classes, min=0, max=num_classes - 1)
    return a_i, a_p, a_l


def segment_image(sim, scores, label, n, top_cls_num, ncand=5, closing=True, fast_morph=None):
    """One image.  sim [n*n, C] float32 numpy, scores [C] torch float32, label [H,W] uint8 numpy ->
    dict(pred [H,W] int64, cand_idx, cand_score, threshold, masks [ncand,16n,16n], hist [3,C])."""
    C = sim.shape[1]
    H, W = label.shape
    idx, sc, thr = select_candidates(scores, top_cls_num, ncand)
    temp_pred = np.zeros((C, H, W))                                              # :126 (float64)
    nh, nw = (n, n) if isinstance(n, int) else n
    masks = np.zeros((ncand, 16 * nh, 16 * nw), np.uint8)
    mf = fast_morph or morph7
    for k, index in enumerate(idx):
        if index < 0:
            continue
        _, binary = normalised_map(sim[:, index], n)
        final = mf(mf(binary, False), True) if closing else binary               # :155-157
        masks[k] = final
        temp_pred[index] = resize_nearest(final, H, W) * sc[k]                   # :159-160
    pred = temp_pred.argmax(0)                                                   # :163
    a_i, a_p, a_l = intersect_and_union(pred, label, C)
    return {"pred": pred, "cand_idx": idx, "cand_score": sc, "threshold": thr, "masks": masks,
            "hist": torch.stack([a_i, a_p, a_l]).long()}


# ---- sliding-window form (BASELINE configs[3] / SURVEY.md 8d cfg 4; the reference tool has only the single resize, :84-85,109) --------
def stitch_windows(win_maps, wy, wx, n, step):
    """win_maps [wy*wx, n*n, C] float32 (window (i, j) at patch offset (i*step, j*step), row-major over the window grid) ->
    [nh*nw, C]: per cell the mean over the covering windows, accumulated in float32 in (i, j) order - written as the explicit loop."""
    C = win_maps.shape[-1]
    nh, nw = n + (wy - 1) * step, n + (wx - 1) * step
    acc = np.zeros((nh, nw, C), np.float32)
    cnt = np.zeros((nh, nw), np.int32)
    for i in range(wy):
        for j in range(wx):
            m = win_maps[i * wx + j].reshape(n, n, C).astype(np.float32)
            acc[i * step:i * step + n, j * step:j * step + n] += m
            cnt[i * step:i * step + n, j * step:j * step + n] += 1
    return (acc / cnt[..., None].astype(np.float32)).reshape(nh * nw, C)


def sliding_window_image(win_maps, win_scores, label, wy, wx, n, step, top_cls_num, ncand=5, closing=True, fast_morph=None):
    """One SOURCE image evaluated through wy*wx windows: win_maps [wy*wx, n*n, C] (each window's normalised patch x class map),
    win_scores [wy*wx, C] (each window's pooled-embedding class scores).  Maps are overlap-averaged on the source patch grid, scores are
    averaged over the windows (float32, window order), and the reference's per-image body runs once on the stitched map."""
    sim = stitch_windows(win_maps, wy, wx, n, step)
    sc = np.zeros(win_scores.shape[1], np.float32)
    for w in range(wy * wx):
        sc += win_scores[w].astype(np.float32)
    sc = sc / np.float32(wy * wx)
    nh, nw = n + (wy - 1) * step, n + (wx - 1) * step
    out = segment_image(sim, torch.from_numpy(sc), label, (nh, nw), top_cls_num, ncand, closing, fast_morph)
    out["sim"], out["scores"] = sim, sc
    return out
