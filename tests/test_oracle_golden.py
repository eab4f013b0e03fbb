"""The oracle (oracle/simseg_ref.py) against the golden vectors captured from the reference itself
(oracle/make_golden.py).  CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

from conftest import tt
from oracle import simseg_ref as R

TOL = dict(rtol=1e-5, atol=1e-6)


def _load_sd(module, g, prefix="sd."):
    sd = {k[len(prefix):]: tt(g[k]) for k in g.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("pos_drop" in m for m in missing), missing
    return module


def test_heads_fwd_bwd(golden):
    g = golden("heads")
    x = tt(g["x"]).requires_grad_(True)
    w = tt(g["w"]).requires_grad_(True)
    tok = x @ w.T
    pooled = R.topk_pool(tok, 5)
    emb = R.l2norm(pooled)
    torch.testing.assert_close(tok, tt(g["tok"]), **TOL)
    torch.testing.assert_close(pooled, tt(g["pooled"]), **TOL)
    torch.testing.assert_close(emb, tt(g["emb"]), **TOL)
    emb.backward(tt(g["gy"]))
    torch.testing.assert_close(x.grad, tt(g["gx"]), **TOL)
    torch.testing.assert_close(w.grad, tt(g["gw"]), rtol=1e-4, atol=1e-5)


def test_masked_text_pool(golden):
    g = golden("heads")
    t = tt(g["t"]).requires_grad_(True)
    mask = tt(g["mask"])
    tp = R.topk_pool(t, 1, mask)
    temb = R.l2norm(tp)
    torch.testing.assert_close(tp, tt(g["tpool"]), **TOL)
    torch.testing.assert_close(temb, tt(g["temb"]), **TOL)
    temb.backward(tt(g["gt"]))
    torch.testing.assert_close(t.grad, tt(g["gt_in"]), **TOL)
    # k clipped to the shortest caption of the batch (pooling.py:61-63)
    torch.testing.assert_close(R.topk_pool(tt(g["t"]), 3, tt(g["mask2"])), tt(g["tpool_k3"]), **TOL)


def test_retrieval(golden):
    g = golden("retrieval")
    ug, ue = R.unique_by_gid(tt(g["gid_rows"]), tt(g["img_rows"]))
    assert torch.equal(ug, tt(g["uni_gid"]))
    torch.testing.assert_close(ue, tt(g["uni_emb"]), rtol=0, atol=0)
    i2t = R.retrieval_recalls(ue, ug, tt(g["txt"]), tt(g["gid_txt"]))
    t2i = R.retrieval_recalls(tt(g["txt"]), tt(g["gid_txt"]), ue, ug)
    np.testing.assert_allclose([i2t["R@1"], i2t["R@5"], i2t["R@10"]], g["i2t"], atol=1e-7)
    np.testing.assert_allclose([t2i["R@1"], t2i["R@5"], t2i["R@10"]], g["t2i"], atol=1e-7)
    assert 0.05 < g["i2t"][0] < 0.999   # the fixture is non-trivial


def test_miou(golden):
    g = golden("miou")
    for i in range(3):
        a, u = R.intersect_and_union(tt(g["pred"][i]), tt(g["gt"][i]).long(), 21)
        np.testing.assert_allclose(a.numpy(), g["inter"][i])
        np.testing.assert_allclose(u.numpy(), g["union"][i])


def test_seg_block(golden):
    g = golden("seg_block")
    sim = R.seg_similarity(tt(g["proj"]), tt(g["text"]))            # [B,N,C]
    maps = sim.transpose(1, 2).reshape(2, 21, 18, 18)
    torch.testing.assert_close(maps, tt(g["maps"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(R.seg_upsample(sim[0, :, 3], 18), tt(g["up_b0_c3"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(R.seg_image_scores(tt(g["pooled"]), tt(g["text"])), tt(g["scores"]), rtol=1e-5, atol=1e-6)


def test_bert_vs_hf(golden):
    g = golden("bert_tiny")
    m = _load_sd(R.RefBert("bert-test"), g).eval()
    for tag in ("a", "b"):
        with torch.no_grad():
            y = m(tt(g[f"ids_{tag}"]), tt(g[f"mask_{tag}"]))
        torch.testing.assert_close(y, tt(g[f"out_{tag}"]), rtol=1e-4, atol=2e-5)


def test_vit_vs_hf(golden):
    """RefViT against the installed transformers ViTModel (the HF port of the timm architecture the reference uses), fixture
    written by oracle/make_golden.py gold_vit under timm's parameter names."""
    g = golden("vit_hf_tiny")
    m = _load_sd(R.RefViT("vit_test_patch16", 96), g).eval()
    for tag in ("a", "b"):
        with torch.no_grad():
            y = m(tt(g[f"image_{tag}"]))
        torch.testing.assert_close(y, tt(g[f"out_{tag}"]), rtol=1e-5, atol=1e-5)


def _ref_clip(g):
    m = R.RefCLIP("vit_test_patch16", "bert-test", img_size=96)
    return _load_sd(m, g).eval()


def test_clip_glue(golden):
    g = golden("clip_glue")
    m = _ref_clip(g)
    with torch.no_grad():
        f = m.forward_image_feature(tt(g["image"]))
        torch.testing.assert_close(f, tt(g["img_feat"]), rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(m.image_projection(f), tt(g["img_tok"]), rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(m.forward_image_project(f), tt(g["img_emb"]), rtol=1e-4, atol=2e-5)
        t = m.forward_text_feature(tt(g["input_ids"]), tt(g["attention_mask"]))
        torch.testing.assert_close(t, tt(g["txt_feat"]), rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(m.forward_text_project(t, tt(g["attention_mask"])), tt(g["txt_emb"]), rtol=1e-4, atol=2e-5)


def _check_train(golden, world, group_size=None):
    """group_size: cfg.loss.group_size sub-groups (mml_loss.py:24-27) - rank r exchanges embeddings with the ranks of its group only
    (one host: consecutive ranks), its targets are indexed by its rank INSIDE the group."""
    gs = group_size or world
    g = golden(f"clip_train_ws{world}" + (f"g{gs}" if gs != world else ""))
    m = _ref_clip(golden("clip_glue"))
    embs = []
    for r in range(world):
        embs.append(m.embeddings(tt(g[f"r{r}.image"]), tt(g[f"r{r}.input_ids"]), tt(g[f"r{r}.attention_mask"])))
    members = lambda r: list(range(r // gs * gs, r // gs * gs + gs))      # noqa: E731
    # reference semantics (GatherLayer.backward = all_reduce(SUM) then slice, utils/dist.py:347-354; DDP is not
    # wrapped in the fixture run): each rank's parameter grads come from its own loss, with embedding grads
    # summed over every rank's loss before flowing into this rank's towers.
    losses = []
    for r in range(world):
        ig = torch.cat([embs[j][0] for j in members(r)]); tg = torch.cat([embs[j][1] for j in members(r)])
        loss, a1, a2 = R.clip_loss(embs[r][0], embs[r][1], ig, tg, m.loss.temperature, r % gs)
        np.testing.assert_allclose(loss.item(), g[f"r{r}.loss"], rtol=2e-5)
        np.testing.assert_allclose(a1.item(), g[f"r{r}.i2t_acc"], atol=1e-6)
        np.testing.assert_allclose(a2.item(), g[f"r{r}.t2i_acc"], atol=1e-6)
        losses.append(loss)
    if world == 1:
        m.zero_grad()
        losses[0].backward()
        params = dict(m.named_parameters())
        for k in g.files:
            if k.startswith("r0.grad."):
                name = k[len("r0.grad."):]
                torch.testing.assert_close(params[name].grad, tt(g[k]), rtol=2e-3, atol=2e-6, msg=lambda s: name + s)
    elif world >= 4:
        # every rank's parameter gradients: the towers of rank r see d/d(emb_r) of the SUM of its group's losses (the all-reduce of the
        # gathered gradient), the temperature only rank r's own loss (it is not gathered)
        params = dict(m.named_parameters())
        for r in range(world):
            mine = m.embeddings(tt(g[f"r{r}.image"]), tt(g[f"r{r}.input_ids"]), tt(g[f"r{r}.attention_mask"]))
            e = {j: (mine if j == r else (embs[j][0].detach(), embs[j][1].detach())) for j in members(r)}
            ig = torch.cat([e[j][0] for j in members(r)]); tg = torch.cat([e[j][1] for j in members(r)])
            tot = 0
            for j in members(r):
                temp = m.loss.temperature if j == r else m.loss.temperature.detach()
                tot = tot + R.clip_loss(e[j][0], e[j][1], ig, tg, temp, j % gs)[0]
            m.zero_grad()
            tot.backward()
            for k in g.files:
                if k.startswith(f"r{r}.grad."):
                    name = k[len(f"r{r}.grad."):]
                    want = tt(g[k])
                    got = params[name].grad
                    got = got[:want.shape[0]] if got.dim() == 2 and got.shape != want.shape else got      # (large matrices: first 16 rows stored)
                    torch.testing.assert_close(got, want, rtol=2e-3, atol=2e-6, msg=lambda s: f"rank {r} {name}" + s)
    # pure-loss fixture with ignore_mask (mml_loss.py:70-71,89-93)
    f1 = [tt(g[f"r{r}.nce_f1"]).requires_grad_(True) for r in range(world)]
    f2 = [tt(g[f"r{r}.nce_f2"]).requires_grad_(True) for r in range(world)]
    ign = [tt(g[f"r{r}.nce_ign"]) for r in range(world)]
    temp = torch.tensor(0.02, requires_grad=True)
    tot = 0
    ls = []
    for r in range(world):
        l, acc = R.nce_global(f1[r], torch.cat([f2[j] for j in members(r)]), temp, r % gs, ign[r], torch.cat([ign[j] for j in members(r)]))
        np.testing.assert_allclose(l.item(), g[f"r{r}.nce_loss"], rtol=2e-5)
        np.testing.assert_allclose(acc.item(), g[f"r{r}.nce_acc"], atol=1e-6)
        ls.append(l)
    sum(ls).backward()       # all_reduce(SUM) of the gathered grad == grad of the sum of all ranks' losses
    for r in range(world):
        torch.testing.assert_close(f1[r].grad, tt(g[f"r{r}.nce_g1"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(f2[r].grad, tt(g[f"r{r}.nce_g2"]), rtol=1e-4, atol=1e-6)
    # per-rank temperature grad comes from that rank's own loss only
    for r in range(world):
        temp2 = torch.tensor(0.02, requires_grad=True)
        l, _ = R.nce_global(f1[r].detach(), torch.cat([f2[j] for j in members(r)]).detach(), temp2, r % gs, ign[r], torch.cat([ign[j] for j in members(r)]))
        l.backward()
        np.testing.assert_allclose(temp2.grad.item(), g[f"r{r}.nce_gt"], rtol=1e-4)


def test_clip_train_ws1(golden):
    _check_train(golden, 1)


def test_clip_train_ws2(golden):
    _check_train(golden, 2)


def test_clip_train_ws4(golden):
    _check_train(golden, 4)


def test_clip_train_ws4_groups_of_two(golden):
    _check_train(golden, 4, group_size=2)
