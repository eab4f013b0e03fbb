"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference)
in the build container.  TEST INFRASTRUCTURE -- never imported by the product.

    python oracle/make_golden.py            # rewrites every fixture (deterministic seeds)

What is pinned here (and what is not) is listed in oracle/simseg_ref.py's header.  The reference's
Python never travels to the GPU box; only the .npz inputs/outputs written here do.

Stubs: `timm` and `wandb` are not installed.  `wandb` is never called on the path.  `timm.create_model`
is bound to the oracle's RefViT for the CLIPModel-glue fixtures only (the ViT block arithmetic is
"parity unpinned", see header); everything else executes reference code unmodified.
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.environ.get("SIMSEG_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")     # the regeneration check writes elsewhere
REF = "/root/reference"
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _import_reference():
    import transformers  # noqa: F401  (must precede the timm stub: its lazy loader find_spec()s timm)
    for name in ("timm", "wandb"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    # the reference's `simseg` package must shadow this repo's host-side mirror of the same name.  Spawned workers inherit the
    # parent's sys.path and then re-execute this module (which puts REPO in front again), so REF is always MOVED to the front.
    while REF in sys.path:
        sys.path.remove(REF)
    sys.path.insert(0, REF)
    if "simseg" in sys.modules and not os.path.abspath(sys.modules["simseg"].__file__).startswith(REF):
        raise RuntimeError("this repo's `simseg` mirror is already imported; fixtures must come from the reference package")
    os.environ.setdefault("HOSTNAME", "localhost")
    import simseg  # noqa: F401
    import simseg.utils  # noqa: F401
    import simseg.core  # noqa: F401  (core before models: the reference has an import cycle)
    import simseg.models  # noqa: F401


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ---------------------------------------------------------------------------------------------
def gold_heads():
    from simseg.models.components import SimpleProjection, TopKPooling, L2norm
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 324, 384, generator=g)
    proj = SimpleProjection(None, 384, 512)
    with torch.no_grad():
        proj.linear.weight.copy_(torch.randn(512, 384, generator=g) * 0.05)
    x.requires_grad_(True)
    tok = proj(x)
    pooled = TopKPooling(5, dim=1)(tok)
    emb = L2norm(pooled, dim=-1)
    gy = torch.randn(2, 512, generator=g)
    emb.backward(gy)
    # masked text pooling, k=1, ragged lengths 25 / 7 / 3
    t = torch.randn(3, 25, 512, generator=g)
    mask = torch.zeros(3, 25, dtype=torch.long)
    for b, n in enumerate((25, 7, 3)):
        mask[b, :n] = 1
    t_in = t.clone().requires_grad_(True)
    tp = TopKPooling(1, dim=1)(t_in * 1.0, mask)      # "*1.0": the reference mutates its input in place
    temb = L2norm(tp, dim=-1)
    gt = torch.randn(3, 512, generator=g)
    temb.backward(gt)
    # masked pooling with k=3 > min length 2  (k is clipped to the shortest caption in the batch)
    mask2 = mask.clone(); mask2[2, 2] = 0
    tp2 = TopKPooling(3, dim=1)(t.clone(), mask2)
    _save("heads", x=_np(x), w=_np(proj.linear.weight), tok=_np(tok), pooled=_np(pooled), emb=_np(emb),
          gy=_np(gy), gx=_np(x.grad), gw=_np(proj.linear.weight.grad),
          t=_np(t), mask=_np(mask), tpool=_np(tp), temb=_np(temb), gt=_np(gt), gt_in=_np(t_in.grad),
          mask2=_np(mask2), tpool_k3=_np(tp2))


def gold_retrieval():
    from simseg.tasks.clip.hooks.utils import IndexedEmbInfo, RetrievalMetric
    g = torch.Generator().manual_seed(5)
    img = torch.nn.functional.normalize(torch.randn(100, 64, generator=g), dim=-1)
    # 5 captions per image = noisy copies so that recalls are non-trivial
    txt = img.repeat_interleave(5, 0) + 0.35 * torch.randn(500, 64, generator=g)
    txt = torch.nn.functional.normalize(txt, dim=-1)
    gid_txt = torch.arange(500) // 5
    # the tools hold one image row per caption row, then .unique() (retrieval_evaluation.py:36-40)
    perm = torch.randperm(500, generator=g)
    img_rows = img.repeat_interleave(5, 0)[perm]
    gid_rows = gid_txt[perm]
    uni = IndexedEmbInfo("image", gid_rows, img_rows).unique()
    left = IndexedEmbInfo("image", uni.group_idx, uni.emb_mat)
    right = IndexedEmbInfo("text", gid_txt, txt)
    m = RetrievalMetric(with_prefix=False)
    i2t = m(left, right)
    t2i = m(right, left)
    _save("retrieval", img_rows=_np(img_rows), gid_rows=_np(gid_rows), txt=_np(txt), gid_txt=_np(gid_txt),
          uni_gid=_np(uni.group_idx), uni_emb=_np(uni.emb_mat),
          i2t=np.array([i2t["R@1"], i2t["R@5"], i2t["R@10"]]), t2i=np.array([t2i["R@1"], t2i["R@5"], t2i["R@10"]]))


def gold_miou():
    from simseg.utils.metrics import mean_iou
    g = torch.Generator().manual_seed(6)
    pred = torch.randint(0, 21, (3, 64, 64), generator=g)
    gt = torch.randint(0, 21, (3, 64, 64), generator=g)
    gt[torch.rand(3, 64, 64, generator=g) < 0.05] = 255
    pred = torch.where(torch.rand(3, 64, 64, generator=g) < 0.5, gt.clamp(max=20), pred)
    inter, union = [], []
    for i in range(3):
        a, b = mean_iou([pred[i].numpy()], [gt[i].numpy().astype(np.uint8)], 21, 255)
        inter.append(_np(a)); union.append(_np(b))
    _save("miou", pred=_np(pred), gt=_np(gt).astype(np.uint8), inter=np.stack(inter), union=np.stack(union))


def gold_interp_pe():
    from simseg.utils.interpolate_pe import interpolate_pos_embed
    g = torch.Generator().manual_seed(7)
    out = {}
    pe = torch.randn(1, 1 + 14 * 14, 96, generator=g)
    out["pe"] = _np(pe)
    for n in (18, 32):
        m = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=n * n),
                                  pos_embed=torch.zeros(1, 1 + n * n, 96))
        out[f"pe_{n}"] = _np(interpolate_pos_embed(pe.clone(), m))
    _save("interp_pe", **out)


def gold_seg_block():
    """tools/seg_evaluation.py:112-139 cannot be imported (cv2/pydensecrf/torchvision missing); these are
    the same torch calls that block makes, on captured-shape inputs."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(8)
    proj = torch.randn(2, 324, 512, generator=g) * 3.0
    text = F.normalize(torch.randn(21, 512, generator=g), dim=-1)
    pooled = F.normalize(torch.randn(2, 512, generator=g), dim=-1)
    im_f = F.normalize(proj, dim=-1, p=2)                                   # :112
    maps = torch.stack([torch.stack([(im_f[b] @ text[c].unsqueeze(-1)).reshape(18, 18)   # :136-137
                                     for c in range(21)]) for b in range(2)])
    up = F.interpolate(maps[0, 3][None, None], scale_factor=16, mode="nearest")[0][0]   # :139
    scores = torch.stack([torch.sum(pooled[b].unsqueeze(0) * text, dim=1) for b in range(2)])   # :119
    _save("seg_block", proj=_np(proj), text=_np(text), pooled=_np(pooled), maps=_np(maps), up_b0_c3=_np(up),
          scores=_np(scores))


def gold_bert():
    """Text tower arithmetic: installed HF transformers BertModel, eager attention, tiny config, eval mode."""
    from transformers import BertConfig, BertModel
    from oracle.simseg_ref import BERT_ARCH, synthetic_text
    a = BERT_ARCH["bert-test"]
    torch.manual_seed(3)
    conf = BertConfig(hidden_size=a["dim"], num_hidden_layers=a["depth"], num_attention_heads=a["heads"],
                      intermediate_size=a["ffn"], vocab_size=a["vocab"], max_position_embeddings=a["max_pos"],
                      attn_implementation="eager")
    m = BertModel(conf, add_pooling_layer=False).eval()
    with torch.no_grad():   # HF init leaves biases 0 and LN at identity: randomise so every term is exercised
        for n, p in m.named_parameters():
            if n.endswith("bias") or "LayerNorm" in n:
                p.add_(torch.randn_like(p) * 0.05)
    out = {}
    for tag, (B, L) in {"a": (4, 25), "b": (2, 77)}.items():
        ids, mask = synthetic_text(B, L, a["vocab"], seed=20 + L)
        with torch.no_grad():
            y = m(input_ids=ids, attention_mask=mask).last_hidden_state
        out[f"ids_{tag}"], out[f"mask_{tag}"], out[f"out_{tag}"] = _np(ids), _np(mask), _np(y)
    sd = {"sd." + k: _np(v) for k, v in m.state_dict().items() if "position_ids" not in k}
    _save("bert_tiny", **out, **sd)


def gold_vit():
    """Image tower arithmetic.  timm (the reference's ViT provider, requirements.txt:9) is not installed here; the installed HF
    transformers ViTModel is the same architecture (its checkpoints are converted from timm's and verified against timm's
    outputs upstream), so it pins the block arithmetic independently of our restatement: pre-LN blocks, fused-softmax
    attention with 1/sqrt(d) scaling, erf GELU, final LayerNorm, eps 1e-6, cls + learned position embeddings.  The
    state dict is saved under timm's parameter names (q/k/v concatenated into attn.qkv, as timm stores them)."""
    from transformers import ViTConfig, ViTModel
    from oracle.simseg_ref import VIT_ARCH
    a = VIT_ARCH["vit_test_patch16"]
    torch.manual_seed(5)
    conf = ViTConfig(hidden_size=a["dim"], num_hidden_layers=a["depth"], num_attention_heads=a["heads"], intermediate_size=4 * a["dim"],
                     image_size=96, patch_size=16, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, attn_implementation="eager")
    m = ViTModel(conf, add_pooling_layer=False).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():                # exercise every term: biases, LN affine, cls / pos embeddings
            p.copy_(torch.randn_like(p) * (0.05 if p.ndim > 1 else 0.1) + (1.0 if ("layernorm" in n and n.endswith("weight")) else 0.0))
    hf = m.state_dict()
    sd = {"cls_token": hf["embeddings.cls_token"], "pos_embed": hf["embeddings.position_embeddings"],
          "patch_embed.proj.weight": hf["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": hf["embeddings.patch_embeddings.projection.bias"],
          "norm.weight": hf["layernorm.weight"], "norm.bias": hf["layernorm.bias"]}
    for i in range(a["depth"]):
        h = f"layers.{i}." if f"layers.{i}.attention.q_proj.weight" in hf else None
        if h is None:
            raise RuntimeError("unexpected transformers ViT parameter naming: " + ", ".join(list(hf)[:12]))
        t = f"blocks.{i}."
        for wb in ("weight", "bias"):
            sd[t + "attn.qkv." + wb] = torch.cat([hf[h + f"attention.{x}_proj.{wb}"] for x in ("q", "k", "v")])
            sd[t + "attn.proj." + wb] = hf[h + f"attention.o_proj.{wb}"]
            sd[t + "norm1." + wb] = hf[h + f"layernorm_before.{wb}"]
            sd[t + "norm2." + wb] = hf[h + f"layernorm_after.{wb}"]
            sd[t + "mlp.fc1." + wb] = hf[h + f"mlp.fc1.{wb}"]
            sd[t + "mlp.fc2." + wb] = hf[h + f"mlp.fc2.{wb}"]
    out = {}
    for tag, B in {"a": 3, "b": 1}.items():
        x = torch.randn(B, 3, 96, 96, generator=torch.Generator().manual_seed(30 + B))
        with torch.no_grad():
            y = m(pixel_values=x).last_hidden_state
        out[f"image_{tag}"], out[f"out_{tag}"] = _np(x), _np(y)
    _save("vit_hf_tiny", **out, **{"sd." + k: _np(v) for k, v in sd.items()})


def gold_config():
    """update_cfg results (plain nested dicts, JSON) for both shipped YAMLs with the README's override styles."""
    import json
    from simseg.core.config import update_cfg
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    import simseg.core.config as rc

    def plain(d):
        return {k: plain(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}

    cases = {
        "vit-s": ("simseg.vit-s.yaml", []),
        "vit-b": ("simseg.vit-b.yaml", []),
        "vit-b-argv": ("simseg.vit-b.yaml", ["data.valid_name=[f30k,coco]", "data.batch_size_val=128", "transforms.input_size=224",
                                           "loss.temperature.value=0.05", "ckpt.dir=/tmp/x", "model.image_encoder.pretrained=False",
                                           "optim.lr.init=5e-5", "data.exp_name=abc"]),
    }
    out = {}
    for name, (y, argv) in cases.items():
        rc.cfg.set_this_dict_immutable(False)
        out[name] = dict(argv=argv, cfg=plain(update_cfg(task_cfg_init_fn, os.path.join(REF, "configs/clip", y), argv, update_clip_config)))
    errs = {}
    for name, argv in {"unknown_key": ["model.nope=1"], "type_mismatch": ["epoch=abc"]}.items():
        rc.cfg.set_this_dict_immutable(False)
        try:
            update_cfg(task_cfg_init_fn, os.path.join(REF, "configs/clip/simseg.vit-b.yaml"), argv, update_clip_config)
            errs[name] = None
        except Exception as e:   # noqa: BLE001
            errs[name] = type(e).__name__
    out["errors"] = errs
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "config.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote config.json", errs)


# ---------------------------------------------------------------------------------------------
TINY_ARGV = ["transforms.input_size=96", "model.image_encoder.tag=vit_test_patch16",
             "model.image_encoder.embedding_dim=128", "model.image_encoder.pretrained=False",
             "model.text_encoder.tag=bert-test", "model.text_encoder.embedding_dim=128",
             "model.text_encoder.pretrained=False"]


def _build_reference_clip(extra_argv, rank):
    """Reference CLIPModel (pipelines/clip.py) on a tiny architecture, eval-mode arithmetic."""
    import timm
    import transformers
    from transformers import BertConfig
    from oracle.simseg_ref import BERT_ARCH, RefViT, init_weights_
    from simseg.core.config import update_cfg
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    from simseg.models.pipelines.clip import CLIPModel

    def create_model(tag, pretrained=False, num_classes=0, img_size=224, **kw):
        return RefViT(tag, img_size)

    timm.create_model = create_model
    a = BERT_ARCH["bert-test"]

    def from_pretrained(tag, **kw):
        return BertConfig(hidden_size=a["dim"], num_hidden_layers=a["depth"], num_attention_heads=a["heads"],
                          intermediate_size=a["ffn"], vocab_size=a["vocab"], max_position_embeddings=a["max_pos"],
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, attn_implementation="eager")

    transformers.AutoConfig.from_pretrained = staticmethod(from_pretrained)
    import simseg.models.backbones.mml.huggingface_builder as hb
    hb.AutoConfig = transformers.AutoConfig
    _orig = hb.AutoModel.from_config
    hb.AutoModel.from_config = staticmethod(lambda config: transformers.BertModel(config, add_pooling_layer=False))
    cfg = update_cfg(task_cfg_init_fn, os.path.join(REF, "configs/clip/simseg.vit-s.yaml"),
                     TINY_ARGV + list(extra_argv), preprocess_fn=update_clip_config)
    model = CLIPModel(cfg, rank)
    hb.AutoModel.from_config = _orig
    init_weights_(model, seed=42)
    return model


def _tiny_batch(B, seed):
    from oracle.simseg_ref import synthetic_text
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, 3, 96, 96, generator=g)
    ids, mask = synthetic_text(B, 25, 1000, seed=seed + 1)
    return {"image": image, "input_ids": ids, "attention_mask": mask}


def gold_clip_glue():
    model = _build_reference_clip(["loss.global_reduce=False"], 0)
    model.eval()   # the reference's train() override returns None (pipelines/clip.py:51-62)
    batch = _tiny_batch(4, 100)
    with torch.no_grad():
        img_feat = model.forward_image_feature(batch["image"])
        img_emb = model.forward_image_project(img_feat)
        img_tok = model.image_projection(img_feat)
        txt_feat = model.forward_text_feature(batch["input_ids"], batch["attention_mask"])
        txt_emb = model.forward_text_project(txt_feat, batch["attention_mask"])
        both = model(batch, embeddings="all")
    assert torch.equal(both[0], img_emb) and torch.equal(both[1], txt_emb)
    sd = {"sd." + k: _np(v) for k, v in model.state_dict().items() if "position_ids" not in k}
    _save("clip_glue", image=_np(batch["image"]), input_ids=_np(batch["input_ids"]),
          attention_mask=_np(batch["attention_mask"]), img_feat=_np(img_feat), img_emb=_np(img_emb),
          img_tok=_np(img_tok), txt_feat=_np(txt_feat), txt_emb=_np(txt_emb), **sd)


def _dist_worker(rank, world, port, q, group_size=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    _import_reference()
    import simseg.utils.dist as rdist
    import simseg.models.criteria.losses.mml_loss as rloss
    from simseg.utils import ENV
    ENV.rank, ENV.local_rank, ENV.size = rank, rank, world
    if group_size is None:
        # utils/dist.py:195 moves pickled bytes to a HIP device; on CPU the world group is the only group we need
        rloss.generate_local_groups = lambda group_size: (dist.group.WORLD, rank)
        rdist.generate_local_groups = rloss.generate_local_groups
        extra = []
    else:
        # the 4-rank fixtures run the reference's REAL generate_local_groups (utils/dist.py:371-427: host-by-host packing, new_group on
        # every rank) and its own all_gather_object (:168-223).  The only thing that cannot run here is the device move at :195 / :198 /
        # :208 (`.to(my_local_rank)`, a HIP device index): ENV.local_rank is the string "cpu" while the loss is built, so those moves are
        # no-ops and not one line of the reference is replaced.
        ENV._local_rank = "cpu"                     # (the property's setter insists on an int)
        extra = [f"loss.group_size={group_size}"]
    model = _build_reference_clip(extra, rank)      # global_reduce=True, gather_backward=True (YAML)
    ENV.local_rank = rank
    if group_size is not None:
        want_rank = rank % group_size                # one host: consecutive ranks share a group
        assert model.loss.rank == want_rank and dist.get_world_size(model.loss.group) == group_size, (model.loss.rank, group_size)
    model.eval()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    batch = _tiny_batch(3, 200 + rank)
    loss_dict, a1, a2 = model(batch)
    loss = loss_dict["nce_loss"]
    loss.backward()
    names = ["image_projection.linear.weight", "text_projection.linear.weight", "loss.temperature",
             "image_encoder.model.model.blocks.0.attn.qkv.weight", "image_encoder.model.model.pos_embed",
             "text_encoder.model.model.encoder.layer.1.output.dense.weight",
             "text_encoder.model.model.embeddings.word_embeddings.weight"]
    if group_size is not None:
        names = names[:-1]                            # (the 4-rank fixtures leave out the [vocab, D] gradient: 0.5 MB per rank)
    params = dict(model.named_parameters())
    grads = {n: _np(params[n].grad) for n in names}
    if group_size is not None:                        # (4-rank fixtures: the first 16 rows of the large matrices - 4 ranks x 2 files)
        grads = {n: (v[:16] if v.ndim == 2 and v.shape[0] > 64 else v) for n, v in grads.items()}
    # pure-loss fixture: NCE on given embeddings with ignore_mask and label smoothing off/on
    g = torch.Generator().manual_seed(300 + rank)
    f1 = torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).requires_grad_(True)
    f2 = torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).requires_grad_(True)
    ign = torch.zeros(8); ign[2 + rank] = 1.0
    nce = model.loss
    nce.temperature.grad = None
    l_i, acc_i = nce(f1, f2, ignore_mask=ign)
    l_i.backward()
    out = dict(image=_np(batch["image"]), input_ids=_np(batch["input_ids"]), attention_mask=_np(batch["attention_mask"]),
               loss=_np(loss), i2t_acc=_np(a1), t2i_acc=_np(a2),
               nce_f1=_np(f1), nce_f2=_np(f2), nce_ign=_np(ign), nce_loss=_np(l_i), nce_acc=_np(acc_i),
               nce_g1=_np(f1.grad), nce_g2=_np(f2.grad), nce_gt=_np(nce.temperature.grad))
    out.update({"grad." + k: v for k, v in grads.items()})
    if rank == 0:   # same deterministic init as clip_glue.npz: the state dict is stored there only
        ref = np.load(os.path.join(GOLD, "clip_glue.npz"))
        for k, v in sd0.items():
            if "position_ids" not in k:
                assert np.array_equal(ref["sd." + k], _np(v)), k
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def gold_dist(world, group_size=None):
    """clip_train_ws{world}.npz (loss over the whole world) / clip_train_ws{world}g{group_size}.npz (cfg.loss.group_size sub-groups,
    mml_loss.py:24-27): every rank's batch, loss, accuracies, selected parameter gradients and a pure-NCE case, from a gloo run of the
    reference."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + world + (10 * group_size if group_size else 0)
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q, group_size)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join()
    flat = {}
    for r in range(world):
        for k, v in res[r].items():
            flat[k if k.startswith("sd.") else f"r{r}.{k}"] = v
    _save(f"clip_train_ws{world}" + (f"g{group_size}" if group_size and group_size != world else ""), **flat)


if __name__ == "__main__":
    which = sys.argv[1:] or ["config", "heads", "retrieval", "miou", "interp_pe", "seg_block", "bert", "vit", "clip_glue", "dist1", "dist2", "dist4", "dist4g2"]
    torch.set_num_threads(4)
    _import_reference()
    for w in which:
        if w.startswith("dist"):
            ws, _, gs = w[4:].partition("g")
            # dist1 / dist2: the world group handed to the loss directly; dist4: the reference's own group construction with
            # group_size = world; dist4g2: two sub-groups of two ranks
            gold_dist(int(ws), int(gs) if gs else (int(ws) if int(ws) >= 4 else None))
        else:
            globals()["gold_" + w]()
