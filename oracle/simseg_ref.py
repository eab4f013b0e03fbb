"""CPU oracle for the SimSeg hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-torch fp32 restatement of the arithmetic the reference executes on the path
BASELINE.json's north_star names.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this file; the product (simseg_amd/, simseg/)
never does and fails loudly when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * projection / LoDA top-k pooling / L2norm / NCE / GatherLayer / RetrievalMetric /
    seg similarity block / CLIPModel glue: PINNED against the reference itself, imported
    in the build container by oracle/make_golden.py (fixtures in tests/golden/).
  * BERT text tower: the arithmetic lives in HuggingFace transformers (pinned 4.21.3 in
    /root/reference/requirements.txt:13, absent from /root/reference).  PINNED against the
    installed transformers' BertModel (eager attention) by oracle/make_golden.py.
  * ViT image tower: the arithmetic lives in timm==0.6.13 (requirements.txt:9), which is
    not installed and not vendored, and no test of the reference pins it.  The restatement
    below follows timm 0.6.13 VisionTransformer as called by
    simseg/models/backbones/mml/vit_builder.py:13-21, and is PINNED AGAINST AN INDEPENDENT
    IMPLEMENTATION of the same architecture: the installed transformers' ViTModel (whose
    checkpoints are conversions of timm's, verified upstream against timm outputs), with
    its weights re-keyed to timm's names (oracle/make_golden.py gold_vit ->
    tests/golden/vit_hf_tiny.npz; bit-equal outputs).  Against timm itself: unpinned.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# architecture tables (timm tags used by configs/clip/simseg.vit-{s,b}.yaml:86)
# --------------------------------------------------------------------------------------
VIT_ARCH = {
    "vit_small_patch16_224_in21k": dict(dim=384, depth=12, heads=6),
    "vit_base_patch16_224_in21k": dict(dim=768, depth=12, heads=12),
    # tiny test-only architectures (not in the reference; used to keep fixtures small)
    "vit_test_patch16": dict(dim=128, depth=2, heads=2),
}
BERT_ARCH = {
    "bert-base-uncased": dict(vocab=30522, dim=768, depth=12, heads=12, ffn=3072, max_pos=512, type_vocab=2),
    "bert-test": dict(vocab=1000, dim=128, depth=2, heads=2, ffn=512, max_pos=128, type_vocab=2),
}


# --------------------------------------------------------------------------------------
# ViT  (timm 0.6.13 VisionTransformer, as driven by vit_builder.py:13-21)
# --------------------------------------------------------------------------------------
class _PatchEmbed(nn.Module):
    def __init__(self, img_size, dim):
        super().__init__()
        self.num_patches = (img_size // 16) ** 2
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)

    def forward(self, x):  # [B,3,H,W] -> [B,N,D], flatten order (h, w)
        return self.proj(x).flatten(2).transpose(1, 2)


class _ViTAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, C = x.shape
        hd = C // self.heads
        qkv = self.qkv(x).reshape(B, T, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, T, C)
        return self.proj(x)


class _ViTMlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))  # erf GELU (nn.GELU default)


class _ViTBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _ViTAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _ViTMlp(dim)

    def forward(self, x):  # pre-LN, no LayerScale / DropPath at timm defaults
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class RefViT(nn.Module):
    """State-dict names equal timm's: cls_token, pos_embed, patch_embed.proj.*, blocks.i.*, norm.*"""

    def __init__(self, tag, img_size):
        super().__init__()
        a = VIT_ARCH[tag]
        self.dim, self.depth, self.heads = a["dim"], a["depth"], a["heads"]
        self.patch_embed = _PatchEmbed(img_size, self.dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + self.patch_embed.num_patches, self.dim))
        self.pos_drop = nn.Identity()   # timm: Dropout(p=drop_rate=0.0)
        self.blocks = nn.Sequential(*[_ViTBlock(self.dim, self.heads) for _ in range(self.depth)])
        self.norm = nn.LayerNorm(self.dim, eps=1e-6)

    def forward(self, x):  # vit_builder.py:13-21 -- all tokens, final LN
        x = self.patch_embed(x)
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        x = torch.cat((cls, x), dim=1)
        x = x + self.pos_embed
        x = self.blocks(x)
        x = self.norm(x)
        return x


# --------------------------------------------------------------------------------------
# BERT encoder (HF BertModel(add_pooling_layer=False), huggingface_builder.py:10-17)
# --------------------------------------------------------------------------------------
class _BertEmbeddings(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.word_embeddings = nn.Embedding(a["vocab"], a["dim"], padding_idx=0)
        self.position_embeddings = nn.Embedding(a["max_pos"], a["dim"])
        self.token_type_embeddings = nn.Embedding(a["type_vocab"], a["dim"])
        self.LayerNorm = nn.LayerNorm(a["dim"], eps=1e-12)

    def forward(self, input_ids):
        L = input_ids.shape[1]
        pos = torch.arange(L, device=input_ids.device)
        x = self.word_embeddings(input_ids) + self.token_type_embeddings.weight[0] + self.position_embeddings(pos)
        return self.LayerNorm(x)


class _BertSelf(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)


class _BertSelfOutput(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=1e-12)


class _BertAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.self = _BertSelf(dim)
        self.output = _BertSelfOutput(dim, dim)


class _BertIntermediate(nn.Module):
    def __init__(self, dim, ffn):
        super().__init__()
        self.dense = nn.Linear(dim, ffn)


class _BertLayer(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.heads = a["heads"]
        self.attention = _BertAttention(a["dim"])
        self.intermediate = _BertIntermediate(a["dim"], a["ffn"])
        self.output = _BertSelfOutput(a["ffn"], a["dim"])

    def forward(self, x, add_mask):
        B, L, C = x.shape
        H, hd = self.heads, C // self.heads
        s = self.attention.self

        def split(t):
            return t.view(B, L, H, hd).permute(0, 2, 1, 3)

        q, k, v = split(s.query(x)), split(s.key(x)), split(s.value(x))
        scores = q @ k.transpose(-1, -2) / math.sqrt(hd) + add_mask
        p = scores.softmax(dim=-1)
        ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, L, C)
        x = self.attention.output.LayerNorm(self.attention.output.dense(ctx) + x)  # post-LN
        h = F.gelu(self.intermediate.dense(x))
        x = self.output.LayerNorm(self.output.dense(h) + x)
        return x


class _BertEncoder(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(a) for _ in range(a["depth"])])


class RefBert(nn.Module):
    """State-dict names equal HF BertModel's: embeddings.*, encoder.layer.i.*"""

    def __init__(self, tag):
        super().__init__()
        a = BERT_ARCH[tag]
        self.arch = a
        self.embeddings = _BertEmbeddings(a)
        self.encoder = _BertEncoder(a)

    def forward(self, input_ids, attention_mask):
        x = self.embeddings(input_ids)
        # HF: (1 - mask) * large_negative added to the scores, broadcast over heads and queries
        add_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * -10000.0
        for layer in self.encoder.layer:
            x = layer(x, add_mask)
        return x  # last_hidden_state


# --------------------------------------------------------------------------------------
# heads  (simseg/models/components/*)
# --------------------------------------------------------------------------------------
def l2norm(x, dim=-1, eps=1e-8):
    """components/normalization.py:6-11  x / (sqrt(sum x^2) + eps)"""
    return x / (x.pow(2).sum(dim=dim, keepdim=True).sqrt() + eps)


def topk_pool(x, k, attention_mask=None):
    """components/pooling.py:52-65 (LoDA): per (batch, channel) mean of the k largest over tokens.
    With a mask: masked tokens are set to -10000 and k = min(k, min valid length).  The reference
    mutates its input in place; the oracle works on a copy."""
    if attention_mask is not None:
        x = x.clone()
        x[attention_mask == 0] = -10000
        min_len = int(attention_mask.sum(1).min())
        k = min(k, min_len)
    return x.topk(k, dim=1)[0].mean(dim=1)


# --------------------------------------------------------------------------------------
# InfoNCE  (simseg/models/criteria/losses/mml_loss.py:51-103) -- global_reduce branch
# --------------------------------------------------------------------------------------
def nce_global(feat1, feat2_global, temperature, rank, ignore_mask=None, ignore_mask_global=None, smoothing=0.0):
    """One direction of the global InfoNCE.  feat1 [Bl,P] local, feat2_global [Bg,P] = all ranks'
    feat2 concatenated in rank order (what GatherLayer returns, utils/dist.py:333-344).
    Returns (loss scalar, top-1 acc scalar)."""
    n1 = feat1.shape[0]
    if ignore_mask is None:
        ignore_mask = torch.zeros(n1, dtype=feat1.dtype)
    if ignore_mask_global is None:
        ignore_mask_global = torch.zeros(feat2_global.shape[0], dtype=feat1.dtype)
    temp = torch.clamp(temperature, 0.001, 0.5)                       # :56
    feat2_global = feat2_global * (1 - ignore_mask_global[:, None])   # :70-71
    logits = feat1 @ feat2_global.T / temp                            # :73
    targets = torch.arange(n1 * rank, n1 * (rank + 1))                # :75
    logp = F.log_softmax(logits, dim=-1)
    nll = -logp.gather(1, targets[:, None])[:, 0]
    if smoothing > 0:                                                 # :350-376
        loss = (1 - smoothing) * nll + smoothing * (-logp.mean(dim=-1))
    else:
        loss = nll
    loss = (loss * (1 - ignore_mask)).mean()                          # :89-91
    keep = ignore_mask < 1
    pred = logits[keep].argmax(dim=1)
    acc = (pred == targets[keep]).float().sum() / keep.sum()          # utils/misc.py:462-478 (top-1)
    return loss, acc


def clip_loss(img_emb, txt_emb, img_global, txt_global, temperature, rank):
    """pipelines/clip.py:129-140: 0.5 * (i2t + t2i), each one NCE call."""
    l1, a1 = nce_global(img_emb, txt_global, temperature, rank)
    l2, a2 = nce_global(txt_emb, img_global, temperature, rank)
    return 0.5 * (l1 + l2), a1, a2


# --------------------------------------------------------------------------------------
# zero-shot segmentation similarity block  (tools/seg_evaluation.py:99-143)
# --------------------------------------------------------------------------------------
def seg_similarity(patch_proj, text_feat, eps=1e-12):
    """Full patch x class similarity map.  patch_proj [B,N,P] = image_projection(patch tokens)
    (seg_evaluation.py:102), rows F.normalize'd (:112), times every class embedding (:136 does it
    one class column at a time).  Returns [B,N,C]."""
    pn = patch_proj / patch_proj.norm(dim=-1, keepdim=True).clamp_min(eps)
    return pn @ text_feat.T


def seg_image_scores(pooled, text_feat):
    """seg_evaluation.py:119  image-level class scores."""
    return pooled @ text_feat.T


def seg_upsample(sim_bcn, n, patch=16):
    """seg_evaluation.py:137-139 nearest x16 upsample of one [n,n] map; input [..., n*n]."""
    m = sim_bcn.reshape(*sim_bcn.shape[:-1], n, n)
    return m.repeat_interleave(patch, dim=-2).repeat_interleave(patch, dim=-1)


# --------------------------------------------------------------------------------------
# retrieval  (simseg/tasks/clip/hooks/utils.py)
# --------------------------------------------------------------------------------------
def unique_by_gid(gid, emb):
    """hooks/utils.py:14-19: sort by gid, keep the LAST row of every run of equal gids."""
    g, idx = torch.sort(gid, stable=True)
    emb = emb[idx]
    uni, cnt = torch.unique_consecutive(g, return_counts=True)
    off = torch.cumsum(cnt, 0) - 1
    return uni, emb[off]


def retrieval_first_match_rank(left, left_gid, right, right_gid):
    """hooks/utils.py:36-42,64-66: rank (0-based) of the best-scoring right row with the same gid,
    restated without the sort: number of right rows scoring strictly higher than the best match.
    Equal to the reference's argsort-based rank on tie-free data.  Returns (has_match, rank)."""
    sim = left @ right.T
    match = left_gid[:, None] == right_gid[None, :]
    has = match.any(dim=1)
    best = torch.where(match, sim, torch.full_like(sim, -float("inf"))).max(dim=1)[0]
    rank = (sim > best[:, None]).sum(dim=1)
    return has, rank


def retrieval_recalls(left, left_gid, right, right_gid, bounds=(1, 5, 10)):
    """hooks/utils.py:59-75."""
    has, rank = retrieval_first_match_rank(left, left_gid, right, right_gid)
    tot = has.sum()
    return {f"R@{b}": ((rank[has] < b).sum() / tot).item() for b in bounds}


# --------------------------------------------------------------------------------------
# IoU  (simseg/utils/metrics.py:40-75)
# --------------------------------------------------------------------------------------
def intersect_and_union(pred, label, num_classes, ignore_index=255):
    keep = label != ignore_index
    pred, label = pred[keep], label[keep]
    inter = pred[pred == label]
    ai = torch.bincount(inter.long(), minlength=num_classes)[:num_classes].double()
    ap = torch.bincount(pred.long(), minlength=num_classes)[:num_classes].double()
    al = torch.bincount(label.long(), minlength=num_classes)[:num_classes].double()
    return ai, ap + al - ai


# --------------------------------------------------------------------------------------
# the whole model (pipelines/clip.py CLIPModel method surface)
# --------------------------------------------------------------------------------------
class _Proj(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.linear = nn.Linear(din, dout, bias=False)

    def forward(self, x):
        return self.linear(x)


class _Wrap(nn.Module):
    """image_encoder.model.model / text_encoder.model.model nesting of the reference."""

    def __init__(self, inner):
        super().__init__()
        self.model = inner


class RefCLIP(nn.Module):
    def __init__(self, vit_tag="vit_base_patch16_224_in21k", bert_tag="bert-base-uncased", img_size=224,
                 proj_dim=512, image_k=5, text_k=1, temperature=0.02):
        super().__init__()
        vit = RefViT(vit_tag, img_size)
        bert = RefBert(bert_tag)
        self.image_encoder = _Wrap(_Wrap(vit))
        self.text_encoder = _Wrap(_Wrap(bert))
        self.image_projection = _Proj(vit.dim, proj_dim)
        self.text_projection = _Proj(bert.arch["dim"], proj_dim)
        self.image_k, self.text_k = image_k, text_k
        self.loss = nn.Module()
        self.loss.temperature = nn.Parameter(torch.ones([]) * temperature)

    @property
    def vit(self):
        return self.image_encoder.model.model

    @property
    def bert(self):
        return self.text_encoder.model.model

    def forward_image_feature(self, image):          # clip.py:65-84 (pool != identity -> drop cls)
        return self.vit(image)[:, 1:]

    def forward_image_project(self, feats):          # clip.py:87-93
        return l2norm(topk_pool(self.image_projection(feats), self.image_k))

    def forward_text_feature(self, input_ids, attention_mask):   # clip.py:96-108, target_token_idx=0
        return self.bert(input_ids, attention_mask)

    def forward_text_project(self, feats, attention_mask):       # clip.py:111-120
        return l2norm(topk_pool(self.text_projection(feats), self.text_k, attention_mask))

    def embeddings(self, image, input_ids, attention_mask):      # clip.py:152-168 embeddings='all'
        i = self.forward_image_project(self.forward_image_feature(image))
        t = self.forward_text_project(self.forward_text_feature(input_ids, attention_mask), attention_mask)
        return i, t

    def forward_loss_local(self, image, input_ids, attention_mask):
        """World-size-1 training forward: clip.py:152-176 with Bg == Bl."""
        i, t = self.embeddings(image, input_ids, attention_mask)
        return clip_loss(i, t, i, t, self.loss.temperature, 0)


def init_weights_(module, seed=0, std=0.02):
    """Deterministic synthetic init used by fixtures/bench (there are no checkpoints offline):
    trunc-normal-ish N(0, std) weights, zero biases, LN gamma=1 beta=0; cls/pos N(0,std)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("temperature"):
                continue
            if "norm" in name.lower() and name.endswith("weight"):
                p.fill_(1.0)
            elif "norm" in name.lower() and name.endswith("bias"):
                p.zero_()
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * std)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return module


def synthetic_text(batch, length, vocab, seed, min_len=3):
    """Synthetic BERT-style token ids (no tokenizer/vocab offline): [CLS]=101 body [SEP]=102 pad=0."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(batch, length, dtype=torch.long)
    mask = torch.zeros(batch, length, dtype=torch.long)
    hi = min(vocab, 30522)
    for b in range(batch):
        n = int(torch.randint(min_len, length + 1, (1,), generator=g))
        ids[b, 0] = 101 % hi
        if n > 2:
            ids[b, 1:n - 1] = torch.randint(min(1000, hi // 2), hi, (n - 2,), generator=g)
        ids[b, n - 1] = 102 % hi
        mask[b, :n] = 1
    return ids, mask
