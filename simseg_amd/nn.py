"""nn.Module shells of the two towers.  They own the parameters under exactly the names timm / HuggingFace use
(checkpoint compatibility, SURVEY.md 8b "state-dict layout") and dispatch forward to the HIP towers; torch layers
(nn.Linear, nn.LayerNorm, nn.Conv2d, nn.Embedding) serve as parameter containers only -- their forward never runs."""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import towers

VIT_ARCH = {
    "vit_small_patch16_224_in21k": dict(dim=384, depth=12, heads=6),
    "vit_base_patch16_224_in21k": dict(dim=768, depth=12, heads=12),
    "vit_small_patch16_224": dict(dim=384, depth=12, heads=6),
    "vit_base_patch16_224": dict(dim=768, depth=12, heads=12),
    "vit_test_patch16": dict(dim=128, depth=2, heads=2),
}
BERT_ARCH = {
    "bert-base-uncased": dict(vocab=30522, dim=768, depth=12, heads=12, ffn=3072, max_pos=512, type_vocab=2),
    "bert-test": dict(vocab=1000, dim=128, depth=2, heads=2, ffn=512, max_pos=128, type_vocab=2),
}


def compute_dtype():
    """The 16-bit type torch.autocast is set to (the reference trains under autocast, clip_runner.py:226-228: fp16 there, with a
    GradScaler - the benchmark headline since round 5; bf16 is the default 16-bit type of this package's own trainer) or the one SIMSEG_AMD_COMPUTE names (bf16 / fp16); exact fp32 otherwise (the
    eval tools run fp32)."""
    env = os.environ.get("SIMSEG_AMD_COMPUTE", "").lower()
    if env in ("bf16", "bfloat16"):
        return torch.bfloat16
    if env in ("fp16", "float16", "half"):
        return torch.float16
    if env in ("fp32", "float32"):
        return torch.float32
    if torch.is_autocast_enabled():
        try:
            return torch.float16 if torch.get_autocast_dtype("cuda") == torch.float16 else torch.bfloat16
        except Exception:       # noqa: BLE001  (older torch)
            return torch.float16 if torch.get_autocast_gpu_dtype() == torch.float16 else torch.bfloat16
    return torch.float32


_seed_state = [0x5EED]


def _rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    return int(os.environ.get("RANK", "0") or 0)


def next_dropout_seed():
    """Seed of the next forward's dropout masks: a per-process LCG stream, offset by the data-parallel rank so that the ranks of
    a job draw different masks (torch seeds every rank's generator separately, simseg/utils/initial.py)."""
    _seed_state[0] = (_seed_state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 62)
    return (_seed_state[0] + _rank() * 0x9E3779B97F4A7C15) % (1 << 62)


def manual_dropout_seed(seed):
    _seed_state[0] = int(seed)


def _init_linear(m, std=0.02):
    nn.init.trunc_normal_(m.weight, std=std)
    if m.bias is not None:
        nn.init.zeros_(m.bias)


# ------------------------------------------------------------------------------------------------------------------
class _PatchEmbed(nn.Module):
    def __init__(self, img_size, dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (16, 16)
        self.grid_size = (img_size // 16, img_size // 16)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(3, dim, kernel_size=16, stride=16)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _Block(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)


class ViT(nn.Module):
    """timm VisionTransformer(patch16, num_classes=0) feature extractor: all tokens after the final norm."""

    def __init__(self, tag, img_size=224):
        super().__init__()
        if tag not in VIT_ARCH:
            raise KeyError(f"unknown ViT tag {tag!r}; known: {sorted(VIT_ARCH)}")
        a = VIT_ARCH[tag]
        if a["dim"] != a["heads"] * 64:
            raise ValueError("the attention kernels are specialised for head_dim 64")
        self.embed_dim, self.num_heads = a["dim"], a["heads"]
        self.patch_embed = _PatchEmbed(img_size, a["dim"])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, a["dim"]))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + self.patch_embed.num_patches, a["dim"]))
        self.blocks = nn.Sequential(*[_Block(a["dim"]) for _ in range(a["depth"])])
        self.norm = nn.LayerNorm(a["dim"], eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _init_linear(m)

    def forward(self, x):
        return towers.vit_forward(self, x, compute_dtype())


# ------------------------------------------------------------------------------------------------------------------
class _BertEmbeddings(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.word_embeddings = nn.Embedding(a["vocab"], a["dim"], padding_idx=0)
        self.position_embeddings = nn.Embedding(a["max_pos"], a["dim"])
        self.token_type_embeddings = nn.Embedding(a["type_vocab"], a["dim"])
        self.LayerNorm = nn.LayerNorm(a["dim"], eps=1e-12)
        self.register_buffer("position_ids", torch.arange(a["max_pos"]).expand((1, -1)).clone(), persistent=True)  # persistent in the pinned transformers 4.21.3


class _BertSelf(nn.Module):
    """HF BertSelfAttention's parameters: three Linear modules (the names checkpoints and the optimizer's regex rules use).  Their
    storage is packed back to back - query, key, value - so that the fused [3D, D] projection GEMM and its [3D] bias read the
    masters in place (towers._as_one) instead of concatenating them every forward.  Re-packed after every device / dtype move."""

    def __init__(self, dim):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self._pack()

    @torch.no_grad()
    def _pack(self):
        for attr in ("weight", "bias"):
            ps = [getattr(m, attr) for m in (self.query, self.key, self.value)]
            flat = torch.cat([p.data for p in ps])
            o = 0
            for p in ps:
                p.data = flat[o:o + p.shape[0]]
                o += p.shape[0]

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._pack()
        return self


class _BertDenseLN(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)
        self.LayerNorm = nn.LayerNorm(dout, eps=1e-12)


class _BertAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.self = _BertSelf(dim)
        self.output = _BertDenseLN(dim, dim)


class _BertIntermediate(nn.Module):
    def __init__(self, dim, ffn):
        super().__init__()
        self.dense = nn.Linear(dim, ffn)


class _BertLayer(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.attention = _BertAttention(a["dim"])
        self.intermediate = _BertIntermediate(a["dim"], a["ffn"])
        self.output = _BertDenseLN(a["ffn"], a["dim"])


class _BertEncoder(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(a) for _ in range(a["depth"])])


class Bert(nn.Module):
    """HF BertModel(add_pooling_layer=False): returns an object with .last_hidden_state like the HF output."""

    def __init__(self, tag):
        super().__init__()
        if tag not in BERT_ARCH:
            raise KeyError(f"unknown text-encoder tag {tag!r}; known: {sorted(BERT_ARCH)}")
        a = BERT_ARCH[tag]
        self.arch = a
        self.num_heads = a["heads"]
        self.hidden_dropout_prob = 0.1              # BertConfig defaults, active in train() only
        self.attention_probs_dropout_prob = 0.1
        self.embeddings = _BertEmbeddings(a)
        self.encoder = _BertEncoder(a)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)

    def forward(self, input_ids, attention_mask=None, **kwargs):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        seed = next_dropout_seed() if self.training else 0
        h = towers.bert_forward(self, input_ids, attention_mask, compute_dtype(), training=self.training, seed=seed)
        return SimpleNamespace(last_hidden_state=h)
