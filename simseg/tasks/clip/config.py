"""CLIP/SimSeg task defaults (the key surface of simseg/tasks/clip/config.py:9-173) and its preprocessing hook."""
import os

from simseg.utils.collections import AttrDict


def _ns(**kw):
    d = AttrDict()
    for k, v in kw.items():
        d[k] = v
    return d


def task_cfg_init_fn(cfg):
    cfg.runner.update(name="clip", log_interval=1, val_interval=1, val_interval_steps=-1, stable_random="none")
    cfg.wandb = _ns(enable=False, project="your_proj", entity="your_entity", train_record_keys=["loss", "i2t_acc", "t2i_acc", "lr"])
    cfg.ckpt.update(dir="./output", step_interval=2000, filename="step_checkpoint.pth", external_resume=None,
                    only_load_image_encoder=False, only_load_text_encoder=False, soft_resume=False, auto_resume=True)
    cfg.log.update(interval_train=1, interval_val=1)
    cfg.dist.update(name="torch", param=dict(), fp16=True)
    cfg.optim.update(name="torch.optim.AdamW", param=dict(betas=(0.9, 0.98), eps=1e-6, weight_decay=0.1), grad_clip=dict())
    cfg.optim.lr.update(name="cosine_schedule_with_warmup", init=1e-4, warmup_proportion=0.025, param=dict(num_cycles=0.5))
    cfg.data.update(exp_name="test", name="parquet", train_type="sequential", train_name=["cc"], valid_name=["f30k", "coco"],
                    data_path="./data/", batch_size=128, batch_size_train=128, batch_size_val=256, num_workers=8,
                    enable_valid=True, single_eval=True, cuda_eval=True)
    cfg.transforms = _ns(
        input_size=224, train_transforms=["resize"], valid_transforms=["resize"],
        resize=_ns(size=224), resize_bicubic=_ns(size=224),
        normalize=_ns(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
        random_crop=_ns(size=224), center_crop=_ns(size=224), random_resize_crop=_ns(size=224, scale=[0.6, 1.0]),
        random_augment=_ns(N=2, M=7), random_erasing=_ns(reprob=0.0, remode="pixel", recount=1), color_jitter=0.4)
    cfg.model.update(
        name="clip", pretrain_prefix_change_list=[], max_length=25, syncbn=True, interpolate_pos_embed=False, freeze_cnn_bn=False,
        image_encoder=_ns(name="timm_modelzoo", tag="vit_base_patch16_224_in21k", embedding_dim=768, pretrained=True, trainable=True,
                          vit=AttrDict()),
        text_encoder=_ns(name="huggingface_modelzoo", tag="bert-base-uncased", embedding_dim=768, pretrained=True, trainable=True,
                         target_token_idx=0),
        projection=_ns(name="simple", dim=512, text_projector_trainable=True, image_projector_trainable=True,
                       complex_projection=_ns(drop_out=0.1)),
        pool=_ns(name="identity", loda=_ns(image_k=5, text_k=5)))
    cfg.loss = _ns(name="NCE", global_reduce=True, group_size=-1, smoothing=0.0, extra_losses=[],
                   nce_loss=_ns(gather_backward=False), temperature=_ns(name="constant", value=0.02),
                   triplet_loss=_ns(reduce_mode="max", margin=0.2))


def update_clip_config(cfg):
    cfg.ckpt.dir = os.path.join(cfg.ckpt.dir, cfg.data.exp_name)
    for key in ("batch_size", "batch_size_val"):
        if isinstance(cfg.data[key], list):
            cfg.data[key] = cfg.data[key][0]
