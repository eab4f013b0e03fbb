"""Retrieval validation loader (Flickr30K / MSCOCO parquet) with the reference's call surface
(simseg/datasets/clip/clip_dataset.py:80-122, 211-234): `<data_path>/<name>/valid.parquet` with columns
imbytes, caption, image_id, id; one row per caption; DistributedSampler without shuffling."""
import os
from io import BytesIO

import torch
from PIL import Image

from simseg.transforms import build_transforms
from simseg.utils import ENV, logger

__all__ = ["ParquetDataset", "build_parquet_valid_loader", "load_tokenizer"]


def load_tokenizer(tag):
    """HF tokenizer of the text tower.  Offline boxes have no hub access: point $SIMSEG_TOKENIZER_DIR at a directory holding
    the tokenizer files (vocab.txt ...) of `tag`."""
    from transformers import AutoTokenizer
    local = os.environ.get("SIMSEG_TOKENIZER_DIR")
    try:
        return AutoTokenizer.from_pretrained(local if local else tag)
    except OSError as e:
        raise OSError(f"tokenizer for {tag!r} is not available offline; set SIMSEG_TOKENIZER_DIR to a local copy ({e})") from e


class ParquetDataset(torch.utils.data.Dataset):
    def __init__(self, cfg, dataset_name, tokenizer, data_path, transforms=None):
        import pyarrow.parquet as pq
        self.cfg, self.name, self.transforms, self.tokenizer = cfg, dataset_name, transforms, tokenizer
        self.target_len = cfg.model.max_length
        self.data_path = os.path.join(data_path, dataset_name, "valid.parquet")
        df = pq.read_table(self.data_path).to_pandas()
        self.images, self.captions = df["imbytes"], df["caption"]
        self.image_ids, self.caption_ids = df["image_id"], df["id"]
        self.length = len(self.captions)

    def __getitem__(self, index):
        caption = self.captions[index]
        enc = self.tokenizer(caption, padding="max_length", truncation=True, max_length=self.target_len)
        image = Image.open(BytesIO(self.images[index])).convert("RGB")
        if self.transforms is not None:
            image = self.transforms(image)
        return (image, torch.tensor(enc["input_ids"]), torch.tensor(enc["attention_mask"]), caption,
                self.image_ids[index], self.caption_ids[index])

    def __len__(self):
        return self.length


def build_parquet_valid_loader(cfg, name, mode="valid", tokenizer=None, **kwargs):
    tokenizer = tokenizer or load_tokenizer(cfg.model.text_encoder.tag)
    batch_size = (cfg.data.batch_size if mode == "train" else cfg.data.batch_size_val) // ENV.size
    ds = ParquetDataset(cfg=cfg, dataset_name=name, data_path=cfg.data.data_path, tokenizer=tokenizer,
                        transforms=build_transforms(cfg, mode=mode))
    logger.info("Building single parquet {} dataset name: {}.".format(mode, name))
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=ENV.size, rank=ENV.rank, shuffle=False)
    return torch.utils.data.DataLoader(ds, sampler=sampler, batch_size=batch_size, num_workers=cfg.data.num_workers, pin_memory=True,
                                       drop_last=False)
