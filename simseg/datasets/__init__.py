from .builder import DATALOADER
from .clip import clip_dataset  # noqa: F401
from .seg import seg_dataset  # noqa: F401
