from .builder import PIPELINE
from . import clip
