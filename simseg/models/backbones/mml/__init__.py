from .huggingface_builder import *
from .vit_builder import *
