from .builder import LOSS
from .mml_loss import *
