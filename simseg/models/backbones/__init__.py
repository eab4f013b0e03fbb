from .builder import BACKBONE
from .mml import *
