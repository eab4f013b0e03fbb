from .normalization import *
from .pooling import *
from .projection import *
