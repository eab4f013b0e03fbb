from .backbones import *
from .components import *
from .criteria import *
from .pipelines import *
from .backbones import BACKBONE
from .criteria.losses import LOSS
from .pipelines import PIPELINE
