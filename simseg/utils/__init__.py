from . import logger
from .collections import AttrDict
from .context import *
from .dist import *
from .misc import *
from .registry import *
