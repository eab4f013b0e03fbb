from .config import *
from .hooks import *
from .initial import *
