from simseg.utils import Registry

PIPELINE = Registry("pipeline")
