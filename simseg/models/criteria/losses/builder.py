from simseg.utils import Registry

LOSS = Registry("loss")
