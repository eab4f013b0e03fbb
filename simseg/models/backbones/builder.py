from simseg.utils import Registry

BACKBONE = Registry("backbone")
