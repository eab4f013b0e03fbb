from simseg.utils import Registry

DATALOADER = Registry("dataloader")
