"""Drop-in `simseg` package: the reference's Python API surface (registries, CLIPModel, config, dist helpers) re-provided
from scratch on top of the MI355X-native engine in `simseg_amd` (HIP kernels behind include/simseg_hip.h).
Only the hot path of SURVEY.md section 8 is provided; everything else of the reference is out of scope."""
__version__ = "0.1.0+mi355x"
__all__ = ["__version__"]
