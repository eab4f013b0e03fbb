from .checkpoint import get_dist_state_dict

__all__ = ["get_dist_state_dict"]
