from .avoiding_sim import Avoiding_Sim  # noqa: F401
from .base_sim import BaseSim  # noqa: F401
