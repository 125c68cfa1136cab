"""Sim protocol (counterpart of simulation/base_sim.py:8-30): selected by Hydra ``simulation._target_``."""
import abc
import os


class BaseSim(abc.ABC):
    def __init__(self, seed: int, device: str, render: bool = True, n_cores: int = 1, if_vision: bool = False):
        self.seed = seed
        self.device = device
        self.render = render
        self.n_cores = n_cores
        self.if_vision = if_vision
        self.working_dir = os.getcwd()
        self.env_name = "BaseEnvironment"

    @abc.abstractmethod
    def test_agent(self, agent):
        pass
