"""The Sim protocol the evaluation entry point talks to.

The reference instantiates ``cfg.simulation`` through Hydra and calls ``env_sim.test_agent(agent)`` once (run.py:61-62); the
protocol class there (simulation/base_sim.py:8-30) fixes the constructor keywords ``seed, device, render, n_cores, if_vision``
and the attributes the task sims read back.  The batched sims of this package keep that contract; ``n_cores`` is accepted for
config compatibility only (rollouts are lanes of one GPU batch, shards of it under torch.distributed - not OS processes).
"""
from __future__ import annotations

import abc
import os


class BaseSim(abc.ABC):
    env_name = "BaseEnvironment"

    def __init__(self, seed, device, render=True, n_cores=1, if_vision=False):
        settings = {"seed": int(seed), "device": str(device), "render": bool(render), "n_cores": int(n_cores), "if_vision": bool(if_vision)}
        if settings["if_vision"]:
            raise NotImplementedError("camera observations are outside the batched state-observation rollout path")
        for key, value in settings.items():
            setattr(self, key, value)
        self.working_dir = os.getcwd()

    @abc.abstractmethod
    def test_agent(self, agent):
        """Roll ``agent`` out on the task and return the task's result tables / metrics (see the subclasses)."""
