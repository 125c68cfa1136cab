from .avoiding import ObstacleAvoidanceVecEnv  # noqa: F401
