from .synthetic_env import SyntheticPushEnv
from .simulator import SyntheticAgent, Sim
