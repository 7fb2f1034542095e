from .cem_sampler import CEMSampler
from .gaussian_sampler import GaussianCEMSampler
from .correlated_noise import CorrelatedNoiseSampler
