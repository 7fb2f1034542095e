"""GPU: a planning call repeated must reproduce itself bit for bit.

The persistent rollout hands items to whichever workgroup is free, so the assignment of tiles to workgroups differs
from call to call; a race between workgroups (an item run twice, a consumer released early) shows up as a few samples
whose cost sums change between repetitions - which is how a scheduler-loop experiment of round 3 was caught
(profiles/r03_tile_plan_sweep.txt).  Sizes: enough samples to keep every workgroup slot busy, 2-4 designated pixels
(the fused decoder top with its cross-tile LayerNorm wait), both architectures."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stress():
    spec = importlib.util.spec_from_file_location('stress_repeat', os.path.join(REPO, 'tools', 'stress_repeat.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('arch,H,W,T,M,nd,prec,ncam', [
    ('cdna', 64, 64, 4, 120, 2, 'fp32', 1),
    ('cdna', 64, 64, 4, 200, 4, 'fp32', 1),
    ('cdna', 64, 64, 4, 100, 2, 'fp32', 2),
    ('cdna', 64, 64, 4, 120, 2, 'bf16x6', 1),
    ('savp', 64, 64, 3, 150, 2, 'fp32', 1),
    ('cdna', 64, 64, 4, 125, 2, 'fp32', 1),         # not a multiple of 8: the tile-by-tile tail of the XCD dealing
    ('cdna', 64, 64, 6, 25, 1, 'fp32', 1),
    ('cdna', 48, 64, 4, 61, 2, 'fp32', 2),
])
def test_repeated_planning_calls_are_bit_identical(arch, H, W, T, M, nd, prec, ncam):
    assert _stress().run(arch, H, W, T, M, nd, prec, seed=11, reps=8, ncam=ncam) == 0
