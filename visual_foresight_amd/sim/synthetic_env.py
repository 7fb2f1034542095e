"""A MuJoCo-free stand-in environment for driving the planner end to end.

The reference's rollouts need MuJoCo or a robot (``visual_mpc/envs``; out of scope, SURVEY.md
section 2 rows 18-19).  This toy environment only honours the *contract* the agent/policy pair
relies on (``visual_mpc/agent/general_agent.py:174-228``): ``reset() -> obs``, ``step(action) ->
obs`` with ``obs = {'images': uint8 [ncam, H, W, 3], 'state': [sdim]}``, designated/goal pixels in
(row, col) of the small image.  A bright square (the "object") sits on a fixed random texture; a
second square (the "gripper") moves with the first two action dimensions and drags the object
along while they overlap.
"""
import numpy as np


class SyntheticPushEnv(object):
    def __init__(self, height=64, width=64, ncam=1, adim=4, sdim=5, seed=0, pixels_per_unit=40.0):
        self.height, self.width, self.ncam, self.adim, self.sdim = height, width, ncam, adim, sdim
        self._scale = pixels_per_unit
        self._rs = np.random.RandomState(seed)
        self._texture = self._rs.randint(40, 120, (ncam, height, width, 3)).astype(np.uint8)
        self.reset()

    def reset(self):
        h, w = self.height, self.width
        self._obj = np.array([h * 0.5, w * 0.5])
        self._grip = np.array([h * 0.5, w * 0.3])
        self.goal = np.array([h * 0.25, w * 0.75])
        self._t = 0
        return self._obs()

    def step(self, action):
        action = np.asarray(action, dtype=np.float64)
        delta = action[:2] * self._scale
        near = np.all(np.abs(self._grip - self._obj) < 5.0)
        self._grip = np.clip(self._grip + delta, 2, [self.height - 3, self.width - 3])
        if near:
            self._obj = np.clip(self._obj + delta, 2, [self.height - 3, self.width - 3])
        self._t += 1
        return self._obs()

    def _obs(self):
        img = self._texture.copy()
        for c in range(self.ncam):
            for (r, col), colour in ((self._obj, (230, 60, 60)), (self._grip, (60, 60, 230))):
                r, col = int(round(r)), int(round(col)) + 3 * c      # views differ by a small parallax
                img[c, max(r - 2, 0):r + 3, max(col - 2, 0):col + 3] = colour
        state = np.zeros(self.sdim)
        state[:2] = self._grip / [self.height, self.width]
        return {'images': img, 'state': state}

    def get_desig_pix(self):
        """[ncam, 1, 2] (row, col) of the object centre."""
        return np.array([[[int(round(self._obj[0])), int(round(self._obj[1])) + 3 * c]] for c in range(self.ncam)])

    def get_goal_pix(self):
        return np.array([[[int(round(self.goal[0])), int(round(self.goal[1])) + 3 * c]] for c in range(self.ncam)])

    def goal_distance(self):
        return float(np.linalg.norm(self._obj - self.goal))
