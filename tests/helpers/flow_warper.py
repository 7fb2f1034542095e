"""A deterministic stand-in for the reference's (absent) registration network.

``make_flow_warper`` returns a plug-in with the warper signature of
``/root/reference/visual_mpc/policy/cem_controllers/register_gtruth_controller.py:64-66``
(``warper(current, reference) -> warped, flow, warp_pts``).  The flow field is a smooth
function of the pixel grid and of the mean intensity difference of the two images, so it
changes when the images change; ``warped`` and ``warp_pts`` come from the oracle's bilinear
warp of that flow, so the host path (which consumes them) and the device path (which only
takes the flow) are fed consistently.
"""
import numpy as np

from oracle.registration import bilinear_warp_loops


def synthetic_flow(current, reference, scale=2.0):
    cur = np.asarray(current, dtype=np.float64)
    ref = np.asarray(reference, dtype=np.float64)
    ncam, H, W = cur.shape[:3]
    rows, cols = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    flow = np.zeros((ncam, H, W, 2), np.float32)
    for c in range(ncam):
        bias = float(cur[c].mean() - ref[c].mean())
        flow[c, :, :, 0] = scale * np.sin(rows / 7.0 + c) + 0.75 + 3.0 * bias
        flow[c, :, :, 1] = scale * np.cos(cols / 5.0 - c) - 0.5 - 3.0 * bias
    return flow


def make_flow_warper(scale=2.0):
    def warper(current, reference):
        flow = synthetic_flow(current, reference, scale)
        warped, pts = bilinear_warp_loops(current, flow)
        return warped, flow, pts
    return warper


def registration_inputs(seed, ncam, H, W, scale=2.5):
    """Seeded images and the fake warper's outputs, shared by ``tools/make_golden.py`` (which feeds
    them to the reference's real ``get_warp_err``) and the tests that must reproduce the fixture."""
    rs = np.random.RandomState(seed)
    start, goal, cur = (rs.uniform(0, 1, (ncam, H, W, 3)).astype(np.float32) for _ in range(3))
    warper = make_flow_warper(scale)
    ws, _, ps = warper(cur, start)
    wg, _, pg = warper(cur, goal)
    return start, goal, cur, ws, ps, wg, pg
