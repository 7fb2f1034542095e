"""CEM controller that re-localises the designated pixels by registering the current frame.

Behavioural restatement of the reference's
``visual_mpc/policy/cem_controllers/register_gtruth_controller.py`` (``Register_Gtruth_Controller``
:10, defaults :28-40, ``register_gtruth`` :54-111, ``act`` :175-195).  That file is mid-refactor in
the reference snapshot - it imports three modules that do not exist and calls a parent method that
was removed - so it documents intent rather than runnable behaviour; this class implements that
intent on top of ``PixelCostController``:

* ``ndesig = ntask * len(register_gtruth)`` designated pixels per camera: every task is tracked
  once per registration target ('start' frame and/or 'goal' image), ``:25``;
* at the first CEM iteration of every planning call the plug-in warper registers the current
  frame to the start frame / goal image, the tracked pixels replace the designated pixels and the
  warp errors give the trade-off weights ``plan_stat['tradeoff']``, ``:46-50,88-94``;
* goal pixels are tiled across registrations, ``:179-181``.

The reference stores the trade-off but its parent scores with a plain mean over tasks
(``pixel_cost_controller.py:153``); ``trade_off_reg=True`` (the reference's commented-out
hyper-parameter, ``:32``) applies the weights, ``False`` keeps the plain mean.

With a predictor that offers ``register()`` (``HipVPredEvaluation``) the arithmetic of
``get_warp_err`` - bilinear warp of the current frame by the registration network's flow field,
window median of the warp points, photometric warp error - runs on the GPU (``vf_register``,
``include/vf_hip.h``) and the trade-off weights are applied inside the fused score kernel; the
plug-in only has to supply the flow field.  Any other predictor takes the NumPy path of
``registration.py`` on the plug-in's own ``(warped, flow, warp_pts)``.
"""
import numpy as np

from .pixel_cost_controller import PixelCostController
from .registration import get_warp_err, tradeoff_weights


class RegisterGtruthController(PixelCostController):
    def __init__(self, ag_params, policyparams, gpu_id, ngpu):
        super(RegisterGtruthController, self).__init__(ag_params, policyparams, gpu_id, ngpu)
        nreg = len(self._hp.register_gtruth)
        if nreg == 0 or self._n_desig % nreg:
            raise ValueError('designated_pixel_count must be a multiple of len(register_gtruth)')
        self.ntask = self._n_desig // nreg
        self.goal_image_warper = self._hp.registration_warper
        self.reg_tradeoff = np.ones([self._n_cam, self._n_desig]) / self._n_cam / self._n_desig
        self.start_image = None

    def _default_hparams(self):
        params = super(RegisterGtruthController, self)._default_hparams()
        params.add_hparam('register_gtruth', ['start', 'goal'])
        params.add_hparam('register_region', False)
        params.add_hparam('trade_off_reg', False)
        params.add_hparam('registration_warper', None)    # callable, see registration.py
        params.add_hparam('registration_on_device', True)  # use predictor.register() when it exists
        return params

    # ------------------------------------------------------------------ registration
    def register_gtruth(self, start_image, current_image):
        """-> (tracked desig pixels [ncam, ndesig, 2], trade-off [ncam, ndesig]) for this frame."""
        if self.goal_image_warper is None:
            raise ValueError("RegisterGtruthController needs the 'registration_warper' hyper-parameter")
        regs = self._hp.register_gtruth
        H = start_image.shape[1]
        if hasattr(self.predictor, 'register') and self._hp.registration_on_device:
            region = (5 if H >= 96 else 2) if self._hp.register_region else 0
            per_reg = []
            if 'start' in regs:
                _, flow, _ = self.goal_image_warper(current_image, start_image)
                per_reg.append(self.predictor.register(current_image, start_image, flow, self.desig_pix_t0,
                                                       region=region, clip_sub=1))
            if 'goal' in regs:
                _, flow, _ = self.goal_image_warper(current_image, self.goal_image)
                per_reg.append(self.predictor.register(current_image, self.goal_image, flow, self.goal_pix_sel,
                                                       region=region, clip_sub=0))
            desig = [np.stack([d[icam] for d, _ in per_reg], axis=1) for icam in range(self._n_cam)]   # [ntask, nreg, 2]
            errs = [np.stack([e[icam] for _, e in per_reg], axis=1) for icam in range(self._n_cam)]    # [ntask, nreg]
        else:
            warped_start = start_pts = warped_goal = goal_pts = None
            if 'start' in regs:
                warped_start, _, start_pts = self.goal_image_warper(current_image, start_image)
            if 'goal' in regs:
                warped_goal, _, goal_pts = self.goal_image_warper(current_image, self.goal_image)
            errs, desig = [], []
            for icam in range(self._n_cam):
                e, d = get_warp_err(icam, self.desig_pix_t0[icam], self.goal_pix_sel[icam], start_image,
                                    self.goal_image, start_pts, goal_pts, warped_start, warped_goal,
                                    register_gtruth=regs, register_region=self._hp.register_region)
                errs.append(e)
                desig.append(d)
        warperrs = np.stack(errs, 0)                                    # [ncam, ntask, nreg]
        tradeoff = tradeoff_weights(warperrs).reshape(self._n_cam, self._n_desig)
        self.plan_stat['tradeoff'] = tradeoff
        self.plan_stat['warperrs'] = warperrs.reshape(self._n_cam, self._n_desig)
        return np.stack(desig, 0).reshape(self._n_cam, self._n_desig, 2), tradeoff

    def evaluate_rollouts(self, actions, cem_itr):
        if self._hp.register_gtruth and cem_itr == 0:
            current = self._images[-1].astype(np.float32) / 255.
            self._desig_pix, self.reg_tradeoff = self.register_gtruth(self.start_image, current)
        return super(RegisterGtruthController, self).evaluate_rollouts(actions, cem_itr)

    def _task_weights(self):
        return self.reg_tradeoff if self._hp.trade_off_reg else None

    def act(self, goal_image=None, t=None, i_tr=None, desig_pix=None, goal_pix=None, images=None, state=None,
            verbose_worker=None):
        """``goal_image`` [.., ncam, H, W, 3] float in [0,1] (the last entry is used, ``:186``)."""
        nreg = len(self._hp.register_gtruth)
        self.goal_pix_sel = np.array(goal_pix).reshape((self._n_cam, self.ntask, 2))
        goal_tiled = np.tile(self.goal_pix_sel[:, :, None, :], [1, 1, nreg, 1])
        self.goal_image = np.asarray(goal_image)[-1]
        if t == 0:
            self.desig_pix_t0 = np.array(desig_pix).reshape((self._n_cam, self.ntask, 2))
            self.start_image = np.asarray(images)[0].astype(np.float32) / 255.
        desig_tiled = np.tile(self.desig_pix_t0[:, :, None, :], [1, 1, nreg, 1])
        return super(RegisterGtruthController, self).act(
            t=t, i_tr=i_tr, desig_pix=desig_tiled.reshape(self._n_cam, self._n_desig, 2),
            goal_pix=goal_tiled.reshape(self._n_cam, self._n_desig, 2), images=images, state=state,
            verbose_worker=verbose_worker)
