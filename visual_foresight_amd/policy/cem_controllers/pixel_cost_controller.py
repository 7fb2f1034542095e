"""CEM controller scoring rollouts by the expected distance of designated pixels to goals.

API-compatible restatement of the reference's
``visual_mpc/policy/cem_controllers/pixel_cost_controller.py`` (ctor :20-50, defaults :52-69,
``evaluate_rollouts`` :76-133, ``_eval_pixel_cost`` :135-166, ``_expected_distance`` :168-187,
``_get_distancegrid`` :189-197, ``_switch_on_pix`` :206-215, ``act`` :217-233).

The one device crossing of the planner is ``self.predictor(context, {'actions': actions})``
(reference :83).  The default ``predictor_class`` here is the MI355X-native
``HipVPredEvaluation``; when the predictor offers the fused ``score`` entry point the
predicted videos never leave the GPU - the designated-pixel distributions are reduced to
per-sample costs on the device and only ``scores[M]`` comes back (sharded over ranks and
all-gathered when ``torch.distributed`` is initialised).  Any other predictor with the
``VPredEvaluation`` duck-type (``__call__`` returning ``predicted_frames`` /
``predicted_pixel_distributions``) is scored on the host exactly as the reference does.
"""
import numpy as np

from .cem_base_controller import CEMBaseController


def _default_predictor_class(ncam=1):
    """``HipVPredEvaluation`` rolls any number of views (``ncam`` reaches it through the predictor hparams)."""
    from visual_foresight_amd.video_prediction.hip_predictor import HipVPredEvaluation
    return HipVPredEvaluation


class PixelCostController(CEMBaseController):
    def __init__(self, ag_params, policyparams, gpu_id, ngpu):
        """
        :param ag_params: agent parameter dict (needs adim, sdim, image_height, image_width)
        :param policyparams: policy parameter dict (overrides of the HParams defaults)
        :param gpu_id: first GPU to use
        :param ngpu: number of GPUs to use
        """
        CEMBaseController.__init__(self, ag_params, policyparams)
        predictor_hparams = {
            'designated_pixel_count': self._hp.designated_pixel_count,
            'run_batch_size': min(self._hp.vpred_batch_size, self._hp.num_samples),
        }
        predictor_class = self._hp.predictor_class
        if predictor_class is None:
            predictor_class = _default_predictor_class(ag_params.get('ncam', 1))
        if getattr(predictor_class, 'wants_agent_params', False):
            # the HIP predictor is shape-specialised at construction (no checkpoint json to read
            # adim/sdim/size/T from), so it is told what the controller will ask of it
            predictor_hparams.update(
                adim=self._adim, sdim=self._sdim,
                image_height=ag_params['image_height'], image_width=ag_params['image_width'],
                ncam=ag_params.get('ncam', 1),
                sequence_length=self._plan_horizon() + predictor_class.n_context_default)
        self.predictor = predictor_class(self._hp.model_path, predictor_hparams,
                                         n_gpus=ngpu, first_gpu=gpu_id)
        self.predictor.restore()

        self._net_context = self.predictor.n_context
        if self._hp.start_planning < self._net_context - 1:
            self._hp.start_planning = self._net_context - 1

        self._n_desig = self._hp.designated_pixel_count
        self._img_height, self._img_width = [ag_params['image_height'], ag_params['image_width']]
        # the reference hard-codes 1 here (:43); multi-view predictors announce their view count
        self._n_cam = getattr(self.predictor, 'n_cam', 1)

        self._desig_pix = None
        self._goal_pix = None
        self._images = None
        if self._hp.predictor_propagation:
            self._chosen_distrib = None     # distributions of the executed plan

    def _plan_horizon(self):
        """Number of predicted steps the sampler will produce (nactions * repeat)."""
        hp = self._hp
        return hp.nactions * hp.get('repeat', 1)

    def _default_hparams(self):
        defaults = [
            ('predictor_class', None),      # None -> HipVPredEvaluation
            ('model_path', ''),
            ('vpred_batch_size', 200),
            ('designated_pixel_count', 1),
            ('verbose_img_height', 128),
            ('predictor_propagation', False),
            ('only_take_first_view', False),
            ('state_append', None),
            ('finalweight', 10.),
        ]
        params = super(PixelCostController, self)._default_hparams()
        for name, value in defaults:
            params.add_hparam(name, value)
        return params

    def reset(self):
        super(PixelCostController, self).reset()
        if self._hp.predictor_propagation:
            self._chosen_distrib = None

    # ------------------------------------------------------------------ rollout scoring
    def evaluate_rollouts(self, actions, cem_itr):
        context = {
            "context_frames": self._images,
            "context_actions": self._sampler.chosen_actions,
            "context_pixel_distributions": self._make_input_distrib(cem_itr),
            "context_states": self._state,
        }
        if hasattr(self.predictor, 'score'):
            weights = self._task_weights()
            on_device = weights is not None and getattr(self.predictor, 'supports_task_weights', False)
            kw = {'task_weights': weights} if on_device else {}
            scores, scores_per_task = self.predictor.score(
                context, {'actions': actions}, goal_pix=self._goal_pix,
                finalweight=self._hp.finalweight,
                only_take_first_view=self._hp.only_take_first_view, **kw)
            if weights is not None and not on_device:
                scores = np.sum(scores_per_task * np.asarray(weights).reshape(1, -1), axis=1)
            self._log_task_scores(scores, scores_per_task)
            if self._hp.predictor_propagation and cem_itr == self._hp.iterations - 1:
                bestind = scores.argsort()[0]
                self._chosen_distrib = self.predictor.fetch_pixel_distributions(bestind)
        else:
            prediction = self.predictor(context, {'actions': actions})
            gen_images = prediction['predicted_frames']
            gen_distrib = prediction['predicted_pixel_distributions']
            scores = self._eval_pixel_cost(cem_itr, gen_distrib, gen_images)

        if self._verbose_condition(cem_itr):
            self._visualize(cem_itr, scores)
        return scores

    def _task_weights(self):
        """Per-(camera, designated pixel) score weights, or None for the reference's plain mean."""
        return None

    def _visualize(self, cem_itr, scores):
        """Hook for plan visualisation (the reference renders an HTML/GIF page, :88-131).

        Rendering is debug tooling outside the planner hot path; subclasses may override this
        and pull videos with ``self.predictor(context, ...)``.
        """
        self._logger.log('best scores itr {}: {}'.format(cem_itr, np.sort(scores)[:10]))

    def _log_task_scores(self, scores, scores_per_task):
        bestind = scores.argsort()[0]
        for icam in range(self._n_cam):
            for p in range(self._n_desig):
                col = p + icam * self._n_desig
                if col < scores_per_task.shape[1]:
                    self._logger.log('best flow score of task {} cam{}  :{}'.format(
                        p, icam, np.min(scores_per_task[:, col])))
                    self._logger.log('flow score of best traj for task{} cam{} :{}'.format(
                        p, icam, scores_per_task[bestind, col]))

    def _eval_pixel_cost(self, cem_itr, gen_distrib, gen_images):
        """Host scoring of materialised distributions ``[M, T, ncam, H, W, ndesig]``."""
        per_task = []
        for icam in range(self._n_cam):
            for p in range(self._n_desig):
                grid = self._get_distancegrid(self._goal_pix[icam, p])
                per_task.append(self._expected_distance(icam, p, gen_distrib[:, :, icam, :, :, p], grid,
                                                        normalize=True))
        scores_per_task = np.stack(per_task, axis=1)
        if self._hp.only_take_first_view:
            scores_per_task = scores_per_task[:, 0][:, None]
        weights = self._task_weights()
        if weights is not None:
            scores = np.sum(scores_per_task * np.asarray(weights).reshape(1, -1), axis=1)
        else:
            scores = np.mean(scores_per_task, axis=1)
        self._log_task_scores(scores, scores_per_task)

        if self._hp.predictor_propagation and cem_itr == self._hp.iterations - 1:
            # propagate the distributions of the plan that will actually be executed
            self._chosen_distrib = gen_distrib[scores.argsort()[0]]
        return scores

    def _expected_distance(self, icam, idesig, gen_distrib, distance_grid, normalize=True):
        """score_b = sum_t w_t * E_{p_bt}[distance] / sum_t w_t, w = (1, ..., 1, finalweight).

        :param gen_distrib: ``[batch, t, r, c]``
        :param distance_grid: ``[r, c]``
        """
        assert len(gen_distrib.shape) == 4
        t_mult = np.ones([self.predictor.sequence_length - self._net_context])
        t_mult[-1] = self._hp.finalweight

        p = gen_distrib.copy()
        if normalize:
            p /= np.sum(np.sum(p, axis=2), 2)[:, :, None, None]
        p *= distance_grid[None, None]
        per_step = np.sum(np.sum(p, axis=2), 2)
        per_step *= t_mult[None]
        return np.sum(per_step, axis=1) / np.sum(t_mult)

    def _get_distancegrid(self, goal_pix):
        """D[i, j] = || goal_pix - (i, j) ||_2 in (row, col) order, float64."""
        rows = np.arange(self._img_height, dtype=np.float64)[:, None] - np.float64(goal_pix[0])
        cols = np.arange(self._img_width, dtype=np.float64)[None, :] - np.float64(goal_pix[1])
        self._logger.log('making distance grid with goal_pix', goal_pix)
        return np.sqrt(rows * rows + cols * cols)

    # ------------------------------------------------------------------ designated pixels
    def _make_input_distrib(self, itr):
        if self._hp.predictor_propagation and self._chosen_distrib is not None:
            # let the predictor's own flow carry the distribution forward, no correction
            return self._chosen_distrib[-self._net_context:]
        return self._switch_on_pix(self._desig_pix)

    def _switch_on_pix(self, desig):
        one_hot = np.zeros((self._net_context, self._n_cam, self._img_height, self._img_width,
                            self._n_desig), dtype=np.float32)
        hi = np.array([self._img_height, self._img_width]).reshape((1, 2)) - 1
        desig = np.clip(desig, np.zeros((1, 2)), hi).astype(int)
        for icam in range(self._n_cam):
            for p in range(self._n_desig):
                one_hot[:, icam, desig[icam, p, 0], desig[icam, p, 1], p] = 1.
                self._logger.log('using desig pix', desig[icam, p, 0], desig[icam, p, 1])
        return one_hot

    def act(self, t=None, i_tr=None, desig_pix=None, goal_pix=None, images=None, state=None,
            verbose_worker=None):
        """
        :param t: the controller's time step
        :param desig_pix: designated pixels, (row, col) in small-image coordinates
        :param goal_pix: goal pixels, same coordinates; both reshapeable to [ncam, ndesig, 2]
        :param images: uint8 history ``[t+1, ncam, H, W, 3]``
        :param state: state history ``[t+1, sdim]``
        """
        self._desig_pix = np.array(desig_pix).reshape((self._n_cam, self._n_desig, 2))
        self._goal_pix = np.array(goal_pix).reshape((self._n_cam, self._n_desig, 2))
        self._images = images
        self._verbose_worker = verbose_worker
        return super(PixelCostController, self).act(t, i_tr, state)
