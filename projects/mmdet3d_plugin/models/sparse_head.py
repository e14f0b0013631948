"""Head dispatcher (registered name ``SparseHead``, reference models/sparse_head.py:14-110): the HiP-AD
configs run a single unified decoder (``task_config = dict(with_onedecoder=True)``); the legacy
per-task heads of the reference are not part of the hot path."""
from hipad_amd.compat import HEADS, BaseModule, build_from_cfg

__all__ = ["SparseHead"]


@HEADS.register_module()
class SparseHead(BaseModule):
    def __init__(self, task_config: dict, init_cfg=None, det_head=None, map_head=None, motion_plan_head=None,
                 onedecoder_head=None, evaluate_bench2dive=False, **kwargs):
        super().__init__(init_cfg)
        self.task_config = task_config
        for legacy in ("with_det", "with_map", "with_motion_plan"):
            if task_config.get(legacy, False):
                raise NotImplementedError(f"task_config[{legacy!r}]: the per-task heads are outside the hot path")
        if not task_config.get("with_onedecoder", False):
            raise ValueError("SparseHead needs task_config['with_onedecoder'] = True")
        self.onedecoder_head = build_from_cfg(onedecoder_head, HEADS)
        self.evaluate_bench2dive = evaluate_bench2dive

    def init_weights(self):
        self.onedecoder_head.init_weights()

    def forward(self, img, feature_maps, metas: dict):
        return self.onedecoder_head(img, feature_maps, metas)

    def loss(self, model_outs, data):
        return self.onedecoder_head.loss(*model_outs, data)

    def post_process(self, model_outs, data):
        """One result dict per sample with the entries of every selected task (reference sparse_head.py:108-154).
        (The reference builds ``[dict()] * batch_size`` -- one dict shared by all samples; a dict per sample here.)"""
        det_output, map_output, ego_output, plan_output, motion_output, _ = model_outs
        per_task = dict(zip(("det", "map", "ego", "plan", "motion"),
                            self.onedecoder_head.post_process(det_output, map_output, ego_output, plan_output,
                                                              motion_output, data)))
        first = next(per_task[t] for t in self.onedecoder_head.task_select if per_task.get(t) is not None)
        results = [dict() for _ in range(len(first))]
        for i, res in enumerate(results):
            for task in self.onedecoder_head.task_select:
                if per_task.get(task) is not None:
                    res.update(per_task[task][i])
        # evaluate_bench2dive (reference sparse_head.py:156-206: the open-loop STP3 planning metric added to every
        # result as 'metric_results') is evaluation tooling outside the hot path: results come back without it
        return results
