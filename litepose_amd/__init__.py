"""litepose_amd -- MI355X-native LitePose inference hot path.

Host-side mirror of the reference surface that ``valid.py`` uses (valid.py:29-47):
    models.pose_mobilenet.get_pose_net      -> litepose_amd.models.pose_mobilenet
    core.inference.get_multi_stage_outputs  -> litepose_amd.core.inference
    core.inference.aggregate_results
    core.group.HeatmapParser                -> litepose_amd.core.group
    utils.transforms.get_final_preds ...    -> litepose_amd.utils.transforms
plus the batched fast path ``litepose_amd.engine.PoseEngine.infer_batch``.
All math runs in hand-written HIP kernels behind the C ABI of include/litepose_amd.h
(litepose_amd/lib/liblitepose_amd.so); there is no CPU fallback.
"""
__version__ = '0.1.0'
