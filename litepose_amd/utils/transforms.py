"""Drop-in for the ``utils.transforms`` functions on the hot path
(reference lib/utils/transforms.py:155-176,195-202).  cv2-free."""
import ctypes as C

import numpy as np
import torch

from .. import _native as nv


def get_multi_scale_size(image, input_size, current_scale, min_scale):
    """transforms.py:155-176 (host arithmetic only).  ``image``: HxWxC array or (h, w)."""
    h, w = (image.shape[0], image.shape[1]) if hasattr(image, 'shape') else image
    center = np.array([int(w / 2.0 + 0.5), int(h / 2.0 + 0.5)])
    min_input_size = int((min_scale * input_size + 63) // 64 * 64)
    if w < h:
        w_resized = int(min_input_size * current_scale / min_scale)
        h_resized = int(int((min_input_size / w * h + 63) // 64 * 64) * current_scale / min_scale)
        scale_w = w / 200.0
        scale_h = h_resized / w_resized * w / 200.0
    else:
        h_resized = int(min_input_size * current_scale / min_scale)
        w_resized = int(int((min_input_size / h * w + 63) // 64 * 64) * current_scale / min_scale)
        scale_h = h / 200.0
        scale_w = w_resized / h_resized * h / 200.0
    return (w_resized, h_resized), center, np.array([scale_w, scale_h])


def final_preds_device(ans, count, center, scale, heatmap_size):
    """In place on device records: ans [N,pcap,J,3+T], count [N] (lp_final_preds)."""
    N, pcap, J, D = ans.shape
    c = (C.c_double * 2)(float(center[0]), float(center[1]))
    s = (C.c_double * 2)(float(scale[0]), float(scale[1]))
    nv.check(nv.lib().lp_final_preds(nv.dptr(ans), nv.dptr(count), N, pcap, J, D - 3, c, s,
                                     int(heatmap_size[0]), int(heatmap_size[1]), nv.stream_ptr()),
             'lp_final_preds')
    return ans


def get_final_preds(grouped_joints, center, scale, heatmap_size):
    """transforms.py:195-202: list of per-person [J, 3+T] arrays in image coordinates."""
    persons = grouped_joints[0]
    if len(persons) == 0:
        return []
    a = torch.as_tensor(np.ascontiguousarray(persons, dtype=np.float32)).cuda()[None].contiguous()
    cnt = torch.tensor([a.shape[1]], dtype=torch.int32, device=a.device)
    final_preds_device(a, cnt, center, scale, heatmap_size)
    out = a[0].cpu().numpy()
    return [out[p] for p in range(out.shape[0])]
