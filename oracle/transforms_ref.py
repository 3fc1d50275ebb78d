"""Coordinate back-projection oracle.  TEST INFRASTRUCTURE.

NumPy restatement of
  * /root/reference/lib/utils/transforms.py:155-176  get_multi_scale_size
  * /root/reference/lib/utils/transforms.py:59-93    get_affine_transform (rot = 0)
  * /root/reference/lib/utils/transforms.py:50-56,96-99 transform_preds / affine_transform
  * /root/reference/lib/utils/transforms.py:195-202  get_final_preds

PARITY UNPINNED for the general (non-square) case: the reference obtains the
2x3 matrix from ``cv2.getAffineTransform`` and cv2 is absent from this image, so
the closed form below (rot = 0 makes the map a uniform scale + translation) could
only be checked against its own derivation.  For every BASELINE.json config
(square input of side INPUT_SIZE) the map is the identity,
which the tests assert.
"""
import numpy as np


def get_multi_scale_size(image_hw, input_size, current_scale, min_scale):
    h, w = image_hw
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


def inverse_affine(center, scale, output_size):
    """2x3 float64 matrix mapping heatmap coords -> image coords (inv=1, rot=0).

    src triangle (image): c, c+(0,-sw/2), c+(sw/2... ) ; dst triangle (heatmap):
    (dw/2,dh/2), (dw/2,dh/2-dw/2), third by 90-degree rule.  Both are congruent up
    to the scale sw/dw, so T = [[s,0,cx-s*dw/2],[0,s,cy-s*dh/2]]."""
    scale = np.asarray(scale, dtype=np.float64)
    sw = scale[0] * 200.0
    dw, dh = float(output_size[0]), float(output_size[1])
    s = sw / dw
    cx, cy = float(center[0]), float(center[1])
    return np.array([[s, 0.0, cx - s * dw * 0.5], [0.0, s, cy - s * dh * 0.5]])


def transform_preds(coords, center, scale, output_size):
    target = coords.copy()
    t = inverse_affine(center, scale, output_size)
    for p in range(coords.shape[0]):
        v = np.array([coords[p, 0], coords[p, 1], 1.0])
        target[p, 0:2] = np.dot(t, v)[:2]
    return target


def get_final_preds(grouped_joints, center, scale, heatmap_size):
    return [transform_preds(person, center, scale, heatmap_size) for person in grouped_joints[0]]
