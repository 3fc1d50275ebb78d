"""Drop-in for the ``utils.transforms`` functions on the hot path and next to it
(reference lib/utils/transforms.py:59-99,155-192,195-202; valid.py:178-186).  cv2-free: the
image warp + ToTensor + Normalize run as one HIP kernel (``lp_preprocess``)."""
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


# ---------------------------------------------------------------- pre-processing
IMAGENET_MEAN = (0.485, 0.456, 0.406)       # valid.py:181-184
IMAGENET_STD = (0.229, 0.224, 0.225)


def _get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def _solve_affine(src, dst):
    """What cv2.getAffineTransform returns: the fp64 2x3 matrix mapping three float32 points."""
    x0, y0, x1, y1, x2, y2 = [float(v) for v in src.reshape(-1)]
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    if det == 0:
        raise ValueError('degenerate point triple')
    rows = []
    for k in range(2):
        u0, u1, u2 = float(dst[0, k]), float(dst[1, k]), float(dst[2, k])
        a = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        b = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        c = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
        rows.append([a, b, c])
    return np.array(rows, np.float64)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    """transforms.py:59-93 (point construction in float32 as there; the 3-point solve is Cramer's
    rule in fp64 instead of cv2.getAffineTransform)."""
    if not isinstance(scale, np.ndarray) and not isinstance(scale, list):
        scale = np.array([scale, scale])
    scale_tmp = np.asarray(scale, dtype=np.float64) * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = _get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = _third_point(src[0, :], src[1, :])
    dst[2:, :] = _third_point(dst[0, :], dst[1, :])
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def warp_normalize_device(image, trans, size, mean=IMAGENET_MEAN, std=IMAGENET_STD, want_u8=True,
                          want_tensor=True):
    """image: HxWx3 uint8 (NumPy array, host or device tensor) -> (warped uint8 [Hd,Wd,3] or None,
    normalised float32 [3,Hd,Wd] or None), both on the GPU (``lp_preprocess``)."""
    if not isinstance(image, torch.Tensor):
        image = torch.from_numpy(np.ascontiguousarray(image))
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise ValueError('image must be HxWx3 uint8')
    image = image.cuda().contiguous()
    H, W = int(image.shape[0]), int(image.shape[1])
    Wd, Hd = int(size[0]), int(size[1])
    u8 = torch.empty((Hd, Wd, 3), dtype=torch.uint8, device=image.device) if want_u8 else None
    ten = torch.empty((3, Hd, Wd), dtype=torch.float32, device=image.device) if want_tensor else None
    m = (C.c_double * 6)(*[float(v) for v in np.asarray(trans, np.float64).reshape(-1)])
    mean_c = (C.c_float * 3)(*[float(v) for v in mean])
    std_c = (C.c_float * 3)(*[float(v) for v in std])
    nv.check(nv.lib().lp_preprocess(nv.dptr(image), H, W, m, Hd, Wd, mean_c, std_c,
                                    nv.dptr(u8) if u8 is not None else None,
                                    nv.dptr(ten) if ten is not None else None, nv.stream_ptr()),
             'lp_preprocess')
    return u8, ten


def normalize_batch_device(images_u8, out=None, trans=None, size=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """A batch of equally sized uint8 images [N,H,W,3] ON THE DEVICE -> normalised float32 [N,3,Hd,Wd] in ONE launch
    (``lp_preprocess_batch``): ToTensor + Normalize of valid.py:178-186 for every image, after the warp ``trans`` of
    ``resize_align_multi_scale`` if one is given (default: identity at the input size -- images the loader already
    delivers at the network resolution).  ``out`` is written in place when given (a serving loop's staging buffer)."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3 or not images_u8.is_cuda:
        raise ValueError('images must be a [N,H,W,3] uint8 device tensor')
    N, H, W = int(images_u8.shape[0]), int(images_u8.shape[1]), int(images_u8.shape[2])
    Wd, Hd = (int(size[0]), int(size[1])) if size is not None else (W, H)
    if out is None:
        out = torch.empty((N, 3, Hd, Wd), dtype=torch.float32, device=images_u8.device)
    elif tuple(out.shape) != (N, 3, Hd, Wd) or out.dtype != torch.float32:
        raise ValueError('out must be float32 [N,3,Hd,Wd]')
    t = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]) if trans is None else np.asarray(trans, np.float64)
    m = (C.c_double * 6)(*[float(v) for v in t.reshape(-1)])
    mean_c = (C.c_float * 3)(*[float(v) for v in mean])
    std_c = (C.c_float * 3)(*[float(v) for v in std])
    nv.check(nv.lib().lp_preprocess_batch(nv.dptr(images_u8), N, H, W, m, Hd, Wd, mean_c, std_c, None, nv.dptr(out),
                                          nv.stream_ptr()), 'lp_preprocess_batch')
    return out


def resize_align_multi_scale(image, input_size, current_scale, min_scale, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """transforms.py:179-192.  Returns (image_resized, center, scale) like the reference;
    ``image_resized`` is a uint8 [Hd,Wd,3] DEVICE tensor that also carries the normalised network
    input computed by the same launch (``ToTensorNormalize`` below hands it out)."""
    size_resized, center, scale = get_multi_scale_size(image, input_size, current_scale, min_scale)
    trans = get_affine_transform(center, scale, 0, size_resized)
    u8, ten = warp_normalize_device(image, trans, size_resized, mean, std)
    u8._lp_tensor = (ten, tuple(mean), tuple(std))
    return u8, center, scale


class ToTensorNormalize(object):
    """The ``transforms`` object of valid.py:178-186 (ToTensor + Normalize) for device images."""

    def __init__(self, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, image_resized):
        cached = getattr(image_resized, '_lp_tensor', None)
        if cached is not None and cached[1] == self.mean and cached[2] == self.std:
            return cached[0]
        h, w = int(image_resized.shape[0]), int(image_resized.shape[1])
        ident = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
        return warp_normalize_device(image_resized, ident, (w, h), self.mean, self.std, want_u8=False)[1]
