"""Pre-processing oracle.  TEST INFRASTRUCTURE.

NumPy restatement of
  * /root/reference/lib/utils/transforms.py:59-99    get_affine_transform / get_dir / get_3rd_point
  * /root/reference/lib/utils/transforms.py:179-192  resize_align_multi_scale (cv2.warpAffine)
  * /root/reference/valid.py:178-186                 torchvision ToTensor + Normalize

PARITY UNPINNED: cv2 (and torchvision) are absent from this image, so the warp is restated from the
published algorithm of OpenCV's ``cv::warpAffine`` for 8-bit images with INTER_LINEAR and a constant
border (modules/imgproc/src/imgwarp.cpp, pinned version unknown -- requirements.txt:4 lists
``opencv-python`` without a version): the 2x3 matrix is inverted in fp64, positions are carried
in 1/1024-pixel fixed point (AB_BITS = 10), reduced to 5 fractional bits (INTER_BITS), and the four
taps are mixed with 15-bit integer weights (INTER_REMAP_COEF_BITS).  ``cv2.getAffineTransform`` solves
the three point pairs (given as float32) in fp64; here that is ``numpy.linalg.solve``.  The only check
possible without cv2 is self-consistency: identity / integer-shift warps must reproduce the image
exactly and the tests compare against a float bilinear resampling within 1 grey level.
"""
import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
COEF_BITS = 15


def get_dir(src_point, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return [src_point[0] * cs - src_point[1] * sn, src_point[0] * sn + src_point[1] * cs]


def get_3rd_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def solve_affine(src, dst):
    """cv2.getAffineTransform: the 2x3 fp64 matrix with M @ [x, y, 1] = dst for three point pairs."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i] = [src[i, 0], src[i, 1], 1, 0, 0, 0]
        a[2 * i + 1] = [0, 0, 0, src[i, 0], src[i, 1], 1]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, shift=np.array([0, 0], dtype=np.float32), inv=0):
    scale = np.asarray(scale, dtype=np.float64)
    if scale.ndim == 0:
        scale = np.array([scale, scale])
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = get_dir([0, src_w * -0.5], rot_rad)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return solve_affine(dst, src) if inv else solve_affine(src, dst)


def invert_affine(m):
    m = np.array(m, np.float64).reshape(2, 3).copy()
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    m[0, 0], m[0, 1] = a11, m[0, 1] * -d
    m[1, 0], m[1, 1] = m[1, 0] * -d, a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    return m


def _round_int(v):
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_u8(image, trans, size):
    """cv2.warpAffine(image, trans, size): uint8 HWC in, uint8 [size[1], size[0], C] out."""
    h, w = image.shape[:2]
    wd, hd = int(size[0]), int(size[1])
    m = invert_affine(trans)
    xs = np.arange(wd, dtype=np.float64)
    ys = np.arange(hd, dtype=np.float64)
    adelta = _round_int(m[0, 0] * xs * AB_SCALE)
    bdelta = _round_int(m[1, 0] * xs * AB_SCALE)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    x0 = _round_int((m[0, 1] * ys + m[0, 2]) * AB_SCALE) + rd
    y0 = _round_int((m[1, 1] * ys + m[1, 2]) * AB_SCALE) + rd
    X = (x0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = X & (INTER_TAB_SIZE - 1)
    fy = Y & (INTER_TAB_SIZE - 1)
    w00 = np.minimum((32 - fx) * (32 - fy) * 32, 32767)
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32
    img = image.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[..., None], v, 0)

    acc = tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + \
        tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None]
    out = (acc + (1 << (COEF_BITS - 1))) >> COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_align_multi_scale(image, input_size, current_scale, min_scale):
    from .transforms_ref import get_multi_scale_size
    size_resized, center, scale = get_multi_scale_size(image.shape[:2], input_size, current_scale, min_scale)
    trans = get_affine_transform(center, scale, 0, size_resized)
    return warp_affine_u8(image, trans, size_resized), center, scale


def to_tensor_normalize(image_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """torchvision ToTensor (uint8 HWC -> float32 CHW / 255) + Normalize, all in float32."""
    t = image_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return (t - m) / s
