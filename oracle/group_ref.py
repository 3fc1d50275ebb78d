"""NumPy restatement of the associative-embedding parser.  TEST INFRASTRUCTURE.

Follows /root/reference/lib/core/group.py:
  * Params                 :100-120
  * HeatmapParser.nms      :131-135   (MaxPool2d(k,1,pad) pads with -inf)
  * HeatmapParser.top_k    :141-176
  * match_by_tag           :26-97     (float64; per-person mean tag in float32)
  * HeatmapParser.adjust   :178-197
  * scores                 :275
  * HeatmapParser.refine   :199-267
  * HeatmapParser.parse    :269-291
Implicit third-party contracts restated explicitly (each probed against
torch 2.10 / numpy 2.2 on this image, see tests/test_oracle_pinning.py):
  * ``torch.mean(x[n,T], dim=0)``: 4 interleaved fp32 partial sums
    (ATen SumKernel ``row_sum``, ilp_factor 4), tail rows into partial 0,
    ((p0+p1)+p2)+p3, then a true division by n.
  * ``ndarray[:,2].mean()`` (fp32, strided): NumPy pairwise sum (8 accumulators
    for n >= 8), then division by n.
  * ``np.mean(list_of_rows, axis=0)``: sequential fp32 row adds, then / n.
  * ``np.round`` / ``torch.round``: half-to-even.  ``argmax``: first maximum.
  * ``torch.topk`` tie order is implementation-defined; this oracle (and the HIP
    path) fix it to (value desc, linear index asc), and fill slots beyond the
    strictly-positive NMS survivors with (val 0, x 0, y 0, tag 0).  Such slots
    can never pass ``val > DETECTION_THRESHOLD`` (threshold must be >= 0).
"""
import numpy as np

from . import munkres_ref

f32 = np.float32


class Params(object):
    # group.py:100-120
    def __init__(self, num_joints=14, max_num_people=30, detection_threshold=0.1,
                 tag_threshold=1.0, use_detection_val=True, ignore_too_much=False,
                 with_center=False, ignore_center=True, nms_kernel=5, nms_padding=2,
                 tag_per_joint=True):
        self.num_joints = num_joints
        self.max_num_people = max_num_people
        self.detection_threshold = detection_threshold
        self.tag_threshold = tag_threshold
        self.use_detection_val = use_detection_val
        self.ignore_too_much = ignore_too_much
        self.nms_kernel = nms_kernel
        self.nms_padding = nms_padding
        self.tag_per_joint = tag_per_joint
        if with_center and ignore_center:
            self.num_joints -= 1
        if with_center and not ignore_center:
            self.joint_order = [i - 1 for i in
                                [18, 1, 2, 3, 4, 5, 6, 7, 12, 13, 8, 9, 10, 11, 14, 15, 16, 17]]
        else:
            self.joint_order = [i - 1 for i in
                                [1, 2, 3, 4, 5, 6, 7, 12, 13, 8, 9, 10, 11, 14, 15, 16, 17]]
        if detection_threshold < 0:
            raise ValueError('DETECTION_THRESHOLD must be >= 0')


# ---------------------------------------------------------------- nms / top_k
def maxpool(det, k, pad):
    """MaxPool2d(k, stride 1, padding pad) over the last two dims, -inf padding."""
    assert k == 2 * pad + 1, 'only same-size windows are on the path'
    H, W = det.shape[-2:]
    p = np.full(det.shape[:-2] + (H + 2 * pad, W + 2 * pad), -np.inf, dtype=det.dtype)
    p[..., pad:pad + H, pad:pad + W] = det
    m = p[..., :, 0:W]
    for d in range(1, k):
        m = np.maximum(m, p[..., :, d:d + W])
    r = m[..., 0:H, :]
    for d in range(1, k):
        r = np.maximum(r, m[..., d:d + H, :])
    return r


def nms(det, k=5, pad=2):
    maxm = maxpool(det, k, pad)
    return det * (maxm == det).astype(f32)


def top_k(det, tag, params):
    """det [N,J,H,W] f32, tag [N,J,H,W,T] f32 ->
    dict(tag_k [N,J,M,T] f32, loc_k [N,J,M,2] i64 (x,y), val_k [N,J,M] f32)."""
    det = np.asarray(det, dtype=f32)
    tag = np.asarray(tag, dtype=f32)
    N, J, H, W = det.shape
    T = tag.shape[4]
    M = params.max_num_people
    d = nms(det, params.nms_kernel, params.nms_padding).reshape(N, J, H * W)
    tg = tag.reshape(N, -1, H * W, T)
    val_k = np.zeros((N, J, M), f32)
    ind_k = np.zeros((N, J, M), np.int64)
    tag_k = np.zeros((N, J, M, T), f32)
    for n in range(N):
        for j in range(J):
            plane = d[n, j]
            cand = np.nonzero(plane > 0)[0]
            if cand.size:
                order = np.lexsort((cand, -plane[cand].astype(np.float64)))
                sel = cand[order[:M]]
                c = sel.size
                val_k[n, j, :c] = plane[sel]
                ind_k[n, j, :c] = sel
                tj = j if params.tag_per_joint else 0
                tag_k[n, j, :c] = tg[n, tj, sel]
    loc_k = np.stack((ind_k % W, ind_k // W), axis=3)
    return dict(tag_k=tag_k, loc_k=loc_k, val_k=val_k)


# ---------------------------------------------------------------- match_by_tag
def _mean_rows_f32(rows):
    """np.mean(list of f32 rows, axis=0): sequential fp32 adds, then / n."""
    acc = rows[0].astype(f32).copy()
    for r in rows[1:]:
        acc = (acc + r).astype(f32)
    return (acc / f32(len(rows))).astype(f32)


def match_by_tag(tag_k, loc_k, val_k, params):
    """One image.  tag_k [J,M,T] f32, loc_k [J,M,2] i64, val_k [J,M] f32 ->
    ans [P,J,3+T] f32 (x, y, val, tags...), persons in creation order."""
    T = tag_k.shape[2]
    default_ = np.zeros((params.num_joints, 3 + T))
    keys = []          # person key = first tag component (float), creation order
    joint_rows = []    # per person [J,3+T] float64
    tag_lists = []     # per person list of f32 tag rows

    def person_slot(key):
        # dict semantics (group.py:50-52,91-94): float ==, so -0.0 == 0.0, NaN never
        for s, k in enumerate(keys):
            if k == key:
                return s
        keys.append(key)
        joint_rows.append(np.copy(default_))
        tag_lists.append(None)
        return len(keys) - 1

    for i in range(params.num_joints):
        idx = params.joint_order[i]
        tags = tag_k[idx]
        joints = np.concatenate((loc_k[idx], val_k[idx, :, None], tags), 1)   # -> float64
        mask = joints[:, 2] > params.detection_threshold
        tags = tags[mask]
        joints = joints[mask]
        if joints.shape[0] == 0:
            continue
        if i == 0 or len(keys) == 0:
            for tag, joint in zip(tags, joints):
                s = person_slot(tag[0])
                joint_rows[s][idx] = joint
                tag_lists[s] = [tag]
        else:
            ngrp = min(len(keys), params.max_num_people)
            grouped_tags = [_mean_rows_f32(tag_lists[s]) for s in range(ngrp)]
            if params.ignore_too_much and ngrp == params.max_num_people:
                continue
            diff = joints[:, None, 3:] - np.array(grouped_tags)[None, :, :]
            # np.linalg.norm(ord=2, axis=2): sqrt(sum(|x|^2)) in float64
            diff_normed = np.sqrt(np.add.reduce((diff * diff), axis=2))
            diff_saved = np.copy(diff_normed)
            if params.use_detection_val:
                diff_normed = np.round(diff_normed) * 100 - joints[:, 2:3]
            num_added, num_grouped = diff.shape[0], diff.shape[1]
            if num_added > num_grouped:
                diff_normed = np.concatenate(
                    (diff_normed, np.zeros((num_added, num_added - num_grouped)) + 1e10), axis=1)
            pairs = munkres_ref.compute(diff_normed)
            for row, col in pairs:
                if row < num_added and col < num_grouped and \
                        diff_saved[row][col] < params.tag_threshold:
                    joint_rows[col][idx] = joints[row]
                    tag_lists[col].append(tags[row])
                else:
                    s = person_slot(tags[row][0])
                    joint_rows[s][idx] = joints[row]
                    tag_lists[s] = [tags[row]]
    if not keys:
        return np.zeros((0,), f32)       # np.array([]).astype(float32), group.py:96
    return np.array(joint_rows).astype(f32)


# ---------------------------------------------------------------- adjust / scores / refine
def adjust(ans, det):
    """ans: list over batch of [P,J,3+T]; det [N,J,H,W].  In place (group.py:178-197)."""
    for b, people in enumerate(ans):
        for p, person in enumerate(people):
            for j, joint in enumerate(person):
                if joint[2] > 0:
                    y, x = joint[0:2]            # NB names swapped in the source
                    xx, yy = int(x), int(y)
                    tmp = det[b][j]
                    if tmp[xx, min(yy + 1, tmp.shape[1] - 1)] > tmp[xx, max(yy - 1, 0)]:
                        y += 0.25
                    else:
                        y -= 0.25
                    if tmp[min(xx + 1, tmp.shape[0] - 1), yy] > tmp[max(0, xx - 1), yy]:
                        x += 0.25
                    else:
                        x -= 0.25
                    ans[b][p, j, 0:2] = (y + 0.5, x + 0.5)
    return ans


def mean_strided_f32(x):
    """fp32 ``ndarray[:, c].mean()``: NumPy pairwise sum then / n."""
    n = len(x)
    if n < 8:
        r = f32(0)
        for v in x:
            r = f32(r + v)
    else:
        acc = [f32(x[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                acc[j] = f32(acc[j] + x[i + j])
            i += 8
        r = f32(f32(f32(acc[0] + acc[1]) + f32(acc[2] + acc[3])) +
                f32(f32(acc[4] + acc[5]) + f32(acc[6] + acc[7])))
        while i < n:
            r = f32(r + x[i])
            i += 1
    return f32(r / f32(n))


def torch_mean_dim0_f32(rows):
    """fp32 ``torch.mean(x[n,T], dim=0)`` on CPU (ATen row_sum, ilp 4; then / n)."""
    rows = np.asarray(rows, dtype=f32)
    n, T = rows.shape
    q = n // 4
    out = np.zeros(T, f32)
    for t in range(T):
        p = [f32(0)] * 4
        for i in range(q):
            for k in range(4):
                p[k] = f32(p[k] + rows[4 * i + k, t])
        for i in range(4 * q, n):
            p[0] = f32(p[0] + rows[i, t])
        r = p[0]
        for k in range(1, 4):
            r = f32(r + p[k])
        out[t] = f32(r / f32(n))
    return out


def refine(det, tag, keypoints):
    """det [J,H,W] f32, tag [J,H,W,T] f32, keypoints [J,3+T] f32 (in place)."""
    J, H, W = det.shape
    tags = []
    for i in range(keypoints.shape[0]):
        if keypoints[i, 2] > 0:
            x, y = keypoints[i][:2].astype(np.int32)
            tags.append(tag[i, y, x])
    prev_tag = torch_mean_dim0_f32(np.stack(tags, 0))
    d = (tag - prev_tag[None, None, None, :]).astype(f32)
    tt = np.sqrt(np.add.reduce((d * d).astype(f32), axis=3, dtype=f32)).astype(f32)
    tmp2 = (det - np.round(tt)).astype(f32).reshape(J, -1)
    pos = tmp2.argmax(axis=1)
    cand = []
    for i in range(J):
        tmp = det[i]
        y = int(pos[i]) // W
        x = int(pos[i]) % W
        xx, yy = x, y
        val = tmp[y, x]
        x += 0.5
        y += 0.5
        if tmp[yy, min(xx + 1, W - 1)] > tmp[yy, max(xx - 1, 0)]:
            x += 0.25
        else:
            x -= 0.25
        if tmp[min(yy + 1, H - 1), xx] > tmp[max(0, yy - 1), xx]:
            y += 0.25
        else:
            y -= 0.25
        cand.append((x, y, val))
    for i in range(J):
        if cand[i][2] > 0 and keypoints[i, 2] == 0:
            keypoints[i, :2] = cand[i][:2]
            keypoints[i, 2] = cand[i][2]
    return keypoints


class HeatmapParser(object):
    def __init__(self, params):
        self.params = params

    def parse_image(self, det, tag, adjust_=True, refine_=True):
        """One image: det [J,H,W], tag [J,H,W,T] -> (ans [P,J,3+T] f32, scores [P] f32).
        This is ``parse`` on a batch of one (the only batch size valid.py allows)."""
        det = np.ascontiguousarray(det, dtype=f32)
        tag = np.ascontiguousarray(tag, dtype=f32)
        tk = top_k(det[None], tag[None], self.params)
        ans = match_by_tag(tk['tag_k'][0], tk['loc_k'][0], tk['val_k'][0], self.params)
        if ans.ndim != 3:
            return np.zeros((0, det.shape[0], 3 + tag.shape[3]), f32), np.zeros((0,), f32)
        if adjust_:
            ans = adjust([ans], det[None])[0]
        scores = np.array([mean_strided_f32(p[:, 2]) for p in ans], dtype=f32)
        if refine_:
            for i in range(len(ans)):
                ans[i] = refine(det, tag, ans[i])
        return ans, scores

    def parse(self, det, tag, adjust=True, refine=True):
        """Reference-shaped: returns ([ans_image0], scores_image0)."""
        a, s = self.parse_image(det[0], tag[0], adjust, refine)
        return [a], [x for x in s]
