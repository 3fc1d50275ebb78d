"""Object-keypoint similarity between two sets of person records -- TEST INFRASTRUCTURE (checker only; used by
bench.py's parity section, tools/p3_agreement.py and tests/).

The reference evaluates with crowdposetools' COCOeval (lib/dataset/CrowdPoseDataset.py:281-312, third-party, not
under /root/reference): per ground-truth person OKS = mean over its labelled joints of exp(-d^2 / (2 s^2 k_j^2)),
s^2 = the person's area, k_j = 2 sigma_j with CrowdPose's published per-joint sigmas.  Here the "ground truth" is the
record set of the fp32 CPU pipeline (oracle network + merge + parser) and the candidate is the device pipeline, so the
number says how far an approximate path (bf16 storage) moves the KEYPOINTS, in the unit mAP is made of -- SURVEY.md
section 7, hard part 1 (iv): a path that cannot be index-exact needs such a metric.  Area: the reference uses the
bounding box of the predicted joints scaled like CrowdPoseDataset.py:171-185 (keypoint-box area); the same here on
the CPU person's joints."""
import numpy as np

# CrowdPose (crowdposetools cocoeval.py): 14 joints, sigmas / 10
CROWDPOSE_SIGMAS = np.array([.79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89, .79, .79]) / 10.0
# COCO (pycocotools cocoeval.py): 17 joints
COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def _sigmas(J):
    return CROWDPOSE_SIGMAS if J == 14 else (COCO_SIGMAS if J == 17 else np.full(J, 0.079))


def person_oks(ref, cand):
    """ref, cand: [J, >=3] rows (x, y, val): OKS of `cand` against `ref` over ref's detected joints (val > 0); a joint
    cand did not detect counts as distance infinity.  Area = bounding box of ref's detected joints (>= 1 px^2)."""
    vis = ref[:, 2] > 0
    if not vis.any():
        return float('nan')
    xs, ys = ref[vis, 0], ref[vis, 1]
    area = max(float((xs.max() - xs.min()) * (ys.max() - ys.min())), 1.0)
    k2 = (2.0 * _sigmas(ref.shape[0])) ** 2
    d2 = (cand[:, 0] - ref[:, 0]) ** 2 + (cand[:, 1] - ref[:, 1]) ** 2
    e = np.exp(-d2 / (2.0 * area * k2))
    e = np.where(cand[:, 2] > 0, e, 0.0)
    return float(e[vis].mean())


def image_oks(ref_persons, cand_persons):
    """Greedy one-to-one matching by descending OKS (COCOeval's rule with one IoU threshold removed): returns the list
    of per-REFERENCE-person OKS values; a reference person left without a candidate scores 0."""
    R, Cn = len(ref_persons), len(cand_persons)
    if R == 0:
        return []
    if Cn == 0:
        return [0.0] * R
    m = np.array([[person_oks(r, c) for c in cand_persons] for r in ref_persons])
    m = np.nan_to_num(m, nan=0.0)
    out = [0.0] * R
    used_r, used_c = set(), set()
    for idx in np.argsort(-m, axis=None):
        r, c = divmod(int(idx), Cn)
        if r in used_r or c in used_c:
            continue
        out[r] = float(m[r, c])
        used_r.add(r)
        used_c.add(c)
        if len(used_r) == R or len(used_c) == Cn:
            break
    return out


def summary(values):
    """mean / 5th percentile / minimum of a list of per-person OKS values."""
    v = np.asarray(values, np.float64)
    if v.size == 0:
        return {'persons': 0, 'mean': None, 'p05': None, 'min': None}
    return {'persons': int(v.size), 'mean': round(float(v.mean()), 5), 'p05': round(float(np.percentile(v, 5)), 5),
            'min': round(float(v.min()), 5)}
