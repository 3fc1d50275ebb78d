"""Result records -> the evaluator's keypoint JSON.

Host-side formatting of the gathered keypoint records, in the format the reference writes for
`crowdposetools` / `pycocotools` (lib/dataset/CrowdPoseDataset.py:147-279, COCODataset.py:159-292):
one entry per person with `image_id`, `category_id`, `keypoints` [J*3] (x, y, joint score), `score`
(mean joint value before refine) and `bbox` = [min x, min y, width, height] over the J joints.  The
reference keeps every detection (its OKS-NMS `keep` list is empty, CrowdPoseDataset.py:196-204), so
does this writer.  Persons are emitted in image order, then record order (= person creation order).
"""
import json

import numpy as np


def person_area(kpt):
    """CrowdPoseDataset.py:173 -- (max x - min x) * (max y - min y) over all J rows."""
    return float((np.max(kpt[:, 0]) - np.min(kpt[:, 0])) * (np.max(kpt[:, 1]) - np.min(kpt[:, 1])))


def records_to_results(kpts, count, scores, image_ids, category_id=1, num_joints=None):
    """kpts [N,Pcap,J,3+T] / count [N] / scores [N,Pcap] (host arrays or tensors, e.g. the output of
    ``parallel.all_gather_records``) -> list of result dicts (CrowdPoseDataset.py:240-279).

    ``num_joints`` drops trailing joints (WITH_CENTER and not IGNORE_CENTER, :174-176)."""
    kpts = np.asarray(kpts.cpu() if hasattr(kpts, 'cpu') else kpts)
    count = np.asarray(count.cpu() if hasattr(count, 'cpu') else count)
    scores = np.asarray(scores.cpu() if hasattr(scores, 'cpu') else scores)
    if len(image_ids) != kpts.shape[0]:
        raise ValueError('one image id per record row is required')
    J = kpts.shape[2] if num_joints is None else int(num_joints)
    out = []
    for n in range(kpts.shape[0]):
        c = int(count[n])
        if c < 0 or c > kpts.shape[1]:
            raise ValueError('image %d: count %d outside the record capacity %d' % (n, c, kpts.shape[1]))
        for p in range(c):
            kp = kpts[n, p, :J, 0:3].astype(np.float64)          # key_points is a float64 array (:254-257)
            lt = np.amin(kp, axis=0)
            rb = np.amax(kp, axis=0)
            out.append({
                'image_id': int(image_ids[n]),
                'category_id': int(category_id),
                'keypoints': [float(v) for v in kp.reshape(-1)],
                'score': float(scores[n, p]),
                'bbox': [float(lt[0]), float(lt[1]), float(rb[0] - lt[0]), float(rb[1] - lt[1])],
            })
    return out


def preds_to_results(all_preds, all_scores, image_ids, category_id=1, num_joints=None):
    """The valid.py accumulators (``all_preds``: per image a list of [J,3+T] arrays from
    get_final_preds; ``all_scores``: per image a list of floats) -> the same result list."""
    out = []
    for n, persons in enumerate(all_preds):
        for p, kpt in enumerate(persons):
            kpt = np.asarray(kpt)
            J = kpt.shape[0] if num_joints is None else int(num_joints)
            kp = kpt[:J, 0:3].astype(np.float64)
            lt = np.amin(kp, axis=0)
            rb = np.amax(kp, axis=0)
            out.append({
                'image_id': int(image_ids[n]),
                'category_id': int(category_id),
                'keypoints': [float(v) for v in kp.reshape(-1)],
                'score': float(all_scores[n][p]),
                'bbox': [float(lt[0]), float(lt[1]), float(rb[0] - lt[0]), float(rb[1] - lt[1])],
            })
    return out


def write_results(results, res_file):
    """json.dump(results, f, sort_keys=True, indent=4) as in CrowdPoseDataset.py:233-236."""
    with open(res_file, 'w') as f:
        json.dump(results, f, sort_keys=True, indent=4)
    return res_file
