"""Seeded synthetic inputs shared by the oracle, the parity tests and bench.py.

TEST INFRASTRUCTURE (SURVEY.md §8d "concrete synthetic inputs").  There are no
checkpoints or datasets offline, so:
  1. images      : ``randn(N,3,R,R)`` from a seeded torch CPU generator
  2. weights     : seeded reference-format ``state_dict`` with RANDOMISED BN
                   statistics (default BN init is ~identity and hides fold bugs)
  3. AE maps     : Gaussian-blob heatmaps + per-person tag plateaus, because
                   random weights give |heatmap| below DETECTION_THRESHOLD
Generators are pure functions of their seed (torch mt19937 / numpy PCG64 streams
are platform independent), so the GPU box regenerates the same bits.
"""
import math

import numpy as np
import torch

from . import spec


def make_state_dict(arch, head=None, seed=1234, head_gain=1.0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in spec.state_dict_shapes(arch, head).items():
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith('running_var') or (k.endswith('.weight') and len(shp) == 1):
            sd[k] = torch.rand(shp, generator=g) + 0.5                   # U(0.5,1.5)
        elif k.endswith('running_mean') or k.endswith('.bias'):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        else:
            if k.startswith('deconv'):
                fan_in = shp[0] * 4                      # 2x2 live taps per output
            else:
                fan_in = shp[1] * shp[2] * shp[3]
            std = math.sqrt(2.0 / fan_in)
            if '.point_conv.' in k:
                std *= 0.35                      # keeps the residual trunk O(1)
            elif k.startswith('deconv'):
                std *= 0.5
            elif k.endswith('conv.3.weight'):
                std *= 0.05 * head_gain          # |heatmap| ~ 0.1-0.3: a few noise peaks > 0.1
            sd[k] = torch.randn(shp, generator=g) * std
    return sd


def make_images(n, r, seed=7, w=None):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, r, w or r, generator=g)


def blob_scene(rng, J, H, W, T, n_people, sigma=4.0, p_joint=0.8):
    """One image: det [J,H,W] f32, tag [J,H,W,T] f32 (SURVEY.md §8d input 3)."""
    det = rng.uniform(0.0, 0.02, size=(J, H, W)).astype(np.float32)
    tag = rng.normal(0.0, 0.05, size=(J, H, W, T)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for p in range(n_people):
        cx0, cy0 = rng.uniform(0.1 * W, 0.9 * W), rng.uniform(0.1 * H, 0.9 * H)
        base = 1.7 * (p + 1)
        for j in range(J):
            if rng.uniform() > p_joint:
                continue
            cx = cx0 + rng.normal(0, 0.08 * W)
            cy = cy0 + rng.normal(0, 0.08 * H)
            amp = rng.uniform(0.3, 1.0)
            gsn = (amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sigma * sigma))).astype(np.float32)
            det[j] = np.maximum(det[j], gsn + det[j] * (gsn < 0.02))
            sup = gsn > 0.05
            for t in range(T):
                tv = np.float32(base + rng.normal(0, 0.1))
                tag[j, :, :, t] = np.where(sup, tv + tag[j, :, :, t], tag[j, :, :, t])
    return det, tag


def blob_batch(seed, N, J=14, H=256, W=256, T=2, people=None, sigma=4.0):
    """Batch of scenes; ``people`` = list of person counts (default: 0..12 cycling)."""
    rng = np.random.default_rng(seed)
    det = np.zeros((N, J, H, W), np.float32)
    tag = np.zeros((N, J, H, W, T), np.float32)
    for n in range(N):
        P = people[n % len(people)] if people is not None else int(rng.integers(0, 13))
        det[n], tag[n] = blob_scene(rng, J, H, W, T, P, sigma=sigma)
    return det, tag


def lowres_offsets(seed, N, J, R, people=None):
    """Blob maps at the NETWORK OUTPUT resolutions, to be added to out0 / out1 so the
    AE stage of an end-to-end run does real work (SURVEY.md §8d input 4).

    Returns (off0 [N,2J,R/4,R/4], off1 [N,J,R/2,R/2]) for the un-flipped pass; the
    flipped pass uses ``flip_offsets`` so both passes describe the same scene."""
    rng = np.random.default_rng(seed)
    h0, h1 = R // 4, R // 2
    off0 = np.zeros((N, 2 * J, h0, h0), np.float32)
    off1 = np.zeros((N, J, h1, h1), np.float32)
    for n in range(N):
        P = people[n % len(people)] if people is not None else int(rng.integers(1, 11))
        d0, t0 = blob_scene(rng, J, h0, h0, 1, P, sigma=1.5)
        off0[n, :J] = d0
        off0[n, J:] = t0[..., 0]
        # stage-1 heatmaps describe the same blobs (nearest x2 of the stage-0 ones)
        off1[n] = d0.repeat(2, axis=1).repeat(2, axis=2)
    return off0, off1


def flip_offsets(off0, off1, flip_index):
    """Offsets for the flipped forward pass: un-flipping them (flip W, permute joints)
    must give back the un-flipped scene."""
    J = off1.shape[1]
    fi = np.asarray(flip_index)
    inv = np.argsort(fi)
    f0 = off0[:, :, :, ::-1].copy()
    f1 = off1[:, :, :, ::-1].copy()
    g0 = f0.copy()
    g0[:, :J] = f0[:, :J][:, inv]
    g0[:, J:] = f0[:, J:][:, inv]
    g1 = f1[:, inv]
    return np.ascontiguousarray(g0), np.ascontiguousarray(g1)
