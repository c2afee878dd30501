"""On-disk depth cache of VEON (second half): the wire format
the reference's offline pass writes and its data pipeline reads back in place of
the depth encoder.

Writer: ``VeonDepthCache.forward_train``
(mmdet3d/models/detectors/veon_depth_cache.py:146-157) -- one file per camera
image, ``torch.save`` of the (H/2, W/2) fp32 metric-depth map on the CPU, at
``<home>/<token[:2]>/<token>/<token>-<CAM>.tensor``; files that exist are kept.
Reader: ``PrepareImageInputs`` (mmdet3d/datasets/pipelines/loading.py:1259-1262,
1283-1286, 1321-1322) -- ``torch.load`` per camera, ``torch.stack`` over cameras
(and past frames).

Files written here are byte-compatible with the reference's (same ``torch.save``
of a plain CPU tensor); files are read with ``weights_only=True`` so nothing in
a cache file is ever executed.
"""
import os

import torch


def cache_path(home, unique_token):
    """``unique_token`` = '<sample token>-<CAM_NAME>' (loading.py:1257)."""
    token, _cam = unique_token.split('-', 1)
    return os.path.join(home, token[:2], token, unique_token + '.tensor')


def store(home, unique_tokens, depth, overwrite=False):
    """``depth``: (N, h, w) or (1, N, h, w) metric depth of the N cameras named by
    ``unique_tokens``.  Returns the paths written (existing files are skipped
    unless ``overwrite``, as the reference does)."""
    if depth.dim() == 4:
        assert depth.shape[0] == 1, 'one sample at a time (the reference uses depth[0])'
        depth = depth[0]
    assert depth.dim() == 3 and depth.shape[0] == len(unique_tokens)
    maps = depth.detach().to('cpu', torch.float32)
    written = []
    for i, tok in enumerate(unique_tokens):
        path = cache_path(home, tok)
        if os.path.exists(path) and not overwrite:
            continue
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + '.part%d' % os.getpid()
        torch.save(maps[i].clone(), tmp)   # clone: save the map, not the batch's storage
        os.replace(tmp, path)
        written.append(path)
    return written


def load(home, unique_tokens, device=None):
    """-> (len(unique_tokens), h, w) fp32, the reader's ``results['depth_preds']``."""
    maps = []
    for tok in unique_tokens:
        t = torch.load(cache_path(home, tok), map_location='cpu', weights_only=True)
        if not isinstance(t, torch.Tensor) or t.dim() != 2:
            raise ValueError('%s: not a 2-D depth map' % cache_path(home, tok))
        maps.append(t.float())
    out = torch.stack(maps)
    return out if device is None else out.to(device, non_blocking=True)
