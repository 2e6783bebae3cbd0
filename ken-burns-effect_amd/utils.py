"""Inference-side helpers of the reference's ``utils/utils.py`` (:17-18, :60-73, :202-217)."""
import os
import warnings

import torch
import torch.nn.functional as F

cuda = torch.cuda.is_available()
device = 'cuda:0' if cuda else 'cpu'


def resize_image(tensorImage, max_size=512):
    """Bilinear resize that keeps the aspect ratio and makes the longer side ``max_size`` (utils.py:60-73)."""
    if tensorImage.size(0) == 0:
        return None
    width, height = tensorImage.size(3), tensorImage.size(2)
    ratio = float(width) / float(height)
    width = min(int(max_size * ratio), max_size)
    height = min(int(max_size / ratio), max_size)
    return F.interpolate(input=tensorImage, size=(height, width), mode='bilinear', align_corners=False)


def load_models(models_list, models_paths, continue_training=False, seed_missing=True):
    """Loads ``{'model_state_dict': ...}`` checkpoints or bare state dicts (utils.py:202-217) into
    ``models_list[i]['model']``.  The reference's checkpoints are a Google-Drive download
    (``download.sh``); when a path is None or missing and ``seed_missing`` is set, the module gets
    deterministic name-seeded weights instead (:func:`synthetic.seeded_fill_`) and a warning says so."""
    from . import synthetic
    iter_nb = 0
    for idx, path in enumerate(models_paths or [None] * len(models_list)):
        if idx >= len(models_list):
            break
        entry = models_list[idx]
        if path is None or not os.path.exists(path):
            if not seed_missing:
                raise FileNotFoundError(path)
            warnings.warn('checkpoint %r not found: %s network runs with seeded random weights' % (path, entry['type']))
            synthetic.seeded_fill_(entry['model'], 1000 + idx)
            continue
        checkpoint = torch.load(path, map_location='cpu')
        if isinstance(checkpoint, dict) and 'model_state_dict' in checkpoint:
            entry['model'].load_state_dict(checkpoint['model_state_dict'])
            if continue_training:
                entry['opt'].load_state_dict(checkpoint['optimizer_' + entry['type'] + '_state_dict'])
                entry['schedule'].load_state_dict(checkpoint['scheduler_' + entry['type'] + '_state_dict'])
                iter_nb = checkpoint['nb_iter']
        else:
            entry['model'].load_state_dict(checkpoint)
    return iter_nb
