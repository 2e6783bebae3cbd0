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
                raise FileNotFoundError('checkpoint %r of the %s network not found (the reference\'s download.sh fetches them); '
                                        'allow_random_weights=True / --allow-random-weights renders with seeded weights instead' % (path, entry['type']))
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


# ---------------------------------------------------------------------------------------
# training-side users of the splat core (utils/utils.py:219-300 of the reference)
# ---------------------------------------------------------------------------------------

def get_item_in_dict(dict_in, idx):
    """Item ``idx`` of every (nested) batched value (utils/utils.py:370-377)."""
    return {k: get_item_in_dict(v, idx) if isinstance(v, dict) else v[idx] for k, v in dict_in.items()}


def get_tensor_shift(objectCommon):
    """Camera shift of the END pose (dblStep = 1) of ``objectCommon['zoomSettings']`` (utils/utils.py:219-245)."""
    from . import common
    zoom = objectCommon['zoomSettings']
    dblFrom, dblTo = 0.0, 1.0
    shiftU = (dblFrom * zoom['objectFrom']['dblCenterU'] + dblTo * zoom['objectTo']['dblCenterU']) - objectCommon['intWidth'] / 2.0
    shiftV = (dblFrom * zoom['objectFrom']['dblCenterV'] + dblTo * zoom['objectTo']['dblCenterV']) - objectCommon['intHeight'] / 2.0
    cropW = dblFrom * zoom['objectFrom']['intCropWidth'] + dblTo * zoom['objectTo']['intCropWidth']
    depthFrom = objectCommon['objectDepthrange'][0]
    depthTo = depthFrom * (cropW / max(zoom['objectFrom']['intCropWidth'], zoom['objectTo']['intCropWidth']))
    _, tensorShift = common.process_shift({'tensorPoints': objectCommon['tensorRawPoints'], 'dblShiftU': shiftU, 'dblShiftV': shiftV,
                                           'dblDepthFrom': depthFrom, 'dblDepthTo': depthTo}, objectCommon)
    return tensorShift


def get_masks(tensorImage, tensorDisparity, tensorDepth, zoom_settings, camera, AFromB=True, tensorContext=None):
    """Disocclusion masks (``AFromB``) or the forward-warped view B with its hole mask, for a batch of RGBD images
    and per-sample zoom settings (utils/utils.py:248-300).  Returns what the reference returns:
    ``(tensorMasks, tensorShift, objectList)`` or ``(tensorRender, tensorMasks, tensorPoints, tensorShift, objectList)``."""
    from . import common, synthetic
    K = common._K()
    focal, baseline = camera['focal'], camera['baseline']
    B, _, H, W = tensorImage.shape
    tensorValid = K.laplacian_valid(tensorDisparity, tensorDisparity.max(), 0.03)
    tensorPoints = K.depth_to_points(tensorDepth, focal, valid=tensorValid)
    shifts, objects = [], []
    for idx in range(B):
        oc = {'dblFocal': focal, 'dblBaseline': baseline, 'intWidth': W, 'intHeight': H,
              'tensorRawImage': tensorImage[idx], 'tensorRawDisparity': tensorDisparity[idx],
              'dblDispmin': tensorDisparity[idx].min().item(), 'dblDispmax': tensorDisparity[idx].max().item(),
              'objectDepthrange': synthetic.depthrange_of(tensorDepth[idx:idx + 1]),
              'tensorRawPoints': tensorPoints[idx].view(1, 3, -1), 'zoomSettings': get_item_in_dict(zoom_settings, idx)}
        shifts.append(get_tensor_shift(oc))
        objects.append(oc)
    tensorShift = torch.cat(shifts)
    flat = tensorPoints.view(B, 3, -1)
    if AFromB:
        return common.generate_mask(flat, tensorShift, W, H, focal, baseline), tensorShift, objects
    data = [tensorImage, tensorDisparity] + ([tensorContext] if tensorContext is not None else [])
    tensorRender, tensorMasks = common.render_pointcloud(flat + tensorShift, torch.cat(data, 1).view(B, -1, H * W), W, H, focal, baseline)
    return tensorRender, (tensorMasks > 0.0).float(), flat, tensorShift, objects


def _miopen_user_db():
    """Where MIOpen keeps this user's find-db (MIOPEN_USER_DB_PATH, default ~/.config/miopen) and what is in it: {file: size}."""
    import os
    root = os.environ.get('MIOPEN_USER_DB_PATH') or os.path.join(os.path.expanduser('~'), '.config', 'miopen')
    found = {}
    for base, _, files in os.walk(root):
        for f in files:
            try:
                found[os.path.join(base, f)] = os.path.getsize(os.path.join(base, f))
            except OSError:
                pass
    return root, found


class miopen_tuned_once:
    """``with miopen_tuned_once(tag, device):`` -- the block's convolutions run under MIOpen's find step
    (torch.backends.cudnn.benchmark) the first time this machine sees ``tag`` and in PyTorch's immediate mode from then on (the find
    step leaves its measurements in MIOpen's user find-db, which the immediate mode of every later process consults; a marker file
    under KBE_CACHE_DIR, default ~/.cache/kbe/miopen-tuned/, records that it ran).  The find step costs tens of seconds, once per
    machine, find-db and tag -- it says so on stderr.  The marker is written only when the find-db GREW during the block (a
    read-only or disabled db -- MIOPEN_DISABLE_CACHE, MIOPEN_USER_DB_PATH on a read-only mount -- keeps nothing: marking the size
    tuned would leave it untuned for good) and its name carries the db's path.  In a process group only rank 0 tunes
    (the others run in immediate mode: several ranks measuring solvers against each other measure each other).  The switch it
    flips, torch.backends.cudnn.benchmark, is process-global: not for use from several threads at once.
    ``enabled=False`` (or a CPU device): nothing happens."""

    def __init__(self, tag, device, enabled=True):
        import hashlib
        import os
        self.marker = None
        if enabled and torch.device(device).type == 'cuda' and torch.cuda.is_available() and not os.environ.get('MIOPEN_DISABLE_CACHE'):
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
                enabled = False
        if enabled and torch.device(device).type == 'cuda' and torch.cuda.is_available() and not os.environ.get('MIOPEN_DISABLE_CACHE'):
            root = os.environ.get('KBE_CACHE_DIR') or os.path.join(os.path.expanduser('~'), '.cache', 'kbe')
            dev = torch.cuda.get_device_name(device).replace(' ', '_')
            self.db_root, _ = _miopen_user_db()
            name = '%s-%s-torch%s-hip%s-db%s' % (tag, dev, torch.__version__, getattr(torch.version, 'hip', None), hashlib.sha1(self.db_root.encode()).hexdigest()[:8])
            self.marker = os.path.join(root, 'miopen-tuned', name.replace('/', '_'))
            if os.path.exists(self.marker):
                self.marker = None
        self.tag = tag

    def __enter__(self):
        self.was = torch.backends.cudnn.benchmark
        if self.marker is not None:
            import sys
            sys.stderr.write('ken_burns_effect_amd: first "%s" on this machine: MIOpen measures its convolution solvers once (tens of seconds; the '
                             'results stay in its find-db under %s, marker %s; KBE_MIOPEN_FIND=0 skips this)\n' % (self.tag, self.db_root, self.marker))
            _, self.db_before = _miopen_user_db()
            torch.backends.cudnn.benchmark = True
        return self

    def __exit__(self, exc_type, exc, tb):
        import os
        import sys
        if self.marker is not None:
            torch.backends.cudnn.benchmark = self.was
            if exc_type is None:
                _, after = _miopen_user_db()
                grew = any(size > self.db_before.get(path, 0) for path, size in after.items())
                if not grew:
                    sys.stderr.write('ken_burns_effect_amd: MIOpen kept nothing of that find step under %s (read-only or disabled find-db?): '
                                     '"%s" is NOT marked tuned\n' % (self.db_root, self.tag))
                    return False
                try:
                    os.makedirs(os.path.dirname(self.marker), exist_ok=True)
                    open(self.marker, 'w').write('tuned\n')
                except OSError:
                    pass        # a read-only home: the find step runs again next time
        return False
