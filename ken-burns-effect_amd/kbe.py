"""Command line of the reference's ``kbe.py`` (:42-181), same long options:

    python -m ken_burns_effect_amd.kbe --in image.jpg --out outdir [--dolly] [--write-frames]
        [--inpaint-path P] [--refine-path P] [--estim-path P] [--inpaint-depth P] [--pretrained-refine]
        [--pretrained-estim] [--2d] [--startU/--startV/--endU/--endV/--startW/--startH/--endW/--endH N]
        [--semantics-path vgg19_bn_state_dict.pth]     (new: the reference downloads these weights through torchvision)
        [--allow-random-weights]                       (new: render with seeded weights where a checkpoint is missing instead of failing)

Images are read with PIL (OpenCV is not a dependency); like ``cv2.imread`` the pixels are handed to the
networks in BGR order unless ``--pretrained-estim`` is given (kbe.py:96-98).
"""
import getopt
import math
import sys

import numpy as np
import torch

LONG_OPTIONS = ['in=', 'out=', 'dolly', 'write-frames', 'inpaint-path=', 'refine-path=', 'estim-path=', 'startU=', 'startV=', 'endU=',
                'endV=', 'startW=', 'startH=', 'endW=', 'endH=', 'pretrained-refine', 'pretrained-estim', 'inpaint-depth=', '2d', 'semantics-path=', 'allow-random-weights']


def parse(argv):
    cfg = {'in': 'images/doublestrike.jpg', 'out': 'images/kbe', 'dolly': False, 'write-frames': False, 'pretrained-refine': False,
           'pretrained-estim': False, '2d': False, 'inpaint-depth': None, 'semantics-path': None, 'allow-random-weights': False,
           'inpaint-path': './models/trained/inpainting-color.tar', 'refine-path': './models/trained/disparity-refinement.tar',
           'estim-path': './models/trained/disparity-estimation-no-mask.tar'}
    window = dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH'))
    for option, argument in getopt.getopt(argv, '', LONG_OPTIONS)[0]:
        name = option[2:]
        if name in window:
            if argument != '':
                window[name] = int(argument)
        elif name in ('dolly', 'write-frames', 'pretrained-refine', 'pretrained-estim', '2d', 'allow-random-weights'):
            cfg[name] = True
        elif argument != '':
            cfg[name] = argument
    return cfg, window


def windows_for(width, height, window, dolly):
    """kbe.py:116-146: aspect-ratio completion, defaults, and the sanity asserts."""
    w = dict(window)
    if w['endH'] is not None and w['endW'] is None:
        w['endW'] = int(width * w['endH'] / height)
    if w['endW'] is not None and w['endH'] is None:
        w['endH'] = int(height * w['endW'] / width)
    if w['startH'] is not None and w['startW'] is None:
        w['startW'] = int(width * w['startH'] / height)
    if w['startW'] is not None and w['startH'] is None:
        w['startH'] = int(height * w['startW'] / width)
    if None in w.values():
        print('At least one of the cropping parameters was not defined, using default ones for %s.' % ('dolly effect' if dolly else '3D kbe'))
        if not dolly:
            w.update(startU=width / 2.15, startV=height / 2.15, startW=int(math.floor(0.90 * width)), startH=int(math.floor(0.90 * height)),
                     endU=width / 1.85, endV=height / 1.85, endW=int(math.floor(0.85 * width)), endH=int(math.floor(0.85 * height)))
        else:
            w.update(startU=width / 2, startV=height / 2, startW=int(math.floor(0.8 * width)), startH=int(math.floor(0.8 * height)),
                     endU=width / 2, endV=height / 2, endW=int(math.floor(0.3 * width)), endH=int(math.floor(0.3 * height)))
    assert height >= w['startV'] + w['startH'] / 2 and w['startV'] - w['startH'] / 2 >= 0, 'Start window too tall compared to given center'
    assert width >= w['startU'] + w['startW'] / 2 and w['startU'] - w['startW'] / 2 >= 0, 'Start window too wide compared to given center'
    assert height >= w['endV'] + w['endH'] / 2 and w['endV'] - w['endH'] / 2 >= 0, 'End window too tall compared to given center'
    assert width >= w['endU'] + w['endW'] / 2 and w['endU'] - w['endW'] / 2 >= 0, 'End window too wide compared to given center'
    return {'objectFrom': {'dblCenterU': w['startU'], 'dblCenterV': w['startV'], 'intCropWidth': w['startW'], 'intCropHeight': w['startH']},
            'objectTo': {'dblCenterU': w['endU'], 'dblCenterV': w['endV'], 'intCropWidth': w['endW'], 'intCropHeight': w['endH']}}


def load_image(path, rgb):
    """[1,3,H,W] in 0..1, H and W cropped to multiples of 4 (kbe.py:96-114)"""
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)
    if not rgb:
        img = img[:, :, ::-1]           # cv2.imread order
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float() / 255.0
    h, w = t.shape[1] - t.shape[1] % 4, t.shape[2] - t.shape[2] % 4
    return t[:, :h, :w].unsqueeze(0)


def main(argv=None):
    from .pipeline import Pipeline
    torch.set_grad_enabled(False)
    cfg, window = parse(sys.argv[1:] if argv is None else argv)
    image = load_image(cfg['in'], cfg['pretrained-estim'])
    zoom = windows_for(image.shape[3], image.shape[2], window, cfg['dolly'])
    paths = [cfg['estim-path'], cfg['refine-path'], cfg['inpaint-path']] + ([cfg['inpaint-depth']] if cfg['inpaint-depth'] else [])
    pipe = Pipeline(model_paths=paths, dolly=cfg['dolly'], output_frames=cfg['write-frames'], pretrain=cfg['pretrained-refine'], d2=cfg['2d'],
                    semantics_path=cfg['semantics-path'], allow_random_weights=cfg['allow-random-weights'] or None)
    frames = pipe(image, zoom, cfg['out'], pretrained_estim=cfg['pretrained-estim'])
    print('%d frames of %dx%d written to %s' % (len(frames), image.shape[3], image.shape[2], cfg['out']))


if __name__ == '__main__':
    main()
