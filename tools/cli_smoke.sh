#!/bin/bash
# GPU box: the command line (kbe.py) end to end on a synthetic photograph, with and without --dolly (dev aid).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<PY
import numpy as np
from PIL import Image
from ken_burns_effect_amd import synthetic
img, _ = synthetic.make_rgbd(384, 512, 5)
Image.fromarray((img[0].numpy().transpose(1, 2, 0) * 255).astype(np.uint8)).save("/tmp/x.png")
PY
for extra in "" "--dolly" "--2d --write-frames"; do
  rm -rf /tmp/kbe_out
  python -m ken_burns_effect_amd.kbe --in /tmp/x.png --out /tmp/kbe_out --allow-random-weights $extra 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -3
  python -c "
import glob, os
f = glob.glob('/tmp/kbe_out/3d_kbe.*'); print('$extra ->', f[0], os.path.getsize(f[0]), 'bytes,', open(f[0], 'rb').read(12))"
  file /tmp/kbe_out/3d_kbe.* 2>/dev/null
done
