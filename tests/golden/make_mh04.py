"""Decodes the 15 EuRoC MH_04 frames the reference ships as test data (line_matching/data/mh04/imgs/1..15.png) into one
compressed array file, tests/golden/mh04_frames.npz (key "frames": uint8 [15, 480, 752], frame k at index k - 1).
Data only (pixels); run in the build container where /root/reference exists:  python tests/golden/make_mh04.py"""
import os
import sys

import numpy as np
from PIL import Image

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/line_matching/data/mh04/imgs"
HERE = os.path.dirname(os.path.abspath(__file__))

frames = []
for k in range(1, 16):
    im = Image.open(os.path.join(SRC, "%d.png" % k))
    a = np.asarray(im.convert("L") if im.mode != "L" else im, dtype=np.uint8)
    assert a.shape == (480, 752), a.shape
    frames.append(a)
frames = np.stack(frames)
for k in (1, 2):   # the two frames committed in round 1 must be the same pixels
    old = os.path.join(HERE, "mh04_%d.npy" % k)
    if os.path.exists(old):
        assert np.array_equal(np.load(old), frames[k - 1])
np.savez_compressed(os.path.join(HERE, "mh04_frames.npz"), frames=frames)
print("wrote", frames.shape, frames.dtype)
