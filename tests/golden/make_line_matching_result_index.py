"""The paint of line_matching/data/line_matching_result.png, pixel by pixel, as an index map.

The picture is what LineMatching::DebugShow (line_matching.cpp:805-920) drew for frames 5 | 10 of data/mh04, side by side:
  * every line of the CURRENT frame (right half) as a dashed 1-px line in BGR (255, 112, 132)            (:822-826)
  * every line i of the REFERENCE frame (left half), in list order, in a colour of its own,
    bgr = Scalar(rand() % 256, rand() % 256, rand() % 256) after srand(0)                                (:811, :833-836)
    - matched: a solid 2-px line in that colour on the left AND on its match on the right                (:843-848)
    - unmatched: dashed in the pink above                                                                (:850)
    - its key points as filled circles in that colour (left) and circles / squares / arrows (right)      (:877-897)
  * the top 6 % darkened, with the two captions                                                          (:907-912)
srand(0) makes the colour of line i a known function of i (glibc's additive-feedback rand(), published algorithm,
restated below; g++ evaluates the three rand() calls of the Scalar right to left, so the picture's R is the first call).
So the picture says, for every pixel, WHICH line of the reference's list was painted there.

Written here (data, not source):  index int16 [480][1504]:  -1 not paint of a known colour, 0..222 the colour of
reference line i, 223 the pink of the dashed lines.  Rows 0..28 (the darkened caption band) are left at -1.

Run in the build container only (reads /root/reference):   python tests/golden/make_line_matching_result_index.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
N_LINES = 223                      # "Line num: 223", line_matching_result.json
PINK_RGB = (132, 112, 255)         # Scalar(255, 112, 132) is B, G, R
BAND = 29                          # int(480 * 0.06) rows darkened + the caption text


def glibc_rand(seed, n):
    """The first n values of glibc rand() after srand(seed) (TYPE_3 generator; srand(0) is srand(1))."""
    seed = seed or 1
    r = [0] * (344 + n)
    r[0] = seed
    for i in range(1, 31):
        r[i] = (16807 * r[i - 1]) % 2147483647
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, 344 + n):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return [x >> 1 for x in r[344:]]


def main():
    im = np.array(Image.open("/root/reference/line_matching/data/line_matching_result.png"))[..., :3].astype(np.int64)
    assert im.shape == (480, 1504, 3)
    key = (im[..., 0] << 16) | (im[..., 1] << 8) | im[..., 2]
    rnd = glibc_rand(0, 3 * N_LINES)
    index = np.full(key.shape, -1, np.int16)
    for i in range(N_LINES):
        r, g, b = (rnd[3 * i + k] % 256 for k in range(3))
        index[key == ((r << 16) | (g << 8) | b)] = i
    index[key == ((PINK_RGB[0] << 16) | (PINK_RGB[1] << 8) | PINK_RGB[2])] = N_LINES
    index[:BAND] = -1
    out = os.path.join(HERE, "line_matching_result_index.npz")
    np.savez_compressed(out, index=index)
    seen = np.unique(index[:, :752])
    print("colours of reference lines seen in the left half: %d of %d; pink pixels %d -> %s"
          % (int(((seen >= 0) & (seen < N_LINES)).sum()), N_LINES, int((index == N_LINES).sum()), out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
