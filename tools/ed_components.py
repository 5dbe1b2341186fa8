"""Would routing per connected component parallelise k_ed_route?  

A walk of EdgeDrawing only steps onto pixels with gImg > 0 and only reads / writes the edge flag of pixels it visits
(edline_detector.cpp:191-647), so walks in different 8-connected components of {gImg > 0} cannot interact: one walker per
component would give bit-identical chains, and the serial critical path would be the largest component's share of the
anchors.  This measures that share on the reference's own frames (tests/golden/mh04_frames.npz) with the oracle's gradient
stage and scipy's labelling.  CPU only.

    python tools/ed_components.py
"""
import os
import sys

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as o  # noqa: E402

F = np.load(os.path.join(ROOT, "tests", "golden", "mh04_frames.npz"))["frames"]
print("frame  components  anchors  largest component: anchors (share)   chain pixels (share)")
shares = []
for fi in range(len(F)):
    for smoothed in (True, False):
        _, st = o.edlines(F[fi], want_stages=True, smoothed=smoothed)
        lab, n = ndimage.label(st["g"] > 0, structure=np.ones((3, 3)))
        a = st["anchors"]
        cnt = np.bincount(lab[a[:, 1], a[:, 0]], minlength=n + 1)
        cpx = np.bincount(lab[st["chain_y"], st["chain_x"]], minlength=n + 1)
        shares.append(cnt.max() / len(a))
        print("%5d%s %10d %8d %18d (%.3f) %16d (%.3f)" % (fi + 1, " " if smoothed else "b", n, len(a), cnt.max(), cnt.max() / len(a),
                                                           cpx.max(), cpx.max() / max(1, len(st["chain_x"]))))
print("largest component's share of the anchors: min %.3f, median %.3f, max %.3f over %d frame variants ('b' = with the Gaussian "
      "pre-blur)" % (min(shares), float(np.median(shares)), max(shares), len(shares)))
