"""CPU oracle for the ProbPose top-down inference hot path.

TEST INFRASTRUCTURE ONLY. This package restates, on the CPU (numpy / scipy / torch-CPU /
plain C), the arithmetic of the reference's hot path so that the HIP implementation in
``probpose_code_amd`` can be checked against it. It is never the thing shipped or
measured: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, call, link or execute anything under ``oracle/``. The product
package must not (``tests/test_no_oracle_in_product.py`` enforces that).

Pinning status (DESIGN.md §Oracle has the full table):

* decode (``decode_ref``; reference ``mmpose/codecs/utils/post_processing.py:13-39,308-430``,
  ``mmpose/codecs/probmap.py:170-220``, ``mmpose/models/utils/tta.py:35-39``) -- PINNED:
  checked bit-for-bit against outputs of the reference's own functions imported in
  isolation in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* Ex-OKS similarity (``exoks_ref``; reference ``mmpose/evaluation/metrics/_cocoeval.py:540-707``,
  ``mmpose/structures/keypoint/keypoints_min_padding.py:68-133``) -- PINNED: 60 synthetic cells scored by the
  reference's own ``COCOeval.computeExtendedOks`` (``tests/golden/make_golden_exoks.py`` ->
  ``tests/golden/exoks_cases.npz``, ``exoks_chain.npz``), matched to 1e-12.
* Ex-mAP evaluator (``exmap_ref``; reference ``mmpose/evaluation/metrics/_cocoeval.py:161-503,709-1190``:
  ``_prepare``, ``evaluateImg``, ``accumulate``, ``summarize``) -- PINNED: five synthetic datasets run through the
  reference's own ``COCOeval.evaluate(); accumulate(); summarize()`` (``tests/golden/make_golden_exmap.py`` ->
  ``tests/golden/exmap_cases.npz``); precision / recall / scores / stats and every per-image match reproduced exactly.
* Sparsemax (PyPI ``sparsemax``, un-vendored, unpinned in ``requirements/build.txt:4``),
  ViT backbone (``mmpretrain==1.2.0`` ``VisionTransformer``, un-vendored), ``ProbMapHead``
  network (needs mmcv/mmengine, not importable) -- PARITY UNPINNED: restated from the
  published algorithm / the cited reference lines; no reference test or golden vector
  exists for them. Known-answer tests pin the restatement itself.
"""
