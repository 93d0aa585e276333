"""oracle/ -- CPU restatement of RC-MVSNet's plane-sweep hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it; nothing under ``rc_mvsnet_amd/`` imports it and the product path raises when
    the HIP library is missing rather than falling back to anything here;
  * every function is written from the formulas of the reference (cited file:line, relative
    to ``/root/reference``) with plain PyTorch-CPU / numpy arithmetic in fp32 -- the fused
    ops (warp, hypothesis planes, depth head, sampler, gathers, compositing, 3-D conv taps)
    are spelled out index by index instead of calling the ATen composite the reference
    calls, so they specify the arithmetic the HIP kernels must reproduce;
  * parity pin: the reference ships NO tests, golden vectors or fixtures for this path
    (SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself,
    imported in the build container by ``tests/golden/make_golden.py`` (committed) and
    stored as small ``.npz`` fixtures under ``tests/golden/``;
    ``tests/test_oracle_golden.py`` checks every oracle function against them.

Later rows: ``unsup_loss`` (pinned to 1e-6 by the reference's own autograd), ``fusion`` and
``dataset`` (pinned by the reference's filter_depth / MVSDataset run with restatements of the absent cv2.remap / cv2.resize --
those two restatements are pinned by known answers worked out from OpenCV's published algorithm,
tests/test_cv_known_answers_cpu.py), ``aten_graph`` (the reference's op graph with autograd over a product module's parameters:
the comparator of the gradient tests, pinned by gradients of the imported reference, tests/golden/train_grads.npz).  ``bench.py --workload unsup_loss|fusion`` times them as
``cpu_baseline``; diagnostics that compare against the oracle live under ``tests/diag/``.

Tolerances: the reference's own CPU path is not bit-reproducible across ATen builds
(GCC contracts the AVX2 grid-sampler/conv code into FMAs in an unspecified order), so the
fixtures are compared at a few fp32 ulp (rtol 2e-5 / atol 2e-6 on O(1) features), and the
integer-valued confidence index is compared where it is not within rounding of a bin edge.
"""

from . import warp, conv3d, depth_head, feature_net, cascade, render, unsup_loss, fusion, dataset, aten_graph  # noqa: F401
