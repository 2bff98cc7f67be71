"""CPU oracle for the CryoVIT hot path -- TEST INFRASTRUCTURE ONLY.

This package is a torch-CPU fp32 restatement of the reference's algorithm for
the feature-extraction + segmentation-head path.  It is the *checker*, never
the product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  ``cryovit_amd`` never does, and fails
loudly when its HIP extension is missing.

Pinning status (see DESIGN.md "Oracle"):

* head (``oracle.head``), pre-processing (``oracle.preprocess``), K9 layout
  (``oracle.features``) and masked Dice (``oracle.dice``) are pinned against
  the reference's own source, executed where it lies under ``/root/reference``
  by AST extraction (``oracle/make_golden.py``); resulting vectors are the
  fixtures under ``tests/golden/``.
* ViT (``oracle.dinov2``): the arithmetic lives in the third-party
  ``facebookresearch/dinov2`` hub repo (unpinned branch, absent offline).  The
  restatement is cross-checked against the locally installed HF port
  ``transformers.models.dinov2_with_registers`` (random init).  Weight-level
  parity with the real checkpoint is **parity unpinned**.
"""
