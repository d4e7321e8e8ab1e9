"""CPU oracle for the Splice per-pair optimisation step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the CPU baseline -- never as the thing that is
measured or shipped.  The product (``splice_amd``) fails loudly when its HIP library
is missing; it never falls back to this package.

What is restated here (fp32, PyTorch-CPU, written from scratch):

* ``dino_vit``   -- the DINO VisionTransformer the reference pulls from torch.hub
                    (``models/extractor.py:20``; third-party ``facebookresearch/dino``
                    ``main``, file ``vision_transformer.py`` -- NOT in /root/reference,
                    un-pinned branch, restated from its published architecture).
* ``extractor``  -- ``models/extractor.py:4-9,132-163`` (key slicing, cosine self-sim).
* ``generator``  -- ``models/unet/skip.py:4-102`` + ``models/unet/common.py:11-124`` +
                    ``models/networks.py:24-58`` (5-scale skip U-Net).
* ``losses``     -- ``util/losses.py:11-105`` (lambda schedule, three losses, transforms).
* ``optim``      -- ``torch.optim.Adam`` as configured by ``util/util.py:28-32``.
* ``step``       -- ``train.py:51-80`` one optimisation step, reference-shaped
                    (6 ViT forwards + 3 backwards).

Pinning status (see DESIGN.md "Oracle"):
  - extractor / generator / losses / optim / step are PINNED against the reference
    modules imported in the build container (``oracle/make_golden.py`` drives the
    reference's own ``VitExtractor``, ``LossG``, ``Model``, ``define_G`` and writes
    ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).
  - the inside of the DINO ViT is "parity unpinned": the reference holds neither its
    source, weights, nor any test vector for it.  It is pinned only structurally
    (hook points, shapes, state-dict key names) through the reference's extractor.
"""
