"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MAAVSS training hot path.

Plain PyTorch fp32 restatements (CPU) of the reference's algorithms. Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product (``maavss_amd``) never does and fails loudly
when its HIP library is missing.

Parity status
-------------
* ``avse_ref_cpu``  -- PINNED: checked against the reference's own
  ``AV_Fusion_Model_Frames`` (imported in the build container by
  ``oracle/make_golden.py``), outputs / loss / gradients / Adam step committed
  under ``tests/golden/``.
* ``stft_ref_cpu``  -- parity UNPINNED: the reference calls
  ``torchaudio.functional.spectrogram`` which is absent from the container and
  whose source is not under /root/reference; restated from its published
  semantics on ``torch.stft`` plus a direct float64 DFT cross-check.
* ``vit_ref_cpu``   -- parity UNPINNED: facebookresearch/dino is an empty,
  un-vendored submodule of the reference (no pinned commit, no weights);
  restated from the published ViT-S/8 architecture, cross-checked in the
  container against ``transformers.ViTModel`` built from a local config.
"""
