"""CPU oracle for the ChessVision CNN hot path  --  TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``chessvision-3lc_amd/``) never imports ``oracle`` and fails loudly
when the HIP extension is missing.

What it restates (reference = /root/reference, read-only):

* the UNet(3,1[,bilinear]) forward that ``chessvision/core.py:88,219-220`` runs.  The module
  source is an un-vendored git submodule (``.gitmodules:1-4``, Pytorch-UNet @ branch
  ``experimental``; directory empty in the reference checkout), so the tree is restated from
  the upstream milesial layout that the reference pins structurally: ctor signature
  (``scripts/train/train_unet.py:461-465``), ``n_channels`` attribute (``train_unet.py:296``)
  and the "module #52 = bottleneck ReLU" index (``train_unet.py:210,219``).
* the timm 1.0.15 ``resnet18(num_classes=13, in_chans=1)`` forward that
  ``chessvision/utils.py:32-39`` builds and ``core.py:240-242`` runs; the module tree is pinned
  by ``notebooks/model-summary.ipynb`` (94 modules, #90 = ``global_pool``, 11,176,909 params).
* the model-facing pre/post arithmetic of ``core.py:215-216,236-237,242,273`` and
  ``utils.py:101-112``.

The arithmetic itself is ``torch`` CPU fp32 -- the very kernels the reference executes
(``torch.nn.functional``), so the oracle is "kind: port" of the module trees on top of the
reference's own arithmetic library.

PARITY PIN STATUS: **parity unpinned at tensor level**.  The reference holds no golden tensors
for the UNet/ResNet forward and cannot be imported here (cv2, chess,
timm and the UNet submodule are absent).  What *is* pinned, in ``tests/test_oracle_structure.py``:
module counts / indices / parameter counts / MAC counts from the reference's own notebook and
training scripts, the ``extract_squares`` known-answer test of
``tests/test_chessvision.py:119-146``, and an independent plain-C restatement of every
primitive op (``oracle/c_ref/ops_ref.c``, fp64 accumulation) against which the torch ops are
cross-checked.
"""
