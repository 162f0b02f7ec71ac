"""CPU oracle for the UNet-ConvLSTM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product path (``unet-convlstm_amd``) never imports this package and raises
when its HIP library is missing.
"""
