"""CPU oracle for the QAT-ViT student step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package; the product package
(``qat-vit_amd/``) never does.  See ``oracle/README.md`` for the pin status.
"""
