"""CPU oracle of the SENAS hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package, and only
as the checker / the timed CPU baseline; ``senas_amd`` never does (``tests/test_host_logic.py::test_product_never_imports_oracle``).
"""
