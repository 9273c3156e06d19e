"""TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch fp32, functional style) of the reference hot path:
the conv-SVG dynamics model, its losses, the train step and the CEM rollout
loop.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this package; the product package
`robot_aware_control_amd` never does.
"""
