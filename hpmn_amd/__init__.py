"""hpmn_amd -- MI355X-native (gfx950) implementation of HPMN's hot path.

Embedding gather -> K-layer periodic GRU memory update (forward + BPTT) -> attention read
-> TF-form Adam, behind the reference's ``hpmn.py`` train/eval entrypoint and the
``dataset_hpmn.pkl`` input format.  The numerical hot path lives in ``libhpmn_hip.so``
(``hpmn_amd/csrc``, C ABI in ``include/hpmn_hip.h``); see DESIGN.md.
"""
__version__ = "0.1.0"
