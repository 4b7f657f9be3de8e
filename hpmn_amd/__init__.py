"""hpmn_amd -- MI355X-native (gfx950) implementation of HPMN's hot path.

Embedding gather -> K-layer periodic GRU memory update (forward + BPTT) -> attention read
-> TF-form Adam, behind the reference's ``hpmn.py`` train/eval entrypoint and the
``dataset_hpmn.pkl`` input format.  The numerical hot path lives in ``libhpmn_hip.so``
(``hpmn_amd/csrc``, C ABI in ``include/hpmn_hip.h``); see DESIGN.md.
"""
__version__ = "0.1.0"

# one rank of a multi-process job (WORLD_SIZE > 1): the HIP runtime settings the data-parallel step was measured with
# (hardware queues, dmabuf IPC) -- defaults only, and only effective before the process's first HIP call (dist.py)
from . import dist as _dist  # noqa: E402

_dist.apply_runtime_env()
