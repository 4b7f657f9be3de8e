"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the HPMN hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  The product path
(``hpmn_amd``) never imports this package and fails loudly when its HIP
library is missing.

PARITY UNPINNED: the reference arithmetic lives in TensorFlow 1.4 (not
installable here) and the reference ships no tests,
golden vectors or seeds.  This oracle is therefore anchored on the cited
reference lines only, plus golden fixtures for the *input surface*
(``DataLoader`` / ``front_padding``) generated from the importable reference
code by ``tests/golden/make_golden.py``.
"""
